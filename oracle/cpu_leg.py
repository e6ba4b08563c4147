"""oracle/cpu_leg.py — TEST INFRASTRUCTURE ONLY: one worker of bench.py's all-cores CPU leg.

    python -m oracle.cpu_leg ROWS E_PERCENT SECONDS R0 GROWTH

Synthesises the bench table (seed 42) and runs the linear-time C restatement of the reference's CLT monitor
(oracle/aqe_oracle.c, aqo_clt_run: custom_bplus_db.cpp:885-1043) on it for about SECONDS; prints one JSON line
{"queries": k, "seconds": t, "samples": n}.  bench.py starts one of these per host core (each its own process:
its own address space, so the workers do not serialise on page faults of one mm).
"""
import json
import sys
import time


def main(argv):
    from oracle.pyoracle import Oracle
    rows_n, e, secs, r0, growth = int(argv[0]), float(argv[1]), float(argv[2]), int(argv[3]), int(argv[4])
    o = Oracle()
    rows = o.synth(rows_n, 42)
    pct = 20.0 if e <= 1.0 else 15.0 if e <= 2.0 else 10.0 if e <= 5.0 else 5.0  # enhanced_aqe_cli.py:243-250
    rc, res, _ = o.clt_run(rows, pct, 0.95, 10, 4, e, R0=r0, growth=growth)  # warm (page faults of the first call)
    print(json.dumps({"ready": True}), flush=True)
    sys.stdin.readline()  # all workers start together
    t0 = time.perf_counter()
    k = 0
    while True:
        rc, res, _ = o.clt_run(rows, pct, 0.95, 10, 4, e, R0=r0, growth=growth)
        k += 1
        if time.perf_counter() - t0 >= secs:
            break
    print(json.dumps({"queries": k, "seconds": time.perf_counter() - t0, "samples": int(res.final.n)}), flush=True)


if __name__ == "__main__":
    main(sys.argv[1:])
