#!/usr/bin/env python3
"""bench.py — aggregates/sec and achieved HBM GB/s of the sampled-reduce path on MI355X.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 10 M-row synthetic `sales` table per GPU (f64 amount, seed 42),
`SELECT AVG(amount) ... --e 0.01` through the CLT dual-pointer monitor exactly as the reference CLI
issues it (enhanced_aqe_cli.py:243-255): pct = 20 (e <= 1), confidence 0.95, check_interval 10,
e = 0.01 PERCENT — which cannot converge on 10 M rows, so every query performs the full fast+slow sweep
(4 M samples per GPU) with the should_stop test armed.  The table is resident in HBM.

One STEP = `--launches-per-step` (100) launches; one launch = one BATCH of `--batch` (32) different queries
(pointer counts T = 4, 6, ... 16, aggregates AVG / SUM / COUNT, thresholds a hair apart), every query doing all of
its own loads and producing value + 95 % interval; two batches alternate so that EVERY result is fetched inside
the timed loop.  value = queries completed per second.

N GPUs (weak scaling: per-GPU rows fixed): each rank holds its own 10 M-row region of an N x 10 M-row table; a query
runs T x N pointers over the global table, each rank sweeps what falls in its region (the same one launch, decisions
left out), ONE RCCL all-reduce of the per-round moment vectors serves the whole batch, and one launch replays the stop
rules.  `value` is GLOBAL queries per second (a query over N regions counts once).

Output: the LAST stdout line is one compact JSON object (< 3 KB: metric, value, roofline, roofline_hbm,
cpu_baseline); everything else — per-configuration lines, cold staging, open loop, every CPU leg — goes to
bench_report.json (repo root, and gpurun_out/ when present) and, one object per line, to stderr.
`--headline-only` skips those.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md; 6.0-6.3 TB/s is what streaming kernels reach)
ROWS_PER_GPU = 10_000_000
SEED = 42
CLT_ROUND0 = 4096   # samples per pointer in round 0 ...
CLT_GROWTH = 4      # ... times 4 every round: 5 rounds cover the 1 M-sample progressions
PMC_FILE = ROOT / "profiles" / "round3_pmc_raw.json"
LAUNCHES_PER_STEP = 100  # batch launches per step: the timed region at --steps 20 is ~100 ms
LINE_LIMIT = 3000       # the driver keeps ~8 KB of stdout; the final line stays well inside it


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--launches-per-step", type=int, default=LAUNCHES_PER_STEP, help="batch launches per step")
    ap.add_argument("--rows-per-gpu", type=int, default=ROWS_PER_GPU)
    ap.add_argument("--error-percent", type=float, default=0.01, help="--e of the reference CLI, in percent")
    ap.add_argument("--batch", type=int, default=32, help="independent queries per step (one launch serves them all)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="skip configs / cold / early-termination reading / cpu baseline")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-configuration list (100 M and 1 B-row tables)")
    ap.add_argument("--max-config-rows", type=int, default=1_000_000_000)
    ap.add_argument("--cpu-sample-rows", type=int, default=2_000_000)
    return ap.parse_args()


def source_hash() -> str:
    """Hash of the library's sources: the PMC traffic in profiles/ is only reported for the code it was measured on."""
    h = hashlib.sha256()
    files = sorted((ROOT / "approximatequeryengine_amd" / "csrc").glob("*")) + [ROOT / "include" / "aqe_hip.h"]
    for f in files:
        if f.suffix in (".hip", ".hpp", ".cpp", ".h"):
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


def e_to_pct(e: float) -> float:  # enhanced_aqe_cli.py:243-250
    return 20.0 if e <= 1.0 else 15.0 if e <= 2.0 else 10.0 if e <= 5.0 else 5.0


def headline_queries(nat, make_query, B: int, world: int, err: float):
    """The B queries of one step.  They differ: T = 4, 6, ... 16 pointers (x N GPUs) — other regions, other families,
    the same 20 % of the rows (the CLI fixes pct by e) — aggregates AVG / SUM / COUNT, and thresholds a hair apart."""
    pct = e_to_pct(err)
    return [make_query(nat.M_CLT_DUAL_POINTER, pct, agg=(nat.AVG, nat.SUM, nat.COUNT)[i % 3], confidence_level=0.95, check_interval=10,
                       num_threads=(4 + 2 * (i % 7)) * world, max_error_percent=err * (1.0 + 1e-3 * (i // 7)),
                       clt_round0=CLT_ROUND0, clt_growth=CLT_GROWTH) for i in range(B)]


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference's own C++ (oracle/_ref, built in the authoring container from
# /root/reference) and its linear-time C restatement (oracle/), timed on this box's host cores.
# ---------------------------------------------------------------------------------------------------------------
def cpu_baseline(rows_per_gpu: int, e: float, sample_rows: int) -> dict:
    import numpy as np
    from oracle.pyoracle import Oracle, Ref, ref_available
    o = Oracle()
    pct = e_to_pct(e)
    cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = cores
    quota = None  # cgroup v2 CPU quota of this container, in cores
    try:
        q_, per_ = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if q_ != "max":
            quota = float(q_) / float(per_)
    except Exception:
        pass
    out = {"host_cores": cores, "usable_cores": usable, "cgroup_cpu_quota": quota}
    rows = o.synth(rows_per_gpu, SEED)

    def port_query():
        rc, res, _ = o.clt_run(rows, pct, 0.95, 10, 4, e, R0=CLT_ROUND0, growth=CLT_GROWTH)
        return res

    # (a) linear-time port, one thread, full workload
    t0 = time.perf_counter()
    reps = 0
    while True:
        res = port_query()
        reps += 1
        if time.perf_counter() - t0 > 2.0 or reps >= 20:
            break
    dt = (time.perf_counter() - t0) / reps
    samples = int(res.final.n)
    out["port"] = {"value": 1.0 / dt, "unit": "aggregates/sec", "cores": 1, "kind": "port",
                   "samples_per_sec": samples / dt, "GBps_touched": 8.0 * samples / dt / 1e9,
                   "sample": f"full workload ({rows_per_gpu:,} rows, {samples:,} samples/query), {reps} queries, "
                             "oracle/aqe_oracle.c single thread, linear-time restatement (in-place moments: best-effort CPU)"}
    # (b) the same port with every usable core busy: one worker PROCESS per core (oracle/cpu_leg.py), each with its own
    # copy of the table, started together, about 3 s each
    import subprocess
    T = max(1, min(usable, int(quota) if quota and quota >= 1 else usable, 32))  # (a one-GPU box gets a 16-core share)
    env = dict(os.environ, MALLOC_MMAP_MAX_="0", MALLOC_TRIM_THRESHOLD_="2000000000", PYTHONPATH=str(ROOT))
    procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_leg", str(rows_per_gpu), repr(e), "3.0", str(CLT_ROUND0), str(CLT_GROWTH)],
                              stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, cwd=str(ROOT), env=env) for _ in range(T)]
    try:
        for p_ in procs:
            assert json.loads(p_.stdout.readline())["ready"]
        for p_ in procs:
            p_.stdin.write("go\n")
            p_.stdin.flush()
        got = [json.loads(p_.stdout.readline()) for p_ in procs]
        rate = sum(g["queries"] / g["seconds"] for g in got)
        out["port_all_cores"] = {"value": rate, "unit": "aggregates/sec", "cores": T, "kind": "port",
                                 "samples_per_sec": samples * rate, "GBps_touched": 8.0 * samples * rate / 1e9,
                                 "sample": f"{T} worker processes (one per usable host core, at most 32), each the full workload on its own copy of the table "
                                           f"for ~3 s ({sum(g['queries'] for g in got)} queries in all)"}
    finally:
        for p_ in procs:
            try:
                p_.stdin.close()
                p_.wait(timeout=30)
            except Exception:
                p_.kill()
    if not ref_available():
        return out
    r = Ref()
    try:
        # (c) the reference's CLT monitor.  It re-scans all of a worker's samples at every check
        # (custom_bplus_db.cpp:936-946, 993-1003): cost quadratic in rows.  Timed on a bounded table first; the full
        # 10 M-row table only if the quadratic law says it ends within the budget.
        n = min(sample_rows, rows_per_gpu)
        r.fill_direct(rows[:n].copy())
        t0 = time.perf_counter()
        ids = r.sample("clt_validated_dual_pointer_sample", pct, 0.95, 10, 4, e)
        amt = r.last_amounts(len(ids))
        _ = float(np.sum(amt)) / max(len(amt), 1)
        dt_small = time.perf_counter() - t0
        scale = (rows_per_gpu / n) ** 2
        predicted = dt_small * scale
        ref_leg = {"unit": "aggregates/sec", "cores": 4, "kind": "reference", "host_cores": cores,
                   "bounded_run": {"rows": n, "seconds": dt_small, "samples_returned": int(len(ids))}}
        r.fill_direct(rows)
        if n < rows_per_gpu and predicted <= 45.0:
            t0 = time.perf_counter()
            ids = r.sample("clt_validated_dual_pointer_sample", pct, 0.95, 10, 4, e)
            amt = r.last_amounts(len(ids))
            _ = float(np.sum(amt)) / max(len(amt), 1)
            dt_full = time.perf_counter() - t0
            ref_leg.update({"value": 1.0 / dt_full, "rows_per_sec": rows_per_gpu / dt_full, "samples_per_sec": len(ids) / dt_full,
                            "sample": f"1 query on the full {rows_per_gpu:,}-row table ({len(ids):,} samples returned, {dt_full:.1f} s), "
                                      "clt_validated_dual_pointer_sample(20, 0.95, 10, 4, e) with 4 std::async workers as the CLI hard-codes "
                                      "(enhanced_aqe_cli.py:253-255) + the CLI's mean over the returned amounts"})
        else:
            ref_leg.update({"value": 1.0 / predicted if n < rows_per_gpu else 1.0 / dt_small,
                            "sample": f"1 query on a {n:,}-row table took {dt_small:.2f} s; the full {rows_per_gpu:,}-row figure is EXTRAPOLATED by the "
                                      f"monitor's quadratic cost (x{scale:.0f} = {predicted:.0f} s per query; it re-scans all samples at every "
                                      "check, custom_bplus_db.cpp:936-946), because running it would take more than the bench's CPU budget"})
        out["reference"] = ref_leg
        # (d) the `--s 1` path as the reference really runs it: sampler returning records on the full table.
        # memory_stride_sample uses the flat cache (DB.cpp:1540-1566); random_pointer_sample copies all 32 N bytes
        # first (collect_leaf_records, DB.cpp:715-735, 856-882) and materialises its records.  The Python-side
        # list conversion and generator sum of the CLI (CLI:189-200) are NOT included (no pybind11 module here):
        # these are lower bounds of the reference's cost.
        legs = {}
        for name, args in (("memory_stride_sample", (1.0, 0.0)), ("random_pointer_sample", (1.0, 42.0)), ("block_sample", (1.0, 1000.0))):
            t0 = time.perf_counter()
            k = 0
            while True:
                ids = r.sample(name, *args)
                amt = r.last_amounts(len(ids))
                _ = float(np.sum(amt)) * rows_per_gpu / max(len(amt), 1)
                k += 1
                if time.perf_counter() - t0 > 1.5 or k >= 50:
                    break
            dts = (time.perf_counter() - t0) / k
            legs[name] = {"aggregates_per_sec": 1.0 / dts, "ms_per_query": 1e3 * dts, "samples": int(len(ids)), "queries_timed": k,
                          "rows_per_sec": rows_per_gpu / dts,
                          "GBps_touched": (32.0 * rows_per_gpu if name == "random_pointer_sample" else 32.0 * len(ids)) / dts / 1e9}
        out["reference_sampling_1pct"] = {"kind": "reference", "cores": 1, "rows": rows_per_gpu, "legs": legs,
                                          "note": "reference-faithful: record-returning sampler + sum of the returned amounts; the CLI's "
                                                  "pybind11 list conversion and Python generator sum (0.67 us per sampled row, BASELINE.md) come on top"}
    finally:
        r.close()
    return out


# ---------------------------------------------------------------------------------------------------------------
# Per-configuration roofline lines (one query in flight, per-launch HIP events attached to the dispatch)
# ---------------------------------------------------------------------------------------------------------------
# Every measured case runs its timed launches on a stream of its OWN (nothing else ever runs there), and says how many
# launches that were: tools/cases_from_trace.py splits a `rocprofv3 --kernel-trace` of this very command by stream and
# writes profiles/round3_configs_cases.csv — one row per (case, kernel, grid) from which every `frac` here can be recomputed.
CASES = []


def fresh_stream():
    import torch
    return torch.cuda.Stream()


def log_case(name, kernel, launches, queries, alg_bytes, event_us):
    CASES.append({"case": name, "kernel": kernel, "launches": int(launches), "queries": int(queries), "algorithmic_bytes_per_query": float(alg_bytes),
                  "event_us_per_query": float(event_us)})


def measure_config(eng, st, name, q, reps=20, note=None):
    import statistics
    import torch
    from approximatequeryengine_amd import _native as nat
    plan = eng.plan(q)
    try:
        for _ in range(3):
            plan.enqueue_all(st)
            r = plan.fetch(st)
        torch.cuda.synchronize()
        own = fresh_stream()
        plan.set_profiling(True)
        per_query, lat = [], []
        for _ in range(reps):
            plan.enqueue_all(own.cuda_stream)
            r = plan.fetch(own.cuda_stream)
            per_query.append(plan.launch_ms())
        plan.set_profiling(False)
        torch.cuda.synchronize()
        for _ in range(reps):  # closed loop without the profiling events (the fetch then polls the pinned result)
            t0 = time.perf_counter()
            plan.enqueue_all(st)
            r = plan.fetch(st)
            lat.append(time.perf_counter() - t0)
        tot = sorted(sum(x) for x in per_query)
        us = 1e3 * statistics.median(tot)
        nbytes = 8.0 * r.visited
        out = {"config": name, "kernel": nat.KERNEL_NAMES.get(plan.last_kernel(), "?"), "samples": int(r.visited), "n": int(r.n), "value": r.value, "ci": [r.ci_lower, r.ci_upper],
               "converged": int(r.converged), "rounds": int(r.rounds), "topup_rows": int(r.topup),
               "launches_per_query": len(per_query[-1]), "kernel_us": us, "kernel_us_min": 1e3 * tot[0],
               "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / (us * 1e-6) / 1e9 if us > 0 else None,
               "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS if us > 0 else None,
               "closed_loop_us_p50": 1e6 * statistics.median(lat)}
        if note:
            out["note"] = note
        log_case(name, out["kernel"], sum(len(x) for x in per_query), reps, nbytes, us)
        return out
    finally:
        plan.close()


def measure_batch(eng, Batch, st, name, queries, reps=20):
    """A batch of queries in ONE launch (k_sweep_lean_multi, or k_sweep_multi when a plan of the batch has families that are
    not plain runs of rows): launch time from the dispatch's events, every result fetched."""
    import statistics
    from approximatequeryengine_amd import _native as nat
    plans = [eng.plan(q) for q in queries]
    b = Batch(plans)
    try:
        for _ in range(3):
            b.enqueue_all(st)
            rs = b.fetch()
        import torch
        torch.cuda.synchronize()
        own = fresh_stream()
        b.set_profiling(True)
        ms, lat = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            b.enqueue_all(own.cuda_stream)
            rs = b.fetch()
            lat.append(time.perf_counter() - t0)
            m, swept, wgs = b.launch_info()
            ms.append(m)
        b.set_profiling(False)
        torch.cuda.synchronize()
        us = 1e3 * statistics.median(ms)
        nbytes = 8.0 * sum(r.visited for r in rs)
        wall = statistics.median(lat)
        log_case(name, nat.KERNEL_NAMES.get(plans[0].last_kernel(), "?"), reps, reps, nbytes, us)
        return {"config": name, "kernel": nat.KERNEL_NAMES.get(plans[0].last_kernel(), "?"), "queries_per_launch": len(queries),
                "samples": int(sum(r.visited for r in rs)), "kernel_us": us,
                "kernel_us_min": 1e3 * min(ms), "workgroups": int(wgs), "algorithmic_bytes": nbytes,
                "achieved_GBps": nbytes / (us * 1e-6) / 1e9, "frac": nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                "closed_loop_us_p50": 1e6 * wall, "aggregates_per_sec_one_batch_in_flight": len(queries) / wall,
                "value0": rs[0].value, "ci0": [rs[0].ci_lower, rs[0].ci_upper]}
    finally:
        b.close()
        for p in plans:
            p.close()


def run_configs(eng, nat, make_query, st, max_rows, Batch=None):
    """SURVEY §8d configs 1, 3-local, 4-local, 5 + the dense scans: launch time from dispatch events, 8 B per sampled row."""
    out = []
    lib = nat.lib()

    def clt(e):
        return make_query(nat.M_CLT_DUAL_POINTER, lib.aqe_error_to_sample_percent(e), agg=nat.AVG, max_error_percent=e,
                          clt_round0=CLT_ROUND0, clt_growth=CLT_GROWTH)

    # configs[0] on the bench table (10 M rows): --s 1 % strided, seeded random, block
    n = eng.info().global_rows
    tag = f"{n // 1_000_000}M"
    out.append(measure_config(eng, st, f"config0 {tag} stride 1% SUM (memory_stride_sample)", make_query(nat.M_MEMORY_STRIDE, 1.0),
                              note="latency-bound: 100 k samples, 0.8 MB"))
    out.append(measure_config(eng, st, f"config0 {tag} random 1% seed 42 SUM (random_pointer_sample, host mt19937 index list)",
                              make_query(nat.M_RANDOM_POINTER, 1.0, seed=42), note="sparse gather: 8 B useful of every 64-B sector"))
    out.append(measure_config(eng, st, f"config0 {tag} random 1% seed 42 SUM drawn ON THE DEVICE (AQE_M_RANDOM_DEVICE: keyed bijection, no index list)",
                              make_query(nat.M_RANDOM_DEVICE, 1.0, seed=42), note="statistical parity with the reference's random samplers; same sparse gather"))
    out.append(measure_config(eng, st, f"{tag} block 1% B=1000 SUM", make_query(nat.M_BLOCK, 1.0)))
    out.append(measure_config(eng, st, f"{tag} exact SUM (full scan)", make_query(nat.M_EXACT, 100.0)))
    q_inplace = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=CLT_ROUND0, clt_growth=CLT_GROWTH)
    q_inplace.flags |= nat.Q_NO_LAYOUT
    out.append(measure_config(eng, st, f"{tag} CLT AVG e=0.01% swept IN PLACE (what a plan gets when all 8 stride-major views of a table are held)",
                              q_inplace, note="rows 0 and 2 of every 5: 40 % of every line touched; the view form of the same query is the headline"))
    out.append(measure_config(eng, st, f"{tag} CLT AVG e=0.01% with T=256 pointers (one pointer per compute unit: the tuned variant of config 2)",
                              make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=CLT_ROUND0, clt_growth=CLT_GROWTH, num_threads=256),
                              note="512 runs: the wide form of the lean launch (run table in LDS, a tile's run found by bisection)"))
    if Batch is not None:  # configs[0] as a service: 32 different 1 % queries per launch
        qs = [make_query(nat.M_MEMORY_STRIDE, 1.0, agg=(nat.SUM, nat.AVG, nat.COUNT)[i % 3], rows=(100_000 * i, n - 50_000 * i)) for i in range(16)]  # key-range windows
        qs += [make_query(nat.M_BLOCK, 1.0, block_size=500 + 100 * i, where=(100.0 + i, 900.0 - i), convention=nat.EST_CPP) for i in range(16)]
        out.append(measure_batch(eng, Batch, st, f"config0 {tag}: batch of 32 different 1% queries (16 strided, 16 block + WHERE) in ONE launch", qs))
    if Batch is not None and max_rows >= 320_000_000:
        # k_sweep_multi where no byte can be shared between the queries of a batch and nothing fits a cache: 32 exact
        # SUMs over DISJOINT 10 M-row key ranges (id BETWEEN ...) of a 320 M-row table = 2.56 GB per launch from HBM
        eng.generate_synthetic(320_000_000)
        qs = [make_query(nat.M_EXACT, 100.0, agg=(nat.SUM, nat.AVG)[i % 2], rows=(10_000_000 * i, 10_000_000 * (i + 1))) for i in range(32)]
        out.append(measure_batch(eng, Batch, st, "320M-row table: batch of 32 exact SUM/AVG over DISJOINT 10M-row key ranges in ONE launch (no shared bytes, HBM proper)", qs, reps=10))
    for rows in (100_000_000, 1_000_000_000):
        if rows > max_rows:
            continue
        t0 = time.perf_counter()
        eng.generate_synthetic(rows)
        gen_s = time.perf_counter() - t0
        tag = f"{rows // 1_000_000}M" if rows < 1_000_000_000 else "1B"
        if rows == 100_000_000:
            out.append(measure_config(eng, st, f"config2-local {tag} stride 1% SUM", make_query(nat.M_MEMORY_STRIDE, 1.0)))
            out.append(measure_config(eng, st, f"config4 {tag} block 1% B=1000 SUM WHERE amount in [250,750]",
                                      make_query(nat.M_BLOCK, 1.0, where=(250.0, 750.0), convention=nat.EST_CPP)))
            out.append(measure_config(eng, st, f"{tag} CLT AVG e=0.01% (never converges)", clt(0.01)))
            if Batch is not None:  # the batch form on an HBM-resident table
                out.append(measure_batch(eng, Batch, st, f"{tag}: batch of 8 different CLT e=0.01% queries in ONE launch (a table 10x the Infinity Cache: what the "
                                         "groups share in the caches depends on how far they drift apart over the sweep - 0.71 to 0.95 from run to run)", headline_queries(nat, make_query, 8, 1, 0.01), reps=10))
        else:
            out.append(measure_config(eng, st, f"config3-local {tag} CLT AVG e=0.005% (never converges: full 20% dual-pointer sweep)", clt(0.005), reps=10))
            out.append(measure_config(eng, st, f"config3-local {tag} CLT AVG e=0.5% (stops early + top-up)", clt(0.5), reps=10))
        out.append(measure_config(eng, st, f"{tag} exact SUM (full scan)", make_query(nat.M_EXACT, 100.0), reps=10))
        out.append(measure_config(eng, st, f"{tag} stride 20% SUM", make_query(nat.M_MEMORY_STRIDE, 20.0), reps=10))
        out.append(measure_config(eng, st, f"{tag} block 20% B=1000 SUM", make_query(nat.M_BLOCK, 20.0), reps=10))
        out[-1]["table_generate_s"] = gen_s
    return out


def cold_numbers(Engine, nat, make_query, rows: int, e: float) -> dict:
    """Cold path (SURVEY §8d, config 3's staging leg): a table in the reference's file format (page cache) -> HBM, then the
    first CLT query (which builds the stride-major views of the column), then the same query warm.  Per staging: where the
    time went (aqe_last_stage_stats) — device allocation, pinning the ring (first staging of a context only), host threads
    filling pinned buffers (pread out of the page cache), the host waiting for the copy engine."""
    import tempfile
    pct = e_to_pct(e)
    q = make_query(nat.M_CLT_DUAL_POINTER, pct, agg=nat.AVG, max_error_percent=e, clt_round0=CLT_ROUND0, clt_growth=CLT_GROWTH)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "sales.db")
        t0 = time.perf_counter()
        with Engine(0) as g:
            g.generate_synthetic(rows, keep_aos=True)
            g.save_file(path)
        size = os.path.getsize(path)
        out = {"rows": rows, "file_bytes": size, "write_file_s": time.perf_counter() - t0}
        for keep in (False, True):
            with Engine(0) as eng:
                t0 = time.perf_counter()
                eng.stage_file(path, keep_aos=keep)
                t1 = time.perf_counter()
                st1 = eng.stage_stats().as_dict()
                r = eng.reduce(q)  # plan + both stride-major views + the query
                t2 = time.perf_counter()
                r = eng.reduce(q)
                t3 = time.perf_counter()
                info = eng.info()
                eng.stage_file(path, keep_aos=keep)  # the same context again: the pinned ring is there already
                t4 = time.perf_counter()
                st2 = eng.stage_stats().as_dict()
                out["amounts_only" if not keep else "rows_kept"] = {
                    "stage_file_ms": 1e3 * (t1 - t0), "file_GBps": size / (t1 - t0) / 1e9, "stages": st1,
                    "restage_ms (pinned ring kept)": st2["total_ms"], "restage_file_GBps": size / (st2["total_ms"] * 1e-3) / 1e9, "restage_stages": st2,
                    "first_query_ms (plan + view build + sweep)": 1e3 * (t2 - t1), "second_query_ms": 1e3 * (t3 - t2),
                    "hbm_bytes": int(info.hbm_bytes), "avg": r.value}
        out["note"] = ("file written by aqe_save_file in the reference's format (24-byte header + 32-byte rows, custom_bplus_db.cpp:665-711), read back from "
                       "the page cache with pread into a ring of pinned buffers (16 host threads), hipMemcpyAsync from there; replaces load_from_file + "
                       "collect_leaf_records (DB.cpp:685-735)")
    return out


# ---------------------------------------------------------------------------------------------------------------
# The line the driver parses (pure functions: tests/test_bench_line.py formats one from canned numbers)
# ---------------------------------------------------------------------------------------------------------------
def unique_sampled_rows(nat, queries, n_global: int, lo: int = 0, hi=None) -> int:
    """How many DISTINCT rows of [lo, hi) the queries of one launch sample (host-side planner, no GPU): the bytes a launch
    has to fetch at least once, whatever its queries share in the caches."""
    import numpy as np
    hi = n_global if hi is None else hi
    seen = np.zeros(hi - lo, dtype=bool)
    done = set()
    for q in queries:
        key = (q.method, q.sample_percent, q.num_threads, q.clt_round0, q.clt_growth, q.check_interval, q.row_lo, q.row_hi)
        if key in done:  # aggregate and threshold do not change which rows a never-converging query sweeps
            continue
        done.add(key)
        _, rounds, _ = nat.plan_families(q, n_global, lo, hi, 0)
        for r in range(rounds):
            for f in nat.plan_families(q, n_global, lo, hi, r)[0]:
                o = np.arange(f.ord_lo, f.ord_hi, dtype=np.int64)
                seen[f.row0 + (o // f.seg_len) * f.pitch + (o % f.seg_len) * f.step - lo] = True
                if f.flags & nat.F_PAIR:
                    o = np.arange(f.ord_lo_b, f.ord_hi_b, dtype=np.int64)
                    seen[f.row0_b + o * f.step - lo] = True
    return int(seen.sum())


def headline_roofline(kernel: str, avg_launch_us: float, algorithmic_bytes: float, unique_bytes: float, traffic, executed_bytes: float,
                      traffic_source=None) -> dict:
    """HBM roofline of the launch that serves a batch.  `algorithmic` = 8 B x the rows the SUM / AVG queries of the batch
    sample (COUNT is metadata, SURVEY 8d: 0 B).  The queries of a batch sample the same rows and meet in the compute dies'
    L2s, so most of those bytes never cross the fabric: `achieved` is priced on the bytes that DO — the PMC traffic when
    profiles/ holds it for these very sources, else the distinct sampled bytes (each fetched at least once) — never more
    than the algorithmic bytes.  The rate of all executed loads is kept as `l2_rate`: informational, not a roofline."""
    t = avg_launch_us * 1e-6
    basis, moved = ("pmc_traffic", float(traffic)) if traffic else ("unique_bytes", float(unique_bytes))
    moved = min(moved, float(algorithmic_bytes))
    achieved = moved / t / 1e9
    return {"kernel": kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": min(achieved / HBM_PEAK_GBPS, 1.0), "traffic": traffic, "basis": basis, "bytes_priced": moved,
            "algorithmic_bytes_per_launch": float(algorithmic_bytes), "unique_bytes_per_launch": float(unique_bytes),
            "avg_launch_us": avg_launch_us, "l2_rate_GBps": executed_bytes / t / 1e9, "traffic_source": traffic_source}


def _r(x, nd=4):
    return None if x is None else (round(x, nd) if isinstance(x, float) else x)


def compact_line(rep: dict) -> dict:
    """bench_report.json -> the one line on stdout: the contract's keys, `roofline`, `roofline_hbm`, `cpu_baseline`."""
    cfg, rf = rep["config"], rep["roofline"]
    line = {k: rep[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data")}
    line["value"], line["ms_per_step"] = _r(line["value"], 1), _r(line["ms_per_step"], 4)
    line["config"] = {k: cfg[k] for k in ("workload", "rows_per_gpu", "global_rows", "queries_per_step", "launches_per_step", "queries_per_launch",
                                          "samples_per_query_per_gpu", "collectives_per_step", "collective") if k in cfg}
    line["roofline"] = {k: _r(rf.get(k), 4 if k == "frac" else 1) for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "basis",
                                                                            "algorithmic_bytes_per_launch", "unique_bytes_per_launch", "avg_launch_us", "l2_rate_GBps")}
    if rep.get("roofline_hbm"):
        line["roofline_hbm"] = [{k: _r(c.get(k), 4 if k == "frac" else 1) for k in ("case", "kernel", "avg_launch_us", "algorithmic_bytes", "achieved", "frac")}
                                for c in rep["roofline_hbm"][:4]]
    if rep.get("single_query"):
        sq = rep["single_query"]
        line["single_query"] = {"kernel": sq["kernel"], "avg_launch_us": _r(sq["avg_launch_us"], 2), "frac": _r(sq["frac"], 4),
                                "closed_loop_us_p50": _r(sq["closed_loop_latency_us"]["p50"], 2)}
        ch = sq.get("closed_loop_c_host")
        if isinstance(ch, list) and ch:
            line["single_query"]["closed_loop_c_host_us_p50"] = _r(ch[0].get("p50_us"), 2)
    cb = rep.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: (_r(cb.get(k), 5) if k == "value" else cb.get(k)) for k in ("value", "unit", "cores", "host_cores", "kind", "sample")}
        if line["cpu_baseline"].get("sample") and len(line["cpu_baseline"]["sample"]) > 240:
            line["cpu_baseline"]["sample"] = line["cpu_baseline"]["sample"][:237] + "..."
    if rep.get("n_gpus", 1) > 1:  # how to read `value` against the N = 1 line
        line["scaling_detail"] = {"rows_swept_per_sec": _r(rep.get("rows_swept_per_sec"), 0), "region_aggregates_per_sec": _r(rep.get("region_aggregates_per_sec"), 1),
                                  "note": "weak: every rank holds rows_per_gpu rows (the table grows with N); value = completed queries over the GLOBAL table, "
                                          "each one sweep per rank + one collective per batch: ideal is value constant in N while rows_swept_per_sec grows N-fold"}
    line["report"] = "bench_report.json"
    if len(json.dumps(line)) > LINE_LIMIT:  # never let a long string cost the driver the whole line
        line["config"]["workload"] = line["config"]["workload"][:200]
        line.pop("single_query", None)
    assert len(json.dumps(line)) <= LINE_LIMIT, len(json.dumps(line))
    return line


def write_report(rep: dict):
    text = json.dumps(rep, indent=1)
    for d in (ROOT, ROOT / "gpurun_out"):
        if d.is_dir():
            try:
                (d / "bench_report.json").write_text(text)
            except OSError:
                pass
    for k, v in rep.items():  # one object per line on stderr: readable in a log tail, never in the way of the stdout line
        if isinstance(v, list) and v and isinstance(v[0], dict):
            for item in v:
                print(json.dumps({k: item}), file=sys.stderr)
        elif isinstance(v, dict):
            print(json.dumps({k: v}), file=sys.stderr)
    sys.stderr.flush()


def c_host_closed_loop(rows: int, reps: int = 300):
    """The same closed loop from a plain-C host (tests/c_host/closed_loop.c: no Python, no ctypes): built with gcc here,
    run as a child process while this one keeps its table (two contexts on one GPU)."""
    import subprocess
    import tempfile
    from approximatequeryengine_amd.build import LIB
    try:
        with tempfile.TemporaryDirectory() as td:
            exe = os.path.join(td, "closed_loop")
            subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-I", str(ROOT / "include"), str(ROOT / "tests" / "c_host" / "closed_loop.c"), "-o", exe,
                                   "-L", str(LIB.parent), "-laqe_hip", f"-Wl,-rpath,{LIB.parent}", "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            env = dict(os.environ)
            env["LD_LIBRARY_PATH"] = os.pathsep.join(["/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
            out = subprocess.run([exe, str(rows), str(reps)], env=env, capture_output=True, text=True, timeout=120)
            if out.returncode != 0:
                return {"error": (out.stderr or out.stdout)[-300:]}
            return [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    except Exception as ex:
        return {"error": repr(ex)}


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    from approximatequeryengine_amd import _native as nat
    from approximatequeryengine_amd.distributed import PipelinedBatches, ShardedBatch, shard_bounds, torch_all_reduce
    from approximatequeryengine_amd.engine import Batch, Engine, make_query

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    # AQE_BENCH_REHEARSAL=1: several ranks share ONE GPU over gloo (to exercise the N>1 code path on a one-GPU
    # box); AQE_BENCH_FORCE_DIST=1: the N>1 code path through RCCL with a world of one.  Never used for reported numbers.
    rehearsal = os.environ.get("AQE_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    use_dist = world > 1 or os.environ.get("AQE_BENCH_FORCE_DIST") == "1"
    torch.cuda.set_device(local_rank)
    collective = None
    if use_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        collective = {"backend": dist.get_backend(), "ranks": dist.get_world_size()}

    rows = args.rows_per_gpu
    n_global = rows * world
    lo, hi = shard_bounds(n_global, world, rank)
    e = args.error_percent
    pct = e_to_pct(e)
    B = max(1, args.batch)
    extras = not args.headline_only

    eng = Engine(local_rank)
    eng.generate_synthetic(hi - lo, shard_lo=lo, n_global=n_global, seed=SEED, keep_aos=False)

    def batch_queries(err):
        return headline_queries(nat, make_query, B, world, err)

    side = [torch.cuda.Stream(), torch.cuda.Stream()]
    st = side[0].cuda_stream
    q_one = make_query(nat.M_CLT_DUAL_POINTER, pct, agg=nat.AVG, confidence_level=0.95, check_interval=10,
                       num_threads=4 * world, max_error_percent=e, clt_round0=CLT_ROUND0, clt_growth=CLT_GROWTH)
    plan_one = eng.plan(q_one)
    plan_sets = [[eng.plan(q) for q in batch_queries(e)] for _ in range(2)]
    natives = [Batch(ps) for ps in plan_sets]
    pipe = None

    with torch.cuda.stream(side[0]):
        if use_dist:
            width = max(p.totals_len for ps in plan_sets for p in ps)
            if not width:
                raise SystemExit("the bench query has no batched (totals) form on this shard")
            sbs = []
            # the collective: torch.distributed's RCCL by default; AQE_BENCH_NATIVE_COMM=1 takes the library's own
            # communicator (aqe_comm_*, what a C++ host uses) — its id travels through the torch group once
            native_comm = None
            if os.environ.get("AQE_BENCH_NATIVE_COMM") == "1" and not rehearsal:
                from approximatequeryengine_amd.distributed import comm_from_torch_group, native_all_reduce
                native_comm = comm_from_torch_group(eng)
                collective = {"backend": "aqe_comm (librccl, dlopen)", "ranks": native_comm.nranks}
            # AQE_BENCH_MAILBOX=1: the peer-mapped mailbox (aqe_mailbox_*: one single-workgroup launch per rank, no library
            # collective) — needs peer access between the ranks' GPUs and a batch of at most 4096 doubles
            mailbox = None
            if os.environ.get("AQE_BENCH_MAILBOX") == "1" and native_comm is None and B * width <= nat.MAILBOX_MAX_DOUBLES:
                from approximatequeryengine_amd.distributed import mailbox_all_reduce, mailbox_from_torch_group
                mailbox = mailbox_from_torch_group(eng)
                collective = {"backend": "aqe_mailbox (peer-mapped, one launch per rank)", "ranks": mailbox.nranks}
            for ps, nb in zip(plan_sets, natives):
                buf = torch.zeros(B, width, dtype=torch.float64, device="cuda")
                ar = (native_all_reduce(native_comm, st) if native_comm is not None else
                      mailbox_all_reduce(mailbox, st) if mailbox is not None else torch_all_reduce())
                sbs.append(ShardedBatch(ps, buf, ar, stream=st, batch=nb))
            pipe = PipelinedBatches(sbs)
            collectives_per_step = 1
            k_state = {"k": 0, "results": None}

            def step():           # sweeps of one batch; collective + replays of the previous one, whose results are then read
                done = pipe.enqueue()
                if done is not None:
                    k_state["results"] = done.fetch()

            def drain():
                done = pipe.flush()
                if done is not None:
                    k_state["results"] = done.fetch()
                return k_state["results"]

            def one():            # one batch, start to end (profiling / latency)
                sbs[0].enqueue()
        else:
            collectives_per_step = 0
            k_state = {"k": 0, "results": None}

            def step():           # one launch for a whole batch; the previous batch's results are read meanwhile
                k = k_state["k"]
                natives[k % 2].enqueue_all(side[k % 2].cuda_stream)
                if k > 0:
                    k_state["results"] = natives[(k - 1) % 2].fetch()
                k_state["k"] = k + 1

            def drain():
                k = k_state["k"]
                if k > 0:
                    k_state["results"] = natives[(k - 1) % 2].fetch()
                return k_state["results"]

            def one():
                natives[0].enqueue_all(st)

        def fence():
            if pipe is not None:
                pipe.flush()
            torch.cuda.synchronize()
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()

        # ---- reference answers of the batch's queries, each as a launch of its own (single GPU) ----
        for _ in range(3):
            step()
        firsts = drain()
        fence()

        # ---- roofline of the dominant kernel (k_sweep_lean_multi): ONE batch in flight, the launch's own begin/end
        #      timestamps (event pair attached to the dispatch), before the throughput loop ----
        prof_steps = 100
        own_h = fresh_stream()
        if not use_dist:
            def one():  # (the timed launches on a stream of their own: see CASES)
                natives[0].enqueue_all(own_h.cuda_stream)
        natives[0].set_profiling(True)
        ms_sum, ms_min, swept, wgs = 0.0, 1e9, 0, 0
        for _ in range(prof_steps):
            one()
            if use_dist:
                fence()
            else:
                natives[0].fetch()
            ms, swept, wgs = natives[0].launch_info()
            ms_sum += ms
            ms_min = min(ms_min, ms)
        natives[0].set_profiling(False)
        avg_launch_ms = ms_sum / prof_steps
        batch_kernel = nat.KERNEL_NAMES.get(plan_sets[0][0].last_kernel(), "?")  # k_sweep_lean_multi when every plan of the batch qualifies
        bytes_per_launch = 8.0 * swept
        torch.cuda.synchronize()
        if not use_dist:
            log_case(f"HEADLINE 10M: batch of {B} CLT e={e}% queries in ONE launch (executed bytes; see roofline for what is priced)", batch_kernel, prof_steps, prof_steps,
                     bytes_per_launch, 1e3 * avg_launch_ms)
        achieved = bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9

        # ---- the same batch with the XCD alignment of the groups turned off (AQE_MULTI_LAYOUT=packed): the layout in which
        #      the queries of a batch do not meet in a compute die's L2 and the fabric moves what the loads ask for ----
        packed = None
        if not use_dist and extras:
            os.environ["AQE_MULTI_LAYOUT"] = "packed"
            try:
                ps_x = [eng.plan(q) for q in batch_queries(e)]
                bx = Batch(ps_x)
                for _ in range(5):
                    bx.enqueue_all(st)
                    bx.fetch()
                bx.set_profiling(True)
                acc_x = []
                for _ in range(30):
                    bx.enqueue_all(st)
                    bx.fetch()
                    acc_x.append(bx.launch_info()[0])
                bx.set_profiling(False)
                x_ms = sum(acc_x) / len(acc_x)
                packed = {"avg_launch_us": 1e3 * x_ms, "achieved_GBps": bytes_per_launch / (x_ms * 1e-3) / 1e9,
                          "frac": bytes_per_launch / (x_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "workgroups": int(bx.launch_info(False)[2])}
                bx.close()
                for p_ in ps_x:
                    p_.close()
            finally:
                del os.environ["AQE_MULTI_LAYOUT"]

        # ---- a single query on the whole chip (k_sweep_lean), one in flight: launch time and closed loop ----
        single = None
        if not use_dist:
            own_s = fresh_stream()
            torch.cuda.synchronize()
            plan_one.set_profiling(True)
            acc = []
            for _ in range(prof_steps):
                plan_one.enqueue_all(own_s.cuda_stream)
                torch.cuda.synchronize()
                acc.append(sum(plan_one.launch_ms()))
            plan_one.set_profiling(False)
            lat = []
            for _ in range(50):
                t1 = time.perf_counter()
                plan_one.enqueue_all(st)
                r_one = plan_one.fetch(st)
                lat.append(time.perf_counter() - t1)
            lat.sort()
            acc.sort()
            s_us = 1e3 * sum(acc) / len(acc)
            single = {"kernel": nat.KERNEL_NAMES.get(plan_one.last_kernel(), "?"), "avg_launch_us": s_us, "min_launch_us": 1e3 * acc[0], "samples": int(r_one.visited),
                      "achieved_GBps": 8.0 * r_one.visited / (s_us * 1e-6) / 1e9, "frac": 8.0 * r_one.visited / (s_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                      "closed_loop_latency_us": {"p50": 1e6 * lat[len(lat) // 2], "min": 1e6 * lat[0]},
                      "aggregates_per_sec_one_in_flight": 1.0 / lat[len(lat) // 2]}
            log_case("10M: the bench query alone (CLT AVG e=0.01%, T=4)", single["kernel"], prof_steps, prof_steps, 8.0 * r_one.visited, s_us)
            if extras:
                single["closed_loop_c_host"] = c_host_closed_loop(rows)

        # ---- the other reading of "--e 0.01" (SURVEY §8d config 2): the FRACTION 0.01 = 1 percent.  That query
        #      converges after the first round, should_stop fires, and the reference's top-up (custom_bplus_db.cpp:
        #      1031-1040) supplies most of the sample: the early-termination path, also one launch per batch. ----
        other = None
        if not use_dist and e == 0.01 and extras:
            e2 = 1.0
            sets2 = [[eng.plan(q) for q in batch_queries(e2)] for _ in range(2)]
            nb2 = [Batch(ps) for ps in sets2]
            for b_ in nb2:
                b_.enqueue_all(st)
                r2 = b_.fetch()
            nb2[0].set_profiling(True)
            acc2 = []
            for _ in range(30):
                nb2[0].enqueue_all(st)
                nb2[0].fetch()
                acc2.append(nb2[0].launch_info())
            nb2[0].set_profiling(False)
            k2 = 200
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for k in range(k2):
                nb2[k % 2].enqueue_all(side[k % 2].cuda_stream)
                if k:
                    r2 = nb2[(k - 1) % 2].fetch()
            r2 = nb2[(k2 - 1) % 2].fetch()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t2
            l_us = 1e3 * sum(a[0] for a in acc2) / len(acc2)
            other = {"error_percent": e2, "aggregates_per_sec": B * k2 / dt2, "steps": k2, "queries_per_step": B,
                     "launch_us": l_us, "rows_swept_per_launch": int(acc2[0][1]), "workgroups": int(acc2[0][2]),
                     "achieved_GBps": 8.0 * acc2[0][1] / (l_us * 1e-6) / 1e9,
                     "result": {"value": r2[0].value, "ci": [r2[0].ci_lower, r2[0].ci_upper], "n": int(r2[0].n), "converged": int(r2[0].converged),
                                "rounds": int(r2[0].rounds), "topup_rows": int(r2[0].topup)},
                     "note": "every plan predicts its early stop from the table's head (cv and the error rule) and takes its HEAD form as its "
                             "group of the launch: the first rounds plus the reference's top-up (every 20th row) as one more slot; the group's "
                             "monitor judges the rounds, finds the sample short (DB.cpp:1032) and adds the top-up; had a query not stopped "
                             "there, fetch() would launch its remaining rounds"}
            for b_ in nb2:
                b_.close()
            for ps in sets2:
                for p in ps:
                    p.close()

        # ---- open loop by batch size (Q in {1, 16, 256}): one launch per batch, results fetched, one batch in flight ----
        open_loop = None
        if not use_dist and extras:
            import statistics
            open_loop = []
            for Qn in (1, 16, 256):
                qs_ = headline_queries(nat, make_query, Qn, 1, e)
                ps_ = [eng.plan(q) for q in qs_]
                b_ = Batch(ps_)
                for _ in range(3):
                    b_.enqueue_all(st)
                    b_.fetch()
                b_.set_profiling(True)
                ms_, lat_ = [], []
                for _ in range(30):
                    t1 = time.perf_counter()
                    b_.enqueue_all(st)
                    b_.fetch()
                    lat_.append(time.perf_counter() - t1)
                    ms_.append(b_.launch_info()[0])
                b_.set_profiling(False)
                med = statistics.median(lat_)
                open_loop.append({"queries_per_launch": Qn, "launch_us_median": 1e3 * statistics.median(ms_), "closed_loop_us_median": 1e6 * med,
                                  "aggregates_per_sec": Qn / med, "workgroups": int(b_.launch_info(False)[2])})
                b_.close()
                for p_ in ps_:
                    p_.close()

        # ---- the timed region: K steps of L launches each, every result fetched ----
        L = max(1, args.launches_per_step)
        for _ in range(max(args.warmup, 1) * L):
            step()
        drain()
        fence()
        k_state["k"] = 0
        t0 = time.perf_counter()
        for _ in range(args.steps * L):
            step()
        lasts = drain()
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        last = lasts[0]
        agree = all((a.n, a.visited, a.rounds, a.converged) == (b_.n, b_.visited, b_.rounds, b_.converged) and a.value == b_.value
                    for a, b_ in zip(firsts[: len(lasts)], lasts)) if not use_dist else True
        assert agree, "results of the timed loop differ from the first execution's"

    # HBM-side traffic of the sweep kernel: a PMC measurement (FETCH_SIZE + WRITE_SIZE in their own rocprofv3 passes,
    # gfx950 correction calibrated on a known 80 MB scan) cannot be taken inside this process; the committed value in
    # profiles/ is reported only when it was measured on these very sources and this workload.
    traffic, traffic_src, packed_traffic = None, None, None
    try:
        if not use_dist and rows == ROWS_PER_GPU and e == 0.01:
            doc = json.loads(PMC_FILE.read_text())
            if doc.get("source_hash") == source_hash() and doc.get("batch") == B:
                traffic = doc["batch_traffic_bytes_per_launch"]
                packed_traffic = doc.get("batch_packed_traffic_bytes_per_launch")
                traffic_src = f"{PMC_FILE.relative_to(ROOT)} (rocprofv3 --pmc, bytes per launch of {B} queries, sources {doc['source_hash']})"
            else:
                traffic_src = f"{PMC_FILE.relative_to(ROOT)} was measured on other sources / another batch size: not reported"
    except Exception:
        traffic = None

    if rank == 0:
        qs0 = batch_queries(e)
        # algorithmic bytes: 8 B per row the SUM / AVG queries sample (COUNT is metadata, SURVEY 8d: 0 B although the
        # reference — and this launch — run the sampler for it); executed: every load of every query
        per_q = [int(r.visited) for r in firsts[:B]] if not use_dist else [int(swept // B)] * B
        alg_bytes = 8.0 * sum(v for q_, v in zip(qs0, per_q) if q_.agg != nat.COUNT)
        try:
            uniq_bytes = 8.0 * unique_sampled_rows(nat, qs0, n_global, lo, hi)
        except Exception:
            uniq_bytes = 8.0 * (hi - lo)  # every row of the shard at most once
        roof = headline_roofline(batch_kernel, 1e3 * avg_launch_ms, alg_bytes, uniq_bytes, traffic, bytes_per_launch, traffic_src)
        roof.update({"min_launch_us": 1e3 * ms_min, "queries_per_launch": B, "workgroups": int(wgs), "launches_timed": prof_steps,
                     "executed_bytes_per_launch": bytes_per_launch, "packed_layout": packed,
                     "l2_rate_guide_GBps": [16800, 18800],
                     "note": "the queries of a batch sample the same 20 % of the rows (the reference's samplers are deterministic in (N, pct)) and the "
                             "launch is XCD-aware (workgroup k of every group on compute die k mod 8, sweeping the k-th share of its query's tiles), so "
                             "they read them out of that die's L2 together: `l2_rate_GBps` is the rate of all executed loads (the guide measures 16.8-18.8 "
                             "TB/s for L2-served reads) and is NOT a roofline figure; `achieved` prices only the bytes that cross the fabric.  HBM proper: "
                             "`roofline_hbm` (tables far larger than the 256 MiB Infinity Cache)"})
        info = eng.info()
        Lps = max(1, args.launches_per_step)
        rep = {
            "metric": "aggregates/sec (10M-row-per-GPU APPROX AVG/SUM/COUNT with 95% CI, CLT --e 0.01) + achieved HBM GB/s",
            "value": B * args.steps * Lps / dt,
            "unit": "aggregates/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"configs[1]: 10M-row APPROX AVG/SUM/COUNT, CLT --e 0.01 percent (never converges: full 20% dual-pointer sweep, "
                            f"should_stop armed), table resident in HBM; step = {Lps} launches x {B} different queries per launch, every result fetched",
                "rows_per_gpu": rows, "global_rows": n_global, "sample_percent": pct, "error_percent": e,
                "pointers": "4,6,...,16 per GPU (x n_gpus)", "samples_per_query_per_gpu": int(swept // B),
                "clt_round0": CLT_ROUND0, "clt_growth": CLT_GROWTH, "launches_per_step": Lps * (1 if not use_dist else 2),
                "queries_per_launch": B, "queries_per_step": B * Lps, "batches_in_flight": 2, "fetched_every_step": True,
                "collectives_per_step": collectives_per_step * Lps, "collective": collective,
                "unit_definition": "one completed query (value + 95 % interval) over the GLOBAL table; with N GPUs every rank sweeps its 10M-row region of it",
                "cadence_note": "rounds of 4096 x 4^r rows per pointer, not the reference's check every 10 samples: immaterial at e = 0.01 % (never "
                                "converges); at e = 1 % the geometric schedule stops at 20 480 rows per pointer where the reference's T = 2 run stops at 12 390",
            },
            "region_aggregates_per_sec": world * B * args.steps * Lps / dt,
            "rows_swept_per_sec": float(swept) * world * args.steps * Lps / dt,
            "timed_seconds": dt,
            "result": {"value": last.value, "ci": [last.ci_lower, last.ci_upper], "n": int(last.n), "converged": int(last.converged),
                       "rounds": int(last.rounds), "same_as_first_execution": bool(agree)},
            "roofline": roof,
            "single_query": single,
            "open_loop_by_batch_size": open_loop,
            "early_termination_reading": other,
            "table": {"hbm_bytes": int(info.hbm_bytes), "view_bytes": int(info.view_bytes), "n_views": int(info.n_views),
                      "view_evictions": int(info.view_evictions), "view_fallbacks": int(info.view_fallbacks)},
            "source_hash": source_hash(),
        }
        if packed is not None:
            packed["traffic"] = packed_traffic
            packed["note"] = ("AQE_MULTI_LAYOUT=packed: no XCD alignment of the groups, the queries of a batch do not meet in a die's L2; "
                              "traffic (PMC) is then about the algorithmic bytes: Infinity-Cache bandwidth of a 10 M-row table")
        if extras and not use_dist:
            if not args.no_configs:
                try:
                    rep["configs"] = run_configs(eng, nat, make_query, st, args.max_config_rows, Batch)
                    # HBM proper: the streaming cases on the largest tables measured (1 B rows = 8 GB column, then 100 M)
                    # (single-query launches and the batch over disjoint key ranges: nothing shared between queries in a cache)
                    big = [c for c in rep["configs"] if c.get("frac") and c.get("algorithmic_bytes", 0) >= 1.0e9
                           and ("queries_per_launch" not in c or "DISJOINT" in c["config"])]
                    big.sort(key=lambda c: -c["algorithmic_bytes"])
                    rep["roofline_hbm"] = [{"case": c["config"][:60], "kernel": c["kernel"], "avg_launch_us": c["kernel_us"],
                                            "algorithmic_bytes": c["algorithmic_bytes"], "achieved": c["achieved_GBps"], "frac": c["frac"]} for c in big]
                except Exception as ex:
                    rep["configs"] = [{"error": repr(ex)}]
            try:
                eng.release_table()
                rep["cold"] = cold_numbers(Engine, nat, make_query, rows, e)
                if args.max_config_rows >= 100_000_000:  # config 3's staging leg at size: a 3.2 GB file
                    rep["cold_100M"] = cold_numbers(Engine, nat, make_query, 100_000_000, e)
            except Exception as ex:
                rep["cold"] = {"error": repr(ex)}
            if not args.no_cpu_baseline:
                try:
                    cb = cpu_baseline(rows, e, args.cpu_sample_rows)
                    rep["cpu_baseline"] = cb.get("reference", cb["port"])
                    rep["cpu_baseline_all"] = cb
                except Exception as ex:  # the bench line must still print
                    rep["cpu_baseline"] = {"value": None, "unit": "aggregates/sec", "cores": 0, "kind": "port", "sample": f"failed: {ex!r}"}
        rep["cases"] = CASES
        write_report(rep)
        print(json.dumps(compact_line(rep)), flush=True)

    for nb in natives:
        nb.close()
    for ps in plan_sets:
        for p in ps:
            p.close()
    plan_one.close()
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
