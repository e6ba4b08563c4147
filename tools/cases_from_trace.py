"""Per-case kernel rows out of ONE `rocprofv3 --kernel-trace` of bench.py.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r3/bench -o b -- python3 bench.py --steps 20 --warmup 5
    python tools/cases_from_trace.py gpurun_out/prof_r3/bench gpurun_out/bench_report.json profiles/round3_configs_cases.csv

bench.py runs the timed launches of every case on a stream of their own and records how many launches that were
(bench_report.json: "cases").  Here the trace's sweep-kernel dispatches are grouped by Stream_Id; walking the groups in
time order, a case is matched with the first unused stream whose dispatch count is exactly the case's.  Output: one row
per (case, kernel template, grid) — calls, avg / min / max ns — and per case the time per query (all its launches summed),
its algorithmic bytes and the fraction of the 8 TB/s HBM roofline, next to the figure bench.py took from HIP events."""
import csv, glob, json, os, re, statistics, sys

trace_dir, report, dst = sys.argv[1:4]
rep = json.load(open(report))
cases = rep["cases"]
SWEEP = re.compile(r"::(k_round|k_sweep_lean_multi|k_sweep_lean|k_sweep_multi|k_sweep_persist|k_indexed|k_permuted)(<[^>]*>)?\(")
rows = []
for path in glob.glob(os.path.join(trace_dir, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        m = SWEEP.search(r["Kernel_Name"])
        if m:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], m.group(1) + (m.group(2) or ""),
                         int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort()
by_stream, first_seen = {}, {}
for t0, t1, sid, kern, grid in rows:
    by_stream.setdefault(sid, []).append((t0, t1, kern, grid))
    first_seen.setdefault(sid, t0)
order = sorted(by_stream, key=lambda s: first_seen[s])
used, out, k = set(), [], 0
for c in cases:
    hit = None
    while k < len(order):
        sid = order[k]
        k += 1
        if sid not in used and len(by_stream[sid]) == c["launches"]:
            hit = sid
            break
    if hit is None:
        raise SystemExit(f"no stream of {c['launches']} sweep dispatches left for case {c['case']!r}")
    used.add(hit)
    d = by_stream[hit]
    L = c["launches"] // max(c["queries"], 1)
    per_query = [sum(t1 - t0 for t0, t1, _, _ in d[i * L:(i + 1) * L]) for i in range(c["queries"])] if L * c["queries"] == c["launches"] else None
    groups = {}
    for t0, t1, kern, grid in d:
        groups.setdefault((kern, grid), []).append(t1 - t0)
    q_ns = statistics.median(per_query) if per_query else sum(t1 - t0 for t0, t1, _, _ in d) / max(c["queries"], 1)
    for (kern, grid), v in groups.items():
        out.append({"case": c["case"], "kernel": kern, "workgroups": grid, "calls": len(v), "avg_ns": round(sum(v) / len(v)), "min_ns": min(v), "max_ns": max(v),
                    "launches_per_query": L, "query_ns_median": round(q_ns), "algorithmic_bytes_per_query": c["algorithmic_bytes_per_query"],
                    "GBps": round(c["algorithmic_bytes_per_query"] / q_ns, 1), "frac_of_8TBps": round(c["algorithmic_bytes_per_query"] / q_ns / 8000.0, 4),
                    "bench_event_us_per_query": round(c["event_us_per_query"], 2)})
with open(dst, "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(out[0].keys()))
    w.writeheader()
    w.writerows(out)
print(f"{len(cases)} cases, {len(out)} rows -> {dst}")
for o in out:
    print("%-90s %-26s wg %5d calls %4d avg %9.2f us  query %9.2f us  frac %.3f  (events: %.2f us)" % (o["case"][:90], o["kernel"], o["workgroups"], o["calls"], o["avg_ns"] / 1e3,
                                                                                                      o["query_ns_median"] / 1e3, o["frac_of_8TBps"], o["bench_event_us_per_query"]))
