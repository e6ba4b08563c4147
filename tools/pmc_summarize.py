"""Turn the two rocprofv3 --pmc passes over tools/pmc_probe.py into profiles/round3_pmc_raw.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 tools/pmc_probe.py
    python tools/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/round2_pmc_raw.json [batch]

Per kernel (name + grid size): median counter value over its launches, in the counter's own unit (KB).  FETCH_SIZE on
gfx950 under-reports coalesced streaming reads (MI355X_MICROARCH.md, HBM recipe): the correction factor is calibrated
on this library's own exact scan of the 10 M-row amount column (known traffic: 80 MB), then applied to the sweeps.
The file carries the hash of the library sources it was measured on: bench.py reports the traffic only for those."""
import csv, glob, json, os, re, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

def medians(d):
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"::(k_\w+(?:<[^>]*>)?)\(", r["Kernel_Name"])
            if not m:
                continue
            out.setdefault((r["Counter_Name"], f'{m.group(1)} grid_threads={r["Grid_Size"]}'), []).append(float(r["Counter_Value"]))
    res = {}
    for (counter, kern), v in out.items():
        res.setdefault(counter, {})[kern] = {"launches": len(v), "median_KB": statistics.median(v)}
    return res

fetch_dir, write_dir, dst = sys.argv[1:4]
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 32
doc = {"source_hash": bench.source_hash(), "batch": batch}
doc.update(medians(fetch_dir))
doc.update(medians(write_dir))
F, W = doc["FETCH_SIZE"], doc.get("WRITE_SIZE", {})
def named(k, name):  # "k_round<false> grid_threads=..." is kernel k_round
    return re.split(r"[<\s]", k, maxsplit=1)[0] == name
scan = max((k for k in F if named(k, "k_round")), key=lambda k: F[k]["median_KB"])  # the exact scan is the biggest k_round
known = 80_000_000
corr = known / (F[scan]["median_KB"] * 1024.0)
doc["calibration"] = {"known_bytes": known, "kernel": f"{scan} (exact scan of the 10M-row amount column)", "fetch_correction": corr}
def traffic(prefix):
    ks = [k for k in F if named(k, prefix)]
    if not ks:
        return None
    k = max(ks, key=lambda k: F[k]["median_KB"])
    return F[k]["median_KB"] * 1024.0 * corr + W.get(k, {"median_KB": 0.0})["median_KB"] * 1024.0
# the bench batch ran in two layouts, same grid: XCD-aligned (the default, first) and packed (second) — told apart by the
# order of the dispatches in the trace (tools/pmc_probe.py runs 20 launches of one, then 20 of the other)
def multi_halves(d):
    vals = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if re.search(r"k_sweep_(lean_)?multi", r["Kernel_Name"]) and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE")]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        vals = [float(r["Counter_Value"]) for r in rows]
    h = len(vals) // 2
    return (statistics.median(vals[:h]), statistics.median(vals[h:])) if h else (None, None)
fa, fp = multi_halves(fetch_dir)
wa, wp = multi_halves(write_dir)
# (the batch's one launch: k_sweep_lean_multi when every plan qualifies for the lean form, else k_sweep_multi)
doc["batch_kernel"] = next((re.split(r"[<\s]", k, maxsplit=1)[0] for k in F if re.match(r"k_sweep_(lean_)?multi", k)), None)
doc["batch_traffic_bytes_per_launch"] = fa * 1024.0 * corr + (wa or 0.0) * 1024.0 if fa else None
doc["batch_packed_traffic_bytes_per_launch"] = fp * 1024.0 * corr + (wp or 0.0) * 1024.0 if fp else None
doc["batch_raw_KB"] = {"xcd_aligned (default)": {"FETCH_SIZE": fa, "WRITE_SIZE": wa}, "packed": {"FETCH_SIZE": fp, "WRITE_SIZE": wp}}
# k_sweep_lean<false, false> on the full grid serves two probes in dispatch order: 40 launches of the CLT bench query (32 MB sampled),
# then 20 exact scans (80 MB): told apart by order
def lean_groups(d):
    vals = []
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(path)) if re.search(r"::k_sweep_lean<", r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE" and r["Grid_Size"] == "262144"]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        vals = [float(r["Counter_Value"]) for r in rows]
    return vals
lv = lean_groups(fetch_dir)
if len(lv) >= 60:
    doc["single_query_traffic_bytes_per_launch"] = statistics.median(lv[:40]) * 1024.0 * corr
    doc["exact_scan_lean_traffic_bytes_per_launch"] = statistics.median(lv[40:60]) * 1024.0 * corr
else:
    doc["single_query_traffic_bytes_per_launch"] = traffic("k_sweep_lean") or traffic("k_sweep_persist")
doc["k_indexed_traffic_bytes_per_launch"] = traffic("k_indexed")
# grouped sweeps (tools/pmc_probe.py runs, per key column, the reference's 10 % rowid sample — 1 M sampled rows, 12 B each: amount + key —
# and the exact scan — 10 M rows): per template instance, the smaller median is the sample, the larger the scan
for inst in ("k_grouped<true>", "k_grouped<false>"):
    ks = sorted((k for k in F if k.startswith(inst + " ")), key=lambda k: F[k]["median_KB"])
    for k in ks:
        doc.setdefault("k_grouped_traffic_bytes_per_launch", {})[k] = F[k]["median_KB"] * 1024.0 * corr + W.get(k, {"median_KB": 0.0})["median_KB"] * 1024.0
if len(sys.argv) > 5:  # a third pass: the batch of 32 exact scans over disjoint key ranges of a 320 M-row table (tools/pmc_probe_disjoint.py)
    D = medians(sys.argv[5])["FETCH_SIZE"]
    k = max((k for k in D if named(k, "k_sweep_multi") or named(k, "k_sweep_lean_multi")), key=lambda k: D[k]["median_KB"])
    doc["batch_disjoint_320M"] = {"kernel": k, "launches": D[k]["launches"], "fetch_raw_KB": D[k]["median_KB"],
                                          "traffic_bytes_per_launch": D[k]["median_KB"] * 1024.0 * corr, "algorithmic_bytes_per_launch": 8.0 * 320_000_000}
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in doc.items() if not isinstance(v, dict) or k == "calibration"}, indent=1))
