"""Turn the two rocprofv3 --pmc passes over tools/pmc_probe.py into profiles/round1_pmc_raw.json.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o f -- python3 tools/pmc_probe.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o w -- python3 tools/pmc_probe.py
    python tools/pmc_summarize.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/round1_pmc_raw.json

Per kernel (name + grid size): median counter value over its launches, in the counter's own unit (KB).  FETCH_SIZE on
gfx950 under-reports coalesced streaming reads (MI355X_MICROARCH.md, HBM recipe): the correction factor is calibrated
on this library's own exact scan of the 10 M-row amount column (known traffic: 80 MB), then applied to the sweep."""
import csv, glob, json, os, re, statistics, sys

def medians(d):
    out = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(r"::(k_\w+)\(", r["Kernel_Name"])
            if not m:
                continue
            out.setdefault((r["Counter_Name"], f'{m.group(1)} grid_threads={r["Grid_Size"]}'), []).append(float(r["Counter_Value"]))
    res = {}
    for (counter, kern), v in out.items():
        res.setdefault(counter, {})[kern] = {"launches": len(v), "median_KB": statistics.median(v)}
    return res

fetch_dir, write_dir, dst = sys.argv[1:4]
doc = {}
doc.update(medians(fetch_dir))
doc.update(medians(write_dir))
F, W = doc["FETCH_SIZE"], doc.get("WRITE_SIZE", {})
scan = max((k for k in F if k.startswith("k_round ")), key=lambda k: F[k]["median_KB"])  # the exact scan is the biggest k_round
known = 80_000_000
corr = known / (F[scan]["median_KB"] * 1024.0)
doc["calibration"] = {"known_bytes": known, "kernel": f"{scan} (exact scan of the 10M-row amount column)", "fetch_correction": corr}
sweep = next(k for k in F if k.startswith("k_sweep_persist "))
doc["k_sweep_persist_traffic_bytes_per_launch"] = F[sweep]["median_KB"] * 1024.0 * corr + W.get(sweep, {"median_KB": 0.0})["median_KB"] * 1024.0
json.dump(doc, open(dst, "w"), indent=1)
print(json.dumps({k: doc[k] for k in ("calibration", "k_sweep_persist_traffic_bytes_per_launch")}, indent=1))
