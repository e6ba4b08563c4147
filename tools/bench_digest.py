"""Dev helper: short digest of a bench.py JSON line (file with the line last)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"]), "ms/step", round(d["ms_per_step"], 4), "| roofline", {k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_us", "min_launch_us", "workgroups", "traffic")})
print("single", json.dumps(d.get("single_query")))
print("early", json.dumps(d.get("early_termination_reading"))[:420])
for c in d.get("configs", []) or []:
    print(json.dumps({k: (round(c[k], 3) if isinstance(c.get(k), float) else c.get(k)) for k in ("config", "samples", "kernel_us", "achieved_GBps", "frac", "closed_loop_us_p50", "launches_per_query", "error") if k in c}))
print("cold", json.dumps(d.get("cold"))[:1500])
cb = d.get("cpu_baseline_all") or {}
for k, v in cb.items():
    print("cpu", k, json.dumps(v)[:700])
