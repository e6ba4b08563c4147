"""The bench's stdout contract (CPU only, canned numbers): the LAST line is one compact JSON object the driver's
~8 KB window always holds whole, and its roofline fraction cannot exceed 1."""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def canned_report(traffic=197.0e6):
    alg = 8.0 * 4_000_000 * 22  # 22 of 32 queries are SUM / AVG
    roof = bench.headline_roofline("k_sweep_lean_multi", 53.7, alg, 8.0 * 4_400_000, traffic, 8.0 * 4_000_000 * 32, "profiles/x.json")
    roof.update({"note": "n" * 900, "packed_layout": {"avg_launch_us": 134.0, "note": "p" * 300}})
    cfgs = [{"config": "1B exact SUM (full scan)" + " padded" * 20, "kernel": "k_sweep_lean", "kernel_us": 1227.0, "algorithmic_bytes": 8.0e9,
             "achieved_GBps": 6520.0, "frac": 0.815, "closed_loop_us_p50": 1240.0} for _ in range(30)]
    return {
        "metric": "aggregates/sec (10M-row-per-GPU APPROX AVG/SUM/COUNT with 95% CI, CLT --e 0.01) + achieved HBM GB/s",
        "value": 612345.678, "unit": "aggregates/sec", "n_gpus": 1, "steps": 20, "warmup": 5, "ms_per_step": 5.2261234,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "w" * 400, "rows_per_gpu": 10_000_000, "global_rows": 10_000_000, "queries_per_step": 3200,
                   "launches_per_step": 100, "queries_per_launch": 32, "samples_per_query_per_gpu": 4_000_000,
                   "collectives_per_step": 0, "collective": None, "unit_definition": "u" * 300, "cadence_note": "c" * 300},
        "roofline": roof,
        "roofline_hbm": [{"case": c["config"][:60], "kernel": c["kernel"], "avg_launch_us": c["kernel_us"], "algorithmic_bytes": c["algorithmic_bytes"],
                          "achieved": c["achieved_GBps"], "frac": c["frac"]} for c in cfgs],
        "single_query": {"kernel": "k_sweep_lean", "avg_launch_us": 9.91, "frac": 0.4036, "closed_loop_latency_us": {"p50": 16.5, "min": 15.0}},
        "configs": cfgs,
        "cold": {"note": "x" * 2000},
        "cpu_baseline": {"value": 0.024812, "unit": "aggregates/sec", "cores": 4, "host_cores": 256, "kind": "reference", "sample": "s" * 600,
                         "bounded_run": {"rows": 2_000_000}},
        "cpu_baseline_all": {"port": {"sample": "y" * 3000}},
    }


def test_compact_line_fits_the_driver_window_and_keeps_the_contract():
    line = bench.compact_line(canned_report())
    text = json.dumps(line)
    assert len(text) < 3000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "roofline_hbm", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"] and "model" not in line["config"]
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_us"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "host_cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert len(line["roofline_hbm"]) <= 4 and line["roofline_hbm"][0]["frac"] == 0.815
    assert json.loads(text) == line


def test_compact_line_of_a_multi_gpu_run_says_how_to_read_its_value():
    rep = canned_report()
    rep.update({"n_gpus": 8, "rows_swept_per_sec": 1.9e13, "region_aggregates_per_sec": 4.8e6})
    rep["config"].update({"global_rows": 80_000_000, "collectives_per_step": 100, "collective": {"backend": "nccl", "ranks": 8}})
    line = bench.compact_line(rep)
    assert len(json.dumps(line)) < 3000
    assert line["n_gpus"] == 8 and line["value"] == 612345.7  # global queries per second, not multiplied by the ranks
    assert line["scaling_detail"]["rows_swept_per_sec"] == 1.9e13 and "constant in N" in line["scaling_detail"]["note"]
    assert "scaling_detail" not in bench.compact_line(canned_report())


def test_headline_roofline_fraction_is_a_fraction():
    alg, uniq, executed = 8.0 * 4e6 * 22, 8.0 * 4.4e6, 8.0 * 4e6 * 32
    with_pmc = bench.headline_roofline("k", 53.7, alg, uniq, 197.0e6, executed)
    assert with_pmc["basis"] == "pmc_traffic" and abs(with_pmc["achieved"] - 197.0e6 / 53.7e-6 / 1e9) < 1e-6
    assert 0.0 < with_pmc["frac"] <= 1.0 and with_pmc["l2_rate_GBps"] > bench.HBM_PEAK_GBPS  # the L2-served rate is kept apart
    without = bench.headline_roofline("k", 53.7, alg, uniq, None, executed)
    assert without["basis"] == "unique_bytes" and without["bytes_priced"] == uniq and without["frac"] <= 1.0
    # traffic above the algorithmic bytes (wasted re-reads) is never credited: the numerator is capped at the algorithmic bytes
    wasteful = bench.headline_roofline("k", 53.7, 32.0e6, 32.0e6, 80.0e6, 32.0e6)
    assert wasteful["bytes_priced"] == 32.0e6
    # a cache-resident table could in principle be read faster than HBM delivers: still reported as at most 1
    assert bench.headline_roofline("k", 1.0, 32.0e6, 32.0e6, None, 32.0e6)["frac"] == 1.0


def test_unique_sampled_rows_matches_a_direct_count():
    import numpy as np
    from approximatequeryengine_amd import _native as nat
    from approximatequeryengine_amd.engine import make_query
    n = 200_000
    qs = bench.headline_queries(nat, make_query, 8, 1, 0.01)
    got = bench.unique_sampled_rows(nat, qs, n)
    rows = set()
    for q in qs:
        _, rounds, samples = nat.plan_families(q, n, 0, n, 0)
        for r in range(rounds):
            for f in nat.plan_families(q, n, 0, n, r)[0]:
                for o in range(f.ord_lo, f.ord_hi):
                    rows.add(f.row0 + (o // f.seg_len) * f.pitch + (o % f.seg_len) * f.step)
                if f.flags & nat.F_PAIR:
                    for o in range(f.ord_lo_b, f.ord_hi_b):
                        rows.add(f.row0_b + o * f.step)
    assert got == len(rows)
    assert 0.2 * n <= got <= n  # every query samples 20 % twice over (fast + slow pointers)
