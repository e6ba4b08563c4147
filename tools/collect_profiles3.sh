#!/bin/bash
# Dev helper: copy what tools/gpu_final3.sh brought back (gpurun_out/prof_r3) into profiles/ under its round-3 names.
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r3
cp $P/round3_pmc_raw.json profiles/round3_pmc_raw.json
tail -n 1 $P/bench_line.json > profiles/round3_bench_line.json
cp $P/bench_report.json profiles/round3_bench_report.json
cp $P/round3_configs_cases.csv profiles/round3_configs_cases.csv
cp $P/bench/b_kernel_stats.csv profiles/round3_bench_kernel_stats.csv
cp $P/bench/b_domain_stats.csv profiles/round3_bench_domain_stats.csv 2>/dev/null
{ echo "# tools/group_time.py on 10 M rows: wall time per aqe_reduce_grouped call through Python (column 1 = region: 4 keys, 2 = product_id: 100 keys);"; echo "# first six lines: the fused single-launch form (default); last six: AQE_GROUP_UNFUSED=1 (sweep + finish launch + stream wait)"; cat $P/group_wall.txt; } > profiles/round3_group_wall.txt
{ echo "# tools/ab_ablate.sh run: the 10 M-row bench query (k_sweep_lean, 4 M samples) with one stage of the launch compiled out (dispatch begin -> end, 300 launches, two passes)."; echo "# base = the product; nojudge = rules + estimate skipped; nostore = everything computed, one 8-byte store instead of state + result;"; echo "# nofold = the folding workgroup returns right after the last ticket; noticket = workgroups return after the sweep (the sweep is then dead code: an empty launch);"; echo "# fold0..3 = the folding workgroup returns before fetching the partials / after summing them per round / after the barrier / after the scan of the rounds (fold3 ends in ONE 8-byte store: a launch that ends in a store waits ~1 us for it to be acknowledged, so fold2 -> fold3 is mostly that, not the scan);"; echo "# nosweep = no loads: all 256 workgroups reach the tickets at once"; cat $P/lean_ablation.txt; } > profiles/round3_lean_ablation.txt
{ echo "# tools/stamp_lean.py clt (library built with -DAQE_LEAN_STAMPS): s_memrealtime marks inside k_sweep_lean, bench query (10 M rows, e = 0.01 %: 5 rounds, 32 MB), microseconds from the first wave's start; stamped launches run ~2 us longer than unstamped ones"; cat $P/lean_timeline.txt; } > profiles/round3_lean_timeline.txt
{ echo "# tools/mailbox_time.py: one peer-mapped all-reduce (aqe_mailbox_all_reduce_sum), G contexts of one process on ONE GPU (no xGMI hop in it: the floor);"; echo "# last two lines: the bench's N > 1 path rehearsed with 2 ranks sharing this GPU (AQE_BENCH_REHEARSAL=1; never a reported number): gloo vs the mailbox as the collective"; cat $P/mailbox.txt; } > profiles/round3_mailbox.txt
tail -3 $P/pytest.log > profiles/round3_pytest_gpu.txt
python - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
d = json.load(open("profiles/round3_pmc_raw.json")); print("pmc hash matches the tree:", d["source_hash"] == bench.source_hash(), d["source_hash"])
b = json.loads(open("profiles/round3_bench_line.json").read()); print(len(open("profiles/round3_bench_line.json").read()), "bytes;", b["value"], b["roofline"]["frac"], b["roofline"]["basis"])
PY
