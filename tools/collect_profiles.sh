#!/bin/bash
# Dev helper: copy what tools/gpu_final.sh brought back (gpurun_out/) into profiles/ under its round-2 names.
cd "$(dirname "$0")/.."
P=gpurun_out/prof_r2
cp $P/bench/b_kernel_stats.csv profiles/round2_bench_kernel_stats.csv
cp $P/bench/b_domain_stats.csv profiles/round2_bench_domain_stats.csv
cp $P/configs/c_kernel_stats.csv profiles/round2_configs_kernel_stats.csv
cp $P/configs_lines.txt profiles/round2_configs.jsonl
cp $P/group/g_kernel_stats.csv profiles/round2_group_kernel_stats.csv
cp $P/group_lines.txt profiles/round2_group_wall.txt
cp $P/round2_pmc_raw.json profiles/round2_pmc_raw.json
cp gpurun_out/bench_r2_final.txt profiles/round2_bench_line.json
[ -s gpurun_out/lean_timeline.txt ] && { echo "# tools/stamp_lean.py clt (library built with -DAQE_LEAN_STAMPS): s_memrealtime marks inside k_sweep_lean, bench query (10 M rows, e = 0.01 %: 5 rounds, 32 MB), microseconds from the first wave's start; the stamps cost a few hundred ns themselves, stamped launches run ~2 us longer than unstamped ones"; cat gpurun_out/lean_timeline.txt; } > profiles/round2_lean_timeline.txt
python tools/multi_timeline.py $P/bench/b_kernel_trace.csv > profiles/round2_multi_timeline.txt
head -3 profiles/round2_bench_kernel_stats.csv | cut -c1-190
python -c "
import json,bench
d=json.load(open('profiles/round2_pmc_raw.json')); print('hash ok', d['source_hash']==bench.source_hash(), d['source_hash'])
b=json.loads(open('profiles/round2_bench_line.json').read().strip().splitlines()[-1]); print(b['value'], b['roofline']['avg_launch_us'], b['steps'])
"
