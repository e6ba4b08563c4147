#!/bin/bash
# Round-2 profiles: kernel stats of the bench, of the 100 M / 1 B-row configurations, of the grouped reductions; PMC passes.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out/prof_r2
rm -rf $O; mkdir -p $O
echo "--- kernel trace + stats: bench headline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python3 bench.py --headline-only --steps 200 --warmup 20 > $O/bench_line.txt 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
head -6 $O/bench/b_kernel_stats.csv | cut -c1-220
echo "--- kernel trace + stats: 10 M, 100 M and 1 B rows"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/configs -o c -- python3 tools/bench_configs.py 10000000,100000000,1000000000 > $O/configs_lines.txt 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }
head -8 $O/configs/c_kernel_stats.csv | cut -c1-220
echo "--- kernel trace + stats: grouped reductions"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/group -o g -- python3 tools/group_time.py > $O/group_lines.txt 2> $O/group.err || { tail -5 $O/group.err; exit 1; }
cat $O/group_lines.txt
echo "--- PMC FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 tools/pmc_probe.py > /dev/null 2> $O/pmc_fetch.err || { tail -5 $O/pmc_fetch.err; exit 1; }
echo "--- PMC WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 tools/pmc_probe.py > /dev/null 2> $O/pmc_write.err || { tail -5 $O/pmc_write.err; exit 1; }
echo "--- PMC FETCH_SIZE, disjoint batch on 320 M rows"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_disjoint -o d -- python3 tools/pmc_probe_disjoint.py > /dev/null 2> $O/pmc_disjoint.err || { tail -5 $O/pmc_disjoint.err; exit 1; }
python tools/pmc_summarize.py $O/pmc_fetch $O/pmc_write $O/round2_pmc_raw.json 32 $O/pmc_disjoint
# keep what travels back small: traces of the bench only
rm -f $O/configs/c_kernel_trace.csv $O/pmc_fetch/f_kernel_trace.csv $O/pmc_write/w_kernel_trace.csv $O/pmc_disjoint/d_kernel_trace.csv
du -sh $O
