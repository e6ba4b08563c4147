#!/bin/bash
# Round-3 evidence in ONE gpurun call: GPU tests, the unprofiled bench (the driver's command), one kernel trace of the same
# command split per case, the PMC passes (counters in their own runs), the lean launch's ablation and in-kernel timeline.
#   tools/gpu_final3.sh prep   (here: builds the stamped and ablation libraries)     tools/gpu_final3.sh   (GPU box)
cd "$(dirname "$0")/.."
if [ "$1" = prep ]; then
  tools/ab_libs.sh stamps "-DAQE_LEAN_STAMPS" 2>&1 | grep -v warning | tail -1
  ABL=1 tools/ab_ablate.sh build
  exit 0
fi
export TMPDIR=/tmp
O=gpurun_out/prof_r3
rm -rf $O; mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; rc=$?; tail -3 $O/pytest.log; [ $rc -eq 0 ] || exit 1
echo "--- PMC FETCH_SIZE / WRITE_SIZE / disjoint batch"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 tools/pmc_probe.py > /dev/null 2> $O/pmc_fetch.err || { tail -5 $O/pmc_fetch.err; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 tools/pmc_probe.py > /dev/null 2> $O/pmc_write.err || { tail -5 $O/pmc_write.err; exit 1; }
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_disjoint -o d -- python3 tools/pmc_probe_disjoint.py > /dev/null 2> $O/pmc_disjoint.err || { tail -5 $O/pmc_disjoint.err; exit 1; }
python tools/pmc_summarize.py $O/pmc_fetch $O/pmc_write $O/round3_pmc_raw.json 32 $O/pmc_disjoint | tail -12
cp $O/round3_pmc_raw.json profiles/round3_pmc_raw.json   # (measured a minute ago on these very sources: the bench reports it as roofline.traffic)
echo "--- the bench, unprofiled (the driver's command)"
( time python bench.py --steps 20 --warmup 5 ) > $O/bench_line.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cp gpurun_out/bench_report.json $O/bench_report.json
tail -c 2600 $O/bench_line.json; grep real $O/bench.err
echo "--- kernel trace + stats of the same command, split per case"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o b -- python3 bench.py --steps 20 --warmup 5 > $O/bench_profiled_line.json 2> $O/bench_profiled.err || { tail -5 $O/bench_profiled.err; exit 1; }
cp gpurun_out/bench_report.json $O/bench_report_profiled.json
python tools/cases_from_trace.py $O/bench $O/bench_report_profiled.json $O/round3_configs_cases.csv | cut -c1-200 | tail -40
head -8 $O/bench/b_kernel_stats.csv | cut -c1-200
cp $O/bench_report.json gpurun_out/bench_report.json
echo "--- GROUP BY wall per call (fused, then the two-launch form)"
python tools/group_time.py > $O/group_wall.txt 2>/dev/null; AQE_GROUP_UNFUSED=1 python tools/group_time.py >> $O/group_wall.txt 2>/dev/null; cat $O/group_wall.txt
echo "--- peer-mapped mailbox: cost of one all-reduce (same-device floor), and the bench's N > 1 path rehearsed on one GPU (2 ranks, gloo vs mailbox)"
timeout -k 10 200 python tools/mailbox_time.py 1 2 4 > $O/mailbox.txt 2>&1; cat $O/mailbox.txt
for mode in gloo mailbox; do
  [ $mode = mailbox ] && export AQE_BENCH_MAILBOX=1
  AQE_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 10 --warmup 3 --headline-only 2> $O/rehearsal_$mode.err | tail -n 1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$mode', d['config'].get('collective'), 'ms_per_step', round(d['ms_per_step'],3), 'value', round(d['value']))" >> $O/mailbox.txt || tail -3 $O/rehearsal_$mode.err
done
unset AQE_BENCH_MAILBOX
tail -2 $O/mailbox.txt
echo "--- lean launch: ablation + in-kernel timeline"
[ -f tools/lib_nofold.bin ] && tools/ab_ablate.sh run > $O/lean_ablation.txt 2>&1; cat $O/lean_ablation.txt
[ -f tools/lib_stamps.bin ] && AQE_HIP_LIB=$PWD/tools/lib_stamps.bin timeout -k 10 100 python tools/stamp_lean.py clt 2>/dev/null | tail -3 > $O/lean_timeline.txt; cat $O/lean_timeline.txt
rm -f $O/pmc_fetch/f_kernel_trace.csv $O/pmc_write/w_kernel_trace.csv $O/pmc_disjoint/d_kernel_trace.csv
du -sh $O
