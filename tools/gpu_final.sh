#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_final.log 2>&1; rc=$?; tail -4 gpurun_out/pytest_final.log; [ $rc -eq 0 ] || exit 1
echo "--- in-kernel timeline of k_sweep_lean (library built with -DAQE_LEAN_STAMPS: tools/ab_libs.sh stamps -DAQE_LEAN_STAMPS)"
if [ -f tools/lib_stamps.bin ]; then AQE_HIP_LIB=$PWD/tools/lib_stamps.bin timeout -k 10 100 python tools/stamp_lean.py clt > gpurun_out/lean_timeline.txt 2>/dev/null; cat gpurun_out/lean_timeline.txt; fi
tools/gpu_f.sh || exit 1
cp gpurun_out/prof_r2/round2_pmc_raw.json profiles/round2_pmc_raw.json  # (measured a minute ago on these very sources: the bench reports it as roofline.traffic)
echo "--- the default bench (unprofiled)"
( time python bench.py ) > gpurun_out/bench_r2_final.txt 2> gpurun_out/bench_r2_final.err
tail -c 300 gpurun_out/bench_r2_final.err
python tools/bench_digest.py gpurun_out/bench_r2_final.txt | cut -c1-330
