#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests -x -q -m gpu 2>&1 | tail -4 || exit 1
tools/gpu_f.sh || exit 1
echo "--- the default bench (unprofiled)"
( time python bench.py ) > gpurun_out/bench_r2_final.txt 2> gpurun_out/bench_r2_final.err
tail -c 300 gpurun_out/bench_r2_final.err
python tools/bench_digest.py gpurun_out/bench_r2_final.txt | cut -c1-330
