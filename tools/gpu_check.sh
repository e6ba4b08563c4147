#!/bin/bash
# Dev helper (GPU box): the whole GPU suite, then the default bench; prints a digest of the bench line.
#   tools/gpu_check.sh [tag] [pytest -k expression]
cd "$(dirname "$0")/.."
tag=${1:-a}
if [ -n "$2" ]; then python -m pytest tests -x -q -m gpu -k "$2" > gpurun_out/pytest_$tag.log 2>&1; rc=$?; tail -8 gpurun_out/pytest_$tag.log; [ $rc -eq 0 ] || exit 1
else python -m pytest tests -x -q -m gpu > gpurun_out/pytest_$tag.log 2>&1; rc=$?; tail -8 gpurun_out/pytest_$tag.log; [ $rc -eq 0 ] || exit 1; fi
( time python bench.py $BENCH_ARGS ) > gpurun_out/bench_r2_$tag.txt 2> gpurun_out/bench_r2_$tag.err
tail -c 800 gpurun_out/bench_r2_$tag.err
python tools/bench_digest.py gpurun_out/bench_r2_$tag.txt
