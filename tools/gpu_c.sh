#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests -x -q -m gpu 2>&1 | tail -15 || exit 1
python tools/group_time.py > gpurun_out/group_time_r2.txt 2>&1; cat gpurun_out/group_time_r2.txt
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_group -o g -- python3 tools/group_time.py > /dev/null 2>&1
f=$(find gpurun_out/prof_group -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-200
