#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests -x -q -m gpu -k "group or batch" 2>&1 | tail -4 || exit 1
python tools/group_time.py 2>&1 | tee gpurun_out/group_time_r2b.txt
tools/ab_run.sh time count 2>&1 | tee gpurun_out/ab_giveup.txt
