#!/bin/bash
# Dev helper (GPU box): alternate bench.py --batch 1 over the variant libraries, 3 rounds each.
cd "$(dirname "$0")/.."
for i in 1 2 3; do
  for lib in "$@"; do
    AQE_HIP_LIB=$PWD/tools/lib_$lib.bin timeout -k 10 120 python bench.py --steps 300 --warmup 30 --no-cpu-baseline --batch 1 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['roofline']['avg_launch_us'],2), round(d['value']))"
  done
done
