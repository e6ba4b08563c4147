#!/bin/bash
# Dev helper (GPU box): tools/ab_latency.py alternated over variant libraries.
cd "$(dirname "$0")/.."
for i in 1 2; do
  for lib in "$@"; do
    AQE_HIP_LIB=$PWD/tools/lib_$lib.bin timeout -k 10 200 python tools/ab_latency.py 150 ${AB_SIZES:-10000000} 2>/dev/null | tail -1
  done
done
