#!/bin/bash
# Dev helper (GPU box): tools/ab_latency.py alternated over variant libraries and k_round grid caps.
cd "$(dirname "$0")/.."
for i in 1 2; do
  for lib in "$@"; do
    for cap in 2048 1024; do
      AQE_ROUND_MAX_BLOCKS=$cap AQE_HIP_LIB=$PWD/tools/lib_$lib.bin timeout -k 10 200 python tools/ab_latency.py 150 2>/dev/null | tail -1
    done
  done
done
