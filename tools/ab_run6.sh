#!/bin/bash
# Dev helper (GPU box): single-round samplers through the lean launch (threshold 0: always) against k_round (threshold huge), same library.
cd "$(dirname "$0")/.."
for i in 1 2; do
  for t in 0 4000000000; do
    AQE_LEAN_SINGLE_MIN_TILES=$t AQE_HIP_LIB=$PWD/tools/lib_$1.bin timeout -k 10 300 python tools/ab_latency.py 100 ${AB_SIZES:-10000000,100000000,1000000000} 2>/dev/null | tail -1 | sed "s/^/single_min_tiles=$t /"
  done
done
