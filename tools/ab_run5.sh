#!/bin/bash
# Dev helper (GPU box): the lean launch against the persistent sweep on larger tables (AQE_LEAN_TILES_PER_WAVE lifts the lean kernel's size limit).
cd "$(dirname "$0")/.."
for i in 1 2; do
  for t in 2 1000; do
    AQE_LEAN_TILES_PER_WAVE=$t timeout -k 10 300 python tools/ab_latency.py 100 100000000,1000000000 2>/dev/null | tail -1 | sed "s/^/tiles_per_wave<=$t /"
  done
done
