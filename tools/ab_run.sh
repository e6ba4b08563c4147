#!/bin/bash
# Dev helper (GPU box): alternate tools/ab_single.py over variant libraries (tools/lib_<name>.bin), 3 rounds each.
cd "$(dirname "$0")/.."
for i in 1 2 3; do for lib in "$@"; do AQE_HIP_LIB=$PWD/tools/lib_$lib.bin timeout -k 10 120 python tools/ab_single.py 300 2>/dev/null | tail -1; done; done
