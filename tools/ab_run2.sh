#!/bin/bash
cd "$(dirname "$0")/.."
for i in 1 2; do for lib in "$@"; do AQE_HIP_LIB=$PWD/tools/lib_$lib.bin timeout -k 10 200 python tools/ab_nt.py 2>/dev/null | tail -1; done; done
