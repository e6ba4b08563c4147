#!/bin/bash
# Dev helper: build variants of libaqe_hip.so with extra -D flags into tools/lib_<name>.bin
#   tools/ab_libs.sh name1 "-DX=1" name2 "-DX=0" ...
set -e
cd "$(dirname "$0")/.."
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -fvisibility=hidden -Wall -Wno-unused-function \
    -fno-fast-math -ffp-contract=off $flags -I include -o tools/lib_$name.bin \
    approximatequeryengine_amd/csrc/capi.hip approximatequeryengine_amd/csrc/table.hip approximatequeryengine_amd/csrc/plans.hip approximatequeryengine_amd/csrc/kernels.hip approximatequeryengine_amd/csrc/persist.hip approximatequeryengine_amd/csrc/lean.hip approximatequeryengine_amd/csrc/grouped.hip \
    approximatequeryengine_amd/csrc/sort.hip approximatequeryengine_amd/csrc/comm.hip approximatequeryengine_amd/csrc/mailbox.hip approximatequeryengine_amd/csrc/planner.cpp -ldl &
done
wait
ls -la tools/lib_*.bin
