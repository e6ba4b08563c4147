#!/bin/bash
# Dev helper: device assembly of one csrc/*.hip file (gfx950), for ISA diffs of a kernel across edits.
#   tools/isa.sh persist.hip /tmp/persist.s
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-fast-math -ffp-contract=off -I include -S --cuda-device-only \
  -o "$2" approximatequeryengine_amd/csrc/"$1"
grep -E "^\s+\.(sgpr_count|vgpr_count|sgpr_spill_count|vgpr_spill_count|name):|\.symbol:" "$2" | paste - - - - - - 2>/dev/null | head -20
