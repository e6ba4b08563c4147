#!/bin/bash
# Dev helper: ablation builds of k_sweep_lean (what each stage of a 10 M-row launch costs), then on the GPU box:
#   tools/ab_ablate.sh build      (here)        tools/ab_ablate.sh run   (GPU box)
cd "$(dirname "$0")/.."
V="${ABL_V:-base noticket nofold fold0 fold1 fold2 fold3 nojudge nostore nosweep}"
if [ "$1" = build ]; then
  tools/ab_libs.sh base "" nojudge "-DAQE_ABL_NOJUDGE" nofold "-DAQE_ABL_NOFOLD" noticket "-DAQE_ABL_NOTICKET" nosweep "-DAQE_ABL_NOSWEEP" nostore "-DAQE_ABL_NOSTORE" fold0 "-DAQE_ABL_FOLD0" fold1 "-DAQE_ABL_FOLD1" fold2 "-DAQE_ABL_FOLD2" fold3 "-DAQE_ABL_FOLD3" 2>&1 | grep -v warning | tail -7
else
  for i in 1 2; do for v in $V; do AQE_NO_POLL=1 AB_NO_LOOP=1 AQE_HIP_LIB=$PWD/tools/lib_$v.bin timeout -k 10 120 python tools/ab_single.py 300 2>/dev/null | tail -1; done; done
fi
