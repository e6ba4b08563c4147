#!/bin/bash
cd "$(dirname "$0")/.."
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 || exit 1
echo "--- RCCL world of one through the N>1 code path"
AQE_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --headline-only --steps 100 > gpurun_out/bench_r2_forcedist.txt 2> gpurun_out/bench_r2_forcedist.err || { tail -20 gpurun_out/bench_r2_forcedist.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r2_forcedist.txt").read().strip().splitlines()[-1])
print("forcedist value", round(d["value"]), "collective", d["config"]["collective"], "roofline", round(d["roofline"]["frac"], 3), d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"], 1))
PY
echo "--- two ranks on one GPU over gloo (rehearsal of the N>1 path)"
AQE_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --headline-only --steps 50 > gpurun_out/bench_r2_rehearsal.txt 2> gpurun_out/bench_r2_rehearsal.err || { tail -20 gpurun_out/bench_r2_rehearsal.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_r2_rehearsal.txt").read().strip().splitlines()[-1])
print("rehearsal n_gpus", d["n_gpus"], "value", round(d["value"]), "collective", d["config"]["collective"], "result n", d["result"]["n"])
PY
echo "--- the default bench"
( time python bench.py ) > gpurun_out/bench_r2_e.txt 2> gpurun_out/bench_r2_e.err
tail -c 600 gpurun_out/bench_r2_e.err
python tools/bench_digest.py gpurun_out/bench_r2_e.txt
