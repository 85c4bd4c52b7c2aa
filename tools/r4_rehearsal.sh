#!/bin/bash
# The driver's multi-GPU command rehearsed on ONE card with gloo: tools/r4_rehearsal.sh RANKS
# (the box allows 6 processes on the card; the launcher holds one, so at most 5 ranks -- 4 leaves a margin)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-4}
cd $ROOT
G2048_DIST_BACKEND=gloo timeout -k 10 1000 python3 bench.py --gpus $N --steps 20 --warmup 5 > gpurun_out/r4_bench_n${N}_gloo.json 2> gpurun_out/r4_bench_n${N}_gloo.err
echo rc=$?
python3 - <<PY
import json
r = json.loads([l for l in open("gpurun_out/r4_bench_n${N}_gloo.json").read().splitlines() if l.startswith("{")][-1])
print({k: r.get(k) for k in ("value", "n_gpus", "ms_per_step", "n_ranks_seen", "backend", "gathered_equals_single_gpu", "allgather_scores_ms")})
print(r.get("evaluation_sharded"))
print("beam", r["beam"]["value"], "rollout_random", r["rollout_random"]["value"])
PY
tail -3 gpurun_out/r4_bench_n${N}_gloo.err | cut -c1-300
