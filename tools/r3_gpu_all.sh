#!/bin/bash
# Round-3 full GPU pass: the whole -m gpu suite, the beam leg alone, the default bench line.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-r3}
cd $ROOT
timeout -k 10 1500 python3 -m pytest tests -x -q -m gpu > $OUT/${TAG}_pytest_gpu.log 2>&1 || { tail -40 $OUT/${TAG}_pytest_gpu.log; exit 1; }
tail -2 $OUT/${TAG}_pytest_gpu.log
timeout -k 10 300 python3 tools/beam_rate.py 4096 > $OUT/${TAG}_beam_rate.txt 2>&1 && timeout -k 10 300 python3 tools/beam_rate.py 8192 >> $OUT/${TAG}_beam_rate.txt 2>&1
grep -v amdgpu.ids $OUT/${TAG}_beam_rate.txt
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -20 $OUT/${TAG}_bench.err; exit 2; }
python3 - <<PY
import json
r = json.load(open("$OUT/${TAG}_bench.json"))
print("value %.3e ms/step %.4f frac %.3f kernel_us %.2f" % (r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["kernel_us"]))
for k in ("roofline_f64_reward", "roofline_hbm_resident"):
    if k in r: print(k, "%.3f" % r[k]["frac"], "%.2f us" % r[k]["kernel_us"])
if "beam" in r: print("beam %.3e  %.4f ms" % (r["beam"]["value"], r["beam"]["ms_per_batch_decision"]))
if "evaluation" in r: print("evaluation", r["evaluation"]["seconds"], r["evaluation"]["seconds_without_helper_wavefronts"], r["evaluation"]["same_games_without_helpers"])
if "rollout_random" in r: print("rollout_random", r["rollout_random"])
if "rollout" in r: print({k: v for k, v in r["rollout"].items() if isinstance(v, float)})
PY
