#!/bin/bash
# Round-4 full GPU pass: the whole -m gpu suite, the beam leg alone, the default bench line (TAG names the outputs).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-r4}
cd $ROOT
timeout -k 10 1500 python3 -m pytest tests -q -m gpu -s > $OUT/${TAG}_pytest_gpu.log 2>&1 || { tail -60 $OUT/${TAG}_pytest_gpu.log; exit 1; }
tail -2 $OUT/${TAG}_pytest_gpu.log; grep "drop-in Game2048Env" $OUT/${TAG}_pytest_gpu.log
timeout -k 10 300 python3 tools/beam_rate.py 4096 > $OUT/${TAG}_beam_rate.txt 2>&1 && timeout -k 10 300 python3 tools/beam_rate.py 8192 >> $OUT/${TAG}_beam_rate.txt 2>&1
grep -v amdgpu.ids $OUT/${TAG}_beam_rate.txt
timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || { tail -20 $OUT/${TAG}_bench.err; exit 2; }
python3 tools/bench_digest.py $OUT/${TAG}_bench.json
