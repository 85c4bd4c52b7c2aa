#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd $ROOT
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/r5_bench_a.json 2> $OUT/r5_bench_a.err || { tail -30 $OUT/r5_bench_a.err; exit 1; }
python3 tools/bench_digest.py $OUT/r5_bench_a.json
timeout -k 10 600 python3 bench.py --steps 100 --warmup 5 --no-beam --no-rollout --no-extra --no-cpu-baseline > $OUT/r5_bench_a100.json 2> $OUT/r5_bench_a100.err || { tail -30 $OUT/r5_bench_a100.err; exit 1; }
python3 tools/bench_digest.py $OUT/r5_bench_a100.json
timeout -k 10 600 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_bench_line.py -x -q -m gpu > $OUT/r5_pytest_a.log 2>&1 || { tail -40 $OUT/r5_pytest_a.log; exit 2; }
tail -3 $OUT/r5_pytest_a.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof_r05_chains_kt2 -- python3 $ROOT/tools/chains_trace.py 20 8 > $OUT/r5_chains_trace2.log 2>&1 || { tail $OUT/r5_chains_trace2.log; exit 3; }
cd $ROOT && python3 tools/chains_timeline.py $OUT/prof_r05_chains_kt2 --dump 12 > $OUT/r5_chains_timeline2.txt 2>&1
cat $OUT/r5_chains_timeline2.txt
