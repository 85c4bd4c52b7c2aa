#!/bin/bash
# The round's rocprofv3 passes on the GPU box (run from the repo root through gpurun):  bash tools/profile_round.sh [TAG]
#   kt     kernel trace + stats of the driver's bench command (per-kernel, per-launch-size durations)
#   fetch / write   FETCH_SIZE and WRITE_SIZE in separate --pmc passes (they do not fit one pass; MI355X_MICROARCH.md) over the
#          step legs, extra legs (f64 reward, 16 Mi boards) included
#   sq     SQ_* counters of the step / beam / evaluation kernels
#   roll_* the same three passes over config 4's kernel (tools/rollout_rate.py 65536)
# The counter passes run the one-launch-per-step form (--chains 1): counters are per dispatch, and two overlapping dispatches
# would share them. Raw output under gpurun_out/prof_TAG_*, summaries via tools/prof_summary.py next to it; copy what is to be
# kept into profiles/.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 5"
R="python3 $ROOT/tools/rollout_rate.py 65536"
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_kt -- $B > $OUT/prof_${TAG}_kt.json 2> $OUT/prof_${TAG}_kt.err || exit 1
echo "kt done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_fetch -- $B --chains 1 --no-cpu-baseline --no-beam --no-ppo-rollout > /dev/null 2> $OUT/prof_${TAG}_fetch.err || exit 2
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_write -- $B --chains 1 --no-cpu-baseline --no-beam --no-ppo-rollout > /dev/null 2> $OUT/prof_${TAG}_write.err || exit 3
echo "write done"
timeout -k 10 600 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/prof_${TAG}_sq -- $B --chains 1 --no-cpu-baseline --no-ppo-rollout > /dev/null 2> $OUT/prof_${TAG}_sq.err || exit 4
echo "sq done"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_roll_fetch -- $R > /dev/null 2> $OUT/prof_${TAG}_roll_fetch.err || exit 5
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_roll_write -- $R > /dev/null 2> $OUT/prof_${TAG}_roll_write.err || exit 6
timeout -k 10 200 rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/prof_${TAG}_roll_sq -- $R > /dev/null 2> $OUT/prof_${TAG}_roll_sq.err || exit 7
echo "rollout passes done"
cd $ROOT
for d in kt fetch write sq; do python3 tools/prof_summary.py $OUT/prof_${TAG}_$d > $OUT/prof_${TAG}_$d.summary.txt 2>&1; done
for d in roll_fetch roll_write roll_sq; do python3 tools/prof_summary.py $OUT/prof_${TAG}_$d rollout_step > $OUT/prof_${TAG}_$d.summary.txt 2>&1; done
python3 tools/chains_timeline.py $OUT/prof_${TAG}_kt > $OUT/prof_${TAG}_kt.chains.txt 2>&1
python3 tools/bench_digest.py $OUT/prof_${TAG}_kt.json
