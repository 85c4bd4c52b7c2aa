#!/bin/bash
# Round-4 profiles on the GPU box (run from the repo root through gpurun): kernel trace + stats of the driver's bench command, the
# separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit one pass) over the step / beam / evaluation legs, and -- new -- the
# same passes for config 4's kernel (rollout_step_kernel) on tools/rollout_rate.py. Summaries via tools/prof_summary.py.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 5"
R="python3 $ROOT/tools/rollout_rate.py 65536"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${TAG}_kt -- $B > $OUT/prof_${TAG}_kt.json 2> $OUT/prof_${TAG}_kt.err || exit 1
echo "kt done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_fetch -- $B --no-cpu-baseline --no-beam --no-ppo-rollout > /dev/null 2> $OUT/prof_${TAG}_fetch.err || exit 2
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_write -- $B --no-cpu-baseline --no-beam --no-ppo-rollout > /dev/null 2> $OUT/prof_${TAG}_write.err || exit 3
echo "write done"
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/prof_${TAG}_sq -- $B --no-cpu-baseline --no-ppo-rollout > /dev/null 2> $OUT/prof_${TAG}_sq.err || exit 4
echo "sq done"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_roll_fetch -- $R > /dev/null 2> $OUT/prof_${TAG}_roll_fetch.err || exit 5
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_${TAG}_roll_write -- $R > /dev/null 2> $OUT/prof_${TAG}_roll_write.err || exit 6
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/prof_${TAG}_roll_sq -- $R > /dev/null 2> $OUT/prof_${TAG}_roll_sq.err || exit 7
echo "rollout passes done"
cd $ROOT
for d in kt fetch write sq; do python3 tools/prof_summary.py $OUT/prof_${TAG}_$d > $OUT/prof_${TAG}_$d.summary.txt 2>&1; done
for d in roll_fetch roll_write roll_sq; do python3 tools/prof_summary.py $OUT/prof_${TAG}_$d rollout_step > $OUT/prof_${TAG}_$d.summary.txt 2>&1; done
cat $OUT/prof_${TAG}_roll_*.summary.txt | grep -v "^==" | cut -c1-40,70-200
timeout -k 10 200 python3 tools/rollout_rate.py > $OUT/${TAG}_rollout_rate.txt 2>&1; grep envs $OUT/${TAG}_rollout_rate.txt
