#!/bin/bash
# Round profiles on the GPU box (run from the repo root through gpurun): kernel trace + stats of the bench command, the
# separate PMC passes the microarchitecture guide prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass), the SQ
# counters of the step / beam / play / rollout kernels. Raw output under gpurun_out/, summaries via tools/prof_summary.py.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r02_kt -- $B > $OUT/prof_r02_kt.json 2> $OUT/prof_r02_kt.err
echo "kt done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_r02_fetch -- $B --no-graph --no-beam --no-rollout --no-extra > /dev/null 2> $OUT/prof_r02_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_r02_write -- $B --no-graph --no-beam --no-rollout --no-extra > /dev/null 2> $OUT/prof_r02_write.err
echo "write done"
# (no graph-captured leg in a counter pass: the rollout collector's hipGraph capture hung under --pmc)
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/prof_r02_sq -- $B --no-graph --no-extra --no-rollout > /dev/null 2> $OUT/prof_r02_sq.err
echo "sq done"
cd $ROOT
for d in kt fetch write sq; do python3 tools/prof_summary.py $OUT/prof_r02_$d > $OUT/prof_r02_$d.summary.txt 2>&1; done
head -c 300 $OUT/prof_r02_kt.json; echo
