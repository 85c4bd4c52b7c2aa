set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-graph --no-beam --no-rollout --no-evaluation"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_r02_hbm_fetch -- $B > /dev/null 2> $OUT/prof_r02_hbm_fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_r02_hbm_write -- $B > /dev/null 2> $OUT/prof_r02_hbm_write.err
echo "write done"
cd $ROOT
for d in hbm_fetch hbm_write; do python3 tools/prof_summary.py $OUT/prof_r02_$d > $OUT/prof_r02_$d.summary.txt 2>&1; done
grep "step_kernel" $OUT/prof_r02_hbm_fetch.summary.txt | head -12
grep "step_kernel" $OUT/prof_r02_hbm_write.summary.txt | head -12
