#!/bin/bash
# Round-3 profiles on the GPU box (run from the repo root through gpurun): kernel trace + stats of the driver's bench command,
# then the separate PMC passes the microarchitecture guide prescribes (FETCH_SIZE and WRITE_SIZE do not fit one pass) and the SQ
# counters of the step / step_many / beam / play kernels. Counter passes keep the hipGraph legs (profiles/r03_pmc_graph_probe.txt:
# capture and replay work under counter collection) but skip the stock-torch PPO policies. Summaries via tools/prof_summary.py.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/bench.py --steps 20 --warmup 5"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r03_kt -- $B > $OUT/prof_r03_kt.json 2> $OUT/prof_r03_kt.err || exit 1
echo "kt done"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_r03_fetch -- $B --no-cpu-baseline --no-beam --no-ppo-rollout > /dev/null 2> $OUT/prof_r03_fetch.err || exit 2
echo "fetch done"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_r03_write -- $B --no-cpu-baseline --no-beam --no-ppo-rollout > /dev/null 2> $OUT/prof_r03_write.err || exit 3
echo "write done"
timeout -k 10 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/prof_r03_sq -- $B --no-cpu-baseline --no-ppo-rollout > /dev/null 2> $OUT/prof_r03_sq.err || exit 4
echo "sq done"
cd $ROOT
for d in kt fetch write sq; do python3 tools/prof_summary.py $OUT/prof_r03_$d > $OUT/prof_r03_$d.summary.txt 2>&1; done
tail -1 $OUT/prof_r03_kt.json | head -c 400; echo
