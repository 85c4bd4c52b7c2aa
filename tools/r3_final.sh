#!/bin/bash
# round 3 closing pass: whole GPU suite + bench line, then the kernel trace of the driver's command
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; TAG=${1:-r3z}
cd $ROOT
bash tools/r3_gpu_all.sh $TAG > $OUT/${TAG}_all.txt 2>&1 || { tail -20 $OUT/${TAG}_all.txt; exit 1; }
head -4 $OUT/${TAG}_all.txt | cut -c1-150; grep "^value\|^beam\|^evaluation\|^roofline" $OUT/${TAG}_all.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r03_kt -- python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/prof_r03_kt.json 2> $OUT/prof_r03_kt.err || exit 2
cd $ROOT && python3 tools/prof_summary.py $OUT/prof_r03_kt > $OUT/prof_r03_kt.summary.txt 2>&1
grep "play_spec_kernel<2>\|play_kernel<2>\|beam_kernel<2>" $OUT/prof_r03_kt.summary.txt | grep "n=" | cut -c1-200
