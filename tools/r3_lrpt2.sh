#!/bin/bash
# round 3: issue-priority thresholds of the beam search, and the 8192-game timeline with and without them
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=gpurun_out
bash tools/ab_beam.sh r3_lrpt2 base lrptC lrptE lrptF lrptG lrptI > $OUT/r3_lrpt2_log.txt 2>&1 || exit 1
timeout -k 10 200 python3 tools/beam_timeline.py expansions 8192 > $OUT/r3_timeline8k.txt 2>&1 || exit 1
for v in timing timingC; do
  G2048_LIB=$ROOT/build_ab/libg2048_$v.so timeout -k 10 200 python3 tools/beam_timeline.py timeline 8192 2>&1 | grep -v amdgpu.ids | head -11 >> $OUT/r3_timeline8k.txt || exit 1
done
