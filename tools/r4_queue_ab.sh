#!/bin/bash
# Evaluation driver, request queue against bound slots: tools/r4_queue_ab.sh TAG  (build_ab/libg2048_<variant>.so [+ _pt = -DG2048_INSTRUMENT=2])
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; TAG=${1:-r4q}; shift
cd $ROOT
TUNES="768,512,16,150;768,1024,16,150;768,512,16,40;768,512,16,0;1536,512,16,150"
: > $OUT/${TAG}_eval.txt
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v (round $rep)" >> $OUT/${TAG}_eval.txt
    G2048_LIB=$ROOT/build_ab/libg2048_$v.so timeout -k 10 300 python3 tools/eval_tail_ab.py 4096 20 30 "$TUNES" 2>&1 | grep -v "amdgpu.ids\|loading" >> $OUT/${TAG}_eval.txt || exit 1
  done
done
cat $OUT/${TAG}_eval.txt
for v in "$@"; do
  if [ -f $ROOT/build_ab/libg2048_${v}_pt.so ]; then
    G2048_LIB=$ROOT/build_ab/libg2048_${v}_pt.so timeout -k 10 200 python3 tools/play_timeline.py 4096 2>&1 | grep -v "amdgpu.ids\|loading" > $OUT/${TAG}_timeline_$v.txt || exit 2
    head -16 $OUT/${TAG}_timeline_$v.txt
  fi
done
