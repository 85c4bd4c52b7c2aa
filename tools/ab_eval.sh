#!/bin/bash
# evaluation driver A/B over build_ab/ variants: tools/ab_eval.sh TAG "tunings" variants...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=$1; TUNES=$2; shift; shift
cd $ROOT
: > $OUT/${TAG}_eval.txt
for v in "$@"; do
  echo "== $v" >> $OUT/${TAG}_eval.txt
  G2048_LIB=$ROOT/build_ab/libg2048_$v.so timeout -k 10 300 python3 tools/eval_tail_ab.py 4096 20 30 "$TUNES" 2>&1 | grep -v "amdgpu.ids\|loading" >> $OUT/${TAG}_eval.txt || exit 1
done
cat $OUT/${TAG}_eval.txt
