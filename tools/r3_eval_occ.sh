#!/bin/bash
# round 3: occupancy of the evaluation kernel (wavefronts per SIMD the register budget allows) and the helper cap
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=gpurun_out
: > $OUT/r3_eval_occ.txt
for v in "$@"; do
  echo "== $v" >> $OUT/r3_eval_occ.txt
  G2048_LIB=$ROOT/build_ab/libg2048_$v.so timeout -k 10 100 python3 -c "
import sys, os
sys.path.insert(0, '$ROOT')
import __graft_entry__ as ge
ge.import_package()
from g2048 import ops
print(ops.device_plan(20, 4096))" 2>&1 | grep -v "amdgpu.ids\|loading" >> $OUT/r3_eval_occ.txt
  G2048_LIB=$ROOT/build_ab/libg2048_$v.so timeout -k 10 300 python3 tools/eval_tail_ab.py 4096 20 30 2>&1 | grep -v "amdgpu.ids\|loading" >> $OUT/r3_eval_occ.txt || exit 1
done
cat $OUT/r3_eval_occ.txt
