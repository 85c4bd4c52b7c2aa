#!/bin/bash
# tools/ab_sizes2.sh TAG "sizes" variants...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=$1; SIZES=$2; shift; shift
cd $ROOT
: > $OUT/${TAG}_sizes.txt
for n in $SIZES; do for v in "$@"; do
  G2048_LIB=$ROOT/build_ab/libg2048_$v.so timeout -k 10 200 python3 tools/beam_rate.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_sizes.txt || exit 1
done; done
cat $OUT/${TAG}_sizes.txt
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
G2048_LIB=$ROOT/build_ab/libg2048_$v.so timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_prof_$v -- python3 $ROOT/tools/beam_rate.py 4096 > $OUT/${TAG}_prof_$v.log 2>&1 || exit 2
python3 $ROOT/tools/prof_summary.py $OUT/${TAG}_prof_$v "beam_kernel<2>" | grep -v "^==" | cut -c1-30,70-200
done
