#!/bin/bash
# beam leg at several batch sizes for one build_ab/ variant: tools/ab_beam_sizes.sh TAG variant sizes...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=$1; V=$2; shift; shift
cd $ROOT
: > $OUT/${TAG}_sizes.txt
for n in "$@"; do
  G2048_LIB=$ROOT/build_ab/libg2048_$V.so timeout -k 10 200 python3 tools/beam_rate.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_sizes.txt || exit 1
done
cat $OUT/${TAG}_sizes.txt
