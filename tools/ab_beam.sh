#!/bin/bash
# Interleaved A/B of build_ab/ variants on the beam leg: tools/ab_beam.sh TAG name1 name2 ...  (two rounds, 4096 and 8192 games)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=$1; shift
cd $ROOT
: > $OUT/${TAG}_ab.txt
for round in 1 2; do
  for v in "$@"; do
    G2048_LIB=$ROOT/build_ab/libg2048_$v.so timeout -k 10 200 python3 tools/beam_rate.py 4096 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_ab.txt || exit 1
  done
done
for v in "$@"; do
  G2048_LIB=$ROOT/build_ab/libg2048_$v.so timeout -k 10 200 python3 tools/beam_rate.py 8192 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_ab.txt || exit 1
done
cat $OUT/${TAG}_ab.txt
