#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=$1
cd $ROOT
: > $OUT/${TAG}_bal.txt
for r in 1 2; do
for n in 4096 8192; do
  python3 tools/beam_rate.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_bal.txt
  NO_BALANCE=1 python3 tools/beam_rate.py $n 2>&1 | grep -v amdgpu.ids >> $OUT/${TAG}_bal.txt
done; done
cat $OUT/${TAG}_bal.txt
