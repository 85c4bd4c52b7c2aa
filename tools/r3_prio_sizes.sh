#!/bin/bash
# round 3: the beam leg with and without issue priority by remaining levels, over batch sizes; timeline of the shipped policy
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; OUT=gpurun_out
: > $OUT/r3_prio_sizes.txt
for n in 1024 2048 3072 4096 5120 6144 7168 8192 16384; do
  for v in noprio prioE prioC; do
    G2048_LIB=$ROOT/build_ab/libg2048_$v.so timeout -k 10 200 python3 tools/beam_rate.py $n 2>&1 | grep balanced >> $OUT/r3_prio_sizes.txt || exit 1
  done
done
timeout -k 10 200 python3 tools/beam_timeline.py expansions 4096 > $OUT/r3_timelineE.txt 2>&1 || exit 1
G2048_LIB=$ROOT/build_ab/libg2048_timingE.so timeout -k 10 200 python3 tools/beam_timeline.py timeline 4096 2>&1 | grep -v amdgpu.ids >> $OUT/r3_timelineE.txt || exit 1
