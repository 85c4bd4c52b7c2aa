#!/bin/bash
# Interleaved A/B of library builds in build_ab/ on the step leg:  bash tools/ab_libs.sh NAME [NAME ...]   (libg2048_NAME.so)
# per build and round: tools/step_rate.py (one launch per step, 100-launch hipGraph) and tools/chains_wall.py 20 (two chains from idle)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for round in 1 2 3; do
  for name in "$@"; do
    echo "== round $round: $name"
    G2048_LIB=build_ab/libg2048_$name.so timeout -k 10 120 python3 tools/step_rate.py 2>&1 | grep "boards"
    G2048_LIB=build_ab/libg2048_$name.so timeout -k 10 120 python3 tools/chains_wall.py 20 2>&1 | grep "chains 2 eager"
  done
done
