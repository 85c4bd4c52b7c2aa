#!/bin/bash
# A/B builds of the library with extra -D flags into build_ab/ (git-ignored; travels to the GPU box): tools/build_ab.sh NAME [-DFLAG ...]
set -e
ROOT=$(cd $(dirname $0)/.. && pwd)
NAME=$1; shift
C=$ROOT/2048-using-reinforcement-learning_amd/csrc
mkdir -p $ROOT/build_ab
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fvisibility=hidden -mllvm -amdgpu-kernarg-preload-count=16 -Wl,--version-script=$C/g2048_exports.map -Wall -Wno-unused-function "$@" -o $ROOT/build_ab/libg2048_$NAME.so $C/g2048_kernels.hip $C/g2048_beam.hip $C/g2048_rollout.hip
echo built build_ab/libg2048_$NAME.so "$@"
