#!/bin/bash
# SQ_INSTS_VALU of the beam kernel alone (bench.py beam leg only), plus the beam parity tests
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out
cd $ROOT && timeout -k 10 300 python3 -m pytest tests/test_gpu_beam.py -x -q -m gpu 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/prof_r03_beamonly -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-ppo-rollout --no-evaluation > /dev/null 2> $OUT/prof_r03_beamonly.err || exit 4
cd $ROOT && python3 tools/prof_summary.py $OUT/prof_r03_beamonly 2>&1 | grep "beam_kernel<2>"
