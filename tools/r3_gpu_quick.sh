#!/bin/bash
# beam / evaluation parity tests on the default build, then an interleaved A/B of build_ab/ variants: tools/r3_gpu_quick.sh TAG variants...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=$1; shift
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_beam.py tests/test_gpu_evaluate.py -x -q -m gpu > $OUT/${TAG}_pytest_beam.log 2>&1 || { tail -30 $OUT/${TAG}_pytest_beam.log; exit 1; }
tail -1 $OUT/${TAG}_pytest_beam.log
bash tools/ab_beam.sh $TAG "$@"
