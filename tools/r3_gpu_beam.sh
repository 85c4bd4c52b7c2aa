#!/bin/bash
# Round-3 beam measurements on the GPU box: parity tests of the beam / evaluation paths, the beam leg alone, and two SQ
# counter passes over it (tools/beam_rate.py is the profiled program: 3 + 60 launches of beam_kernel<2>, no hipGraph).
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
TAG=${1:-r3a}
cd $ROOT
timeout -k 10 900 python3 -m pytest tests/test_gpu_beam.py tests/test_gpu_evaluate.py -x -q -m gpu > $OUT/${TAG}_pytest_beam.log 2>&1 || { tail -30 $OUT/${TAG}_pytest_beam.log; exit 1; }
tail -2 $OUT/${TAG}_pytest_beam.log
timeout -k 10 300 python3 tools/beam_rate.py 4096 > $OUT/${TAG}_beam_rate.txt 2>&1 && timeout -k 10 300 python3 tools/beam_rate.py 8192 >> $OUT/${TAG}_beam_rate.txt 2>&1
cat $OUT/${TAG}_beam_rate.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/${TAG}_prof_sq1 -- python3 $ROOT/tools/beam_rate.py 4096 > $OUT/${TAG}_prof_sq1.log 2>&1 || exit 2
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_prof_sq2 -- python3 $ROOT/tools/beam_rate.py 4096 > $OUT/${TAG}_prof_sq2.log 2>&1 || exit 3
cd $ROOT
for d in sq1 sq2; do python3 tools/prof_summary.py $OUT/${TAG}_prof_$d beam_kernel > $OUT/${TAG}_prof_$d.summary.txt 2>&1; done
cat $OUT/${TAG}_prof_sq1.summary.txt $OUT/${TAG}_prof_sq2.summary.txt | grep -v "^==" | cut -c1-40,70-200
