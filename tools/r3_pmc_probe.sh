#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
export PROBE_T=128 PROBE_POLICY=transformer
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES --kernel-trace --output-format csv -d $OUT/r3_probe_prof2 -- python3 $ROOT/tools/pmc_graph_probe.py r3pmc2 > $OUT/r3pmc2_probe.log 2>&1; echo "pmc rc=$?"; cat $OUT/r3pmc2_pmc_probe.txt; tail -3 $OUT/r3pmc2_probe.log | cut -c1-300; ls $OUT/r3_probe_prof2/*/ 2>/dev/null | head; rm -rf $OUT/r3_probe_prof2
