#!/bin/bash
# usage: tools/prof_crf.sh <tag>: rocprofv3 kernel trace of the seg-eval legs with the device DenseCRF -> gpurun_out/<tag>.txt
tag=$1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/crfprof
timeout 500 rocprofv3 --kernel-trace -d /tmp/crfprof -o p -- python $GRAFT_REPO_ROOT/tools/seg_bench.py crf > /tmp/crfprof.log 2>&1
db=$(find /tmp/crfprof -name "*.db" 2>/dev/null | head -1)
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}.txt
grep "windows_per_s" /tmp/crfprof.log > $out
if [ -n "$db" ]; then timeout 120 python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" 30 >> $out 2>&1; timeout 120 python $GRAFT_REPO_ROOT/tools/rocpd_gaps.py "$db" 0.0 >> $out 2>&1; else tail -5 /tmp/crfprof.log >> $out; fi
