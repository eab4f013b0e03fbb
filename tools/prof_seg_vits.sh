#!/bin/bash
# rocprofv3 kernel trace of the ViT-S @512 bf16 seg-eval leg (BASELINE configs[1], 256 windows, no CRF) -> gpurun_out/<tag>.txt
tag=${1:-seg_vits}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/segprof
timeout 500 rocprofv3 --kernel-trace -d /tmp/segprof -o p -- python $GRAFT_REPO_ROOT/tools/seg_vits_prof.py > /tmp/segprof.log 2>&1
db=$(find /tmp/segprof -name "*.db" 2>/dev/null | head -1)
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}.txt
grep "windows_per_s" /tmp/segprof.log > $out
if [ -n "$db" ]; then timeout 120 python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" 30 >> $out 2>&1; else tail -5 /tmp/segprof.log >> $out; fi
