#!/bin/bash
# rocprofv3 kernel trace of the exact-mode (fp32) ViT-B@512 seg-eval leg without the CRF -> gpurun_out/<tag>.txt
tag=${1:-seg_fp32}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/segprof
SEG_BENCH_LEGS=1 timeout 500 rocprofv3 --kernel-trace -d /tmp/segprof -o p -- python $GRAFT_REPO_ROOT/tools/seg_bench.py nocrf > /tmp/segprof.log 2>&1
db=$(find /tmp/segprof -name "*.db" 2>/dev/null | head -1)
out=$GRAFT_REPO_ROOT/gpurun_out/${tag}.txt
grep "windows_per_s" /tmp/segprof.log > $out
if [ -n "$db" ]; then timeout 120 python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" 25 >> $out 2>&1; else tail -5 /tmp/segprof.log >> $out; fi
