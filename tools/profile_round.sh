#!/bin/bash
# Round profile of the training step on the GPU box (every stage under its own timeout):
#   1. rocprofv3 --kernel-trace of 7 single-stream training steps  -> gpurun_out/<tag>_train_step_kernel_stats_single_stream.txt
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE) of one step          -> gpurun_out/<tag>_pmc_fetch_size.txt / _pmc_write_size.txt
tag=${1:-r2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k /tmp/prof_f /tmp/prof_w
SIMSEG_BENCH_FP16=0 SIMSEG_AMD_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_k -o k -- python $R/bench.py --steps 5 --warmup 2 --no-seg --no-cpu-baseline > /tmp/prof_k.log 2>&1
db=$(find /tmp/prof_k -name "*.db" 2>/dev/null | head -1)
if [ -n "$db" ]; then
  { echo "# SIMSEG_AMD_TWO_STREAMS=0 rocprofv3 --kernel-trace -- python bench.py --steps 5 --warmup 2 --no-seg --no-cpu-baseline   (8 training steps, both towers on ONE stream: per-kernel durations are each kernel's own; tools/rocpd_stats.py)"; timeout 120 python $R/tools/rocpd_stats.py "$db" 45; } > $R/gpurun_out/${tag}_train_step_kernel_stats_single_stream.txt 2>&1
fi
for c in FETCH_SIZE WRITE_SIZE; do
  d=/tmp/prof_$c; rm -rf $d
  SIMSEG_BENCH_FP16=0 SIMSEG_AMD_TWO_STREAMS=0 timeout 400 rocprofv3 --pmc $c --kernel-trace -d $d -o p -- python $R/bench.py --steps 1 --warmup 1 --no-seg --no-cpu-baseline > $d.log 2>&1
  db=$(find $d -name "*.db" 2>/dev/null | head -1)
  lc=$(echo $c | tr A-Z a-z)
  if [ -n "$db" ]; then timeout 120 python $R/tools/pmc_stats.py "$db" gemm > $R/gpurun_out/${tag}_pmc_${lc}.txt 2>&1; else tail -5 $d.log > $R/gpurun_out/${tag}_pmc_${lc}.txt; fi
done
