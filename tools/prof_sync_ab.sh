#!/bin/bash
# Kernel statistics of the training step with and without the N > 1 gradient exchange machinery forced on one rank (GradSync bucket):
# which kernels and how much time the exchange adds.  -> gpurun_out/<tag>_kstats_{none,bucket}.txt
tag=${1:-r4}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for mode in none bucket; do
  d=/tmp/prof_s_$mode; rm -rf $d
  if [ $mode = bucket ]; then export SIMSEG_BENCH_FORCE_SYNC=1 SIMSEG_BENCH_DP=bucket; else unset SIMSEG_BENCH_FORCE_SYNC; fi
  SIMSEG_BENCH_FP16=0 timeout 400 rocprofv3 --kernel-trace -d $d -o k -- python $R/bench.py --steps 5 --warmup 2 --no-seg --no-cpu-baseline > $d.log 2>&1
  db=$(find $d -name "*.db" 2>/dev/null | head -1)
  { tail -1 $d.log | cut -c1-200; timeout 120 python $R/tools/rocpd_stats.py "$db" 40; } > $R/gpurun_out/${tag}_kstats_$mode.txt 2>&1
done
