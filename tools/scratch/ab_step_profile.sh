#!/bin/bash
# Same-box kernel statistics of the single-stream training step for two attention variants (argument list, default "3 0"):
# does a kernel change move the OTHER kernels' durations (shader clock under the package power cap)?
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in ${@:-3 0}; do
  rm -rf /tmp/prof_v$v
  SIMSEG_ATTN_VARIANT=$v SIMSEG_AMD_TWO_STREAMS=0 timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_v$v -o k -- python $R/bench.py --steps 5 --warmup 2 --no-seg --no-cpu-baseline > /tmp/prof_v$v.log 2>&1
  db=$(find /tmp/prof_v$v -name "*.db" 2>/dev/null | head -1)
  { echo "# SIMSEG_ATTN_VARIANT=$v SIMSEG_AMD_TWO_STREAMS=0 rocprofv3 --kernel-trace -- python bench.py --steps 5 --warmup 2 --no-seg --no-cpu-baseline"; tail -1 /tmp/prof_v$v.log | cut -c1-300; timeout 120 python $R/tools/rocpd_stats.py "$db" 16; } > $R/gpurun_out/ab_step_variant_$v.txt 2>&1
done
