#!/bin/bash
# usage: tools/prof_b1.sh <tag>: rocprofv3 kernel trace of the three batch-1 latency configurations -> gpurun_out/<tag>_{0,1,2}.txt
tag=$1
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "fp32 288 21 vit_small_patch16_224_in21k 384" "fp32 512 171 vit_base_patch16_224_in21k 768" "bf16 512 171 vit_base_patch16_224_in21k 768"; do
  rm -rf /tmp/b1_$i
  timeout 280 rocprofv3 --kernel-trace -d /tmp/b1_$i -o p -- python $GRAFT_REPO_ROOT/tools/latency_b1.py $cfg 20 > /tmp/b1_$i.log 2>&1
  db=$(find /tmp/b1_$i -name "*.db" 2>/dev/null | head -1)
  out=$GRAFT_REPO_ROOT/gpurun_out/${tag}_$i.txt
  echo "## $cfg" > $out
  grep "eager_ms" /tmp/b1_$i.log >> $out
  if [ -n "$db" ]; then timeout 120 python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" 25 >> $out 2>&1; echo "## last 30 % of the timeline (graph replay)" >> $out; timeout 120 python $GRAFT_REPO_ROOT/tools/rocpd_stats.py "$db" 12 0.7 >> $out 2>&1; timeout 120 python $GRAFT_REPO_ROOT/tools/rocpd_gaps.py "$db" 0.7 >> $out 2>&1; else tail -5 /tmp/b1_$i.log >> $out; fi
  i=$((i+1))
done
