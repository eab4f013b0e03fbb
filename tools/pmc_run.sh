#!/bin/bash
# usage: tools/pmc_run.sh <tag> <kernel-name-substring> <counters (space separated, <= 8 SQ)> -- <python args...>
# one rocprofv3 --pmc pass (kernel trace only, as the pool requires), per-kernel averages appended to gpurun_out/<tag>.txt
tag=$1; pat=$2; ctr=$3; shift 4
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d /tmp/pmc_$tag -o p -- python "$@" > /tmp/pmc_$tag.log 2>&1
db=$(find /tmp/pmc_$tag -name "*.db" 2>/dev/null | head -1)
if [ -n "$db" ]; then
  echo "## $ctr -- $*" >> $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
  timeout 120 python $GRAFT_REPO_ROOT/tools/pmc_stats.py "$db" "$pat" >> $GRAFT_REPO_ROOT/gpurun_out/$tag.txt 2>&1
else
  echo "no database for $ctr" >> $GRAFT_REPO_ROOT/gpurun_out/$tag.txt; tail -5 /tmp/pmc_$tag.log >> $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
fi
