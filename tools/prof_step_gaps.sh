#!/bin/bash
# Idle gaps of the DEFAULT (two-stream) training step under rocprofv3 --kernel-trace -> gpurun_out/<tag>_step_gaps.txt, next to the
# un-profiled host-lead table (tools/step_host_lead.py) in gpurun_out/<tag>_host_lead.txt.  The tracer slows every launch on the host side,
# so the gaps it shows are an upper bound on what the un-profiled step has.
tag=${1:-r4}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_g
SIMSEG_BENCH_FP16=0 timeout 400 rocprofv3 --kernel-trace -d /tmp/prof_g -o g -- python $R/bench.py --steps 6 --warmup 3 --no-seg --no-cpu-baseline > /tmp/prof_g.log 2>/tmp/prof_g.err
db=$(find /tmp/prof_g -name "*.db" 2>/dev/null | head -1)
{ echo "# rocprofv3 --kernel-trace -- python bench.py --steps 6 --warmup 3 --no-seg --no-cpu-baseline   (default schedule: two tower streams; tools/rocpd_step_gaps.py, gaps >= 20 us listed)";
  echo "# ms_per_step of this traced run: $(python -c "import json,sys; print(json.loads(open('/tmp/prof_g.log').read().strip().splitlines()[-1])['ms_per_step'])" 2>/dev/null)";
  timeout 120 python $R/tools/rocpd_step_gaps.py "$db" 20; } > $R/gpurun_out/${tag}_step_gaps.txt 2>&1
timeout 600 python $R/tools/step_host_lead.py > $R/gpurun_out/${tag}_host_lead.txt 2>/dev/null
