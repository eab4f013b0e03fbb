set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/scratch/attn_w64_occ.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r6_attn_occ.txt
python tools/scratch/attn_w64_check.py 2>&1 | grep -v amdgpu.ids | grep -c " ok" 
( ./tools/scratch/ubench_issue_m0 | grep -v "tile,\|mix [0-9]"; for m in 0 1 2 3 4; do ./tools/scratch/ubench_issue_m$m 2>&1 | grep "tile,\|mix [0-9]" | grep -v behind; done ) > gpurun_out/r6_ubench_issue.txt 2>&1
tail -5 gpurun_out/r6_attn_occ.txt
