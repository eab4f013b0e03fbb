#!/bin/bash
# Round-4 SQ / cache counter passes (rocprofv3 --pmc with --kernel-trace only, one pass per counter set) of the kernels that dominate the
# training step now: the persistent dgrad / forward GEMM (gemm_pp2_kernel), the split-K weight-gradient GEMM (gemm_pp_kernel<float, TN>),
# the one-kernel attention backward and the resident attention forward.  -> gpurun_out/<tag>_pmc_sq_gemm.txt, <tag>_pmc_sq_attn.txt
tag=${1:-r4}
R=$GRAFT_REPO_ROOT
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU"
B="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAVES"
C="GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"
rm -f $R/gpurun_out/${tag}_pmc_sq_gemm.txt $R/gpurun_out/${tag}_pmc_sq_attn.txt
for ctr in "$A" "$B" "$C"; do
  bash $R/tools/pmc_run.sh ${tag}_pmc_sq_gemm gemm_pp "$ctr" -- $R/tools/gemm_bench.py --iters 3 --only nn
  bash $R/tools/pmc_run.sh ${tag}_pmc_sq_gemm gemm_pp "$ctr" -- $R/tools/gemm_bench.py --iters 3 --only nt
  bash $R/tools/pmc_run.sh ${tag}_pmc_sq_gemm gemm_pp "$ctr" -- $R/tools/gemm_bench.py --iters 3 --only tn
  bash $R/tools/pmc_run.sh ${tag}_pmc_sq_attn attn_ "$ctr" -- $R/tools/dbg_attn_one.py 197 0 bwd
  bash $R/tools/pmc_run.sh ${tag}_pmc_sq_attn attn_ "$ctr" -- $R/tools/dbg_attn_one.py 197 0
  bash $R/tools/pmc_run.sh ${tag}_pmc_sq_attn attn_ "$ctr" -- $R/tools/dbg_attn_one.py 1025 0
  bash $R/tools/pmc_run.sh ${tag}_pmc_sq_attn attn_ "$ctr" -- $R/tools/dbg_attn_one.py 77 0 bwd
done
