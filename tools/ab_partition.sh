# CU partition experiment: weight-gradient GEMMs on a side stream with a capped block count, persistent GEMMs leaving CUs free
run() { python bench.py --steps 12 --warmup 4 --no-seg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['ms_per_step'], d['step_model']['final_loss'])"; }
export SIMSEG_BENCH_FP16=0
run base
SIMSEG_AMD_WGRAD_STREAM=1 run wgstream
for rb in "64 64" "96 96" "128 128" "112 108" "160 160" "96 128"; do
  set -- $rb
  SIMSEG_AMD_WGRAD_STREAM=1 SIMSEG_GEMM_PP2_RESERVE=$1 SIMSEG_AMD_WGRAD_BLOCKS=$2 run "wgstream_reserve$1_wgblocks$2"
done
SIMSEG_GEMM_PP2_RESERVE=96 run reserve96_only
SIMSEG_AMD_WGRAD_BLOCKS=128 run wgblocks128_only
run base
