# same-box A/B: LayerNorm backward of the ViT blocks takes xhat from the saved 16-bit output (default) against from the fp32 input
run() { python bench.py --steps 20 --warmup 5 --no-seg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['ms_per_step'],d['step_model']['final_loss'])"; }
export SIMSEG_BENCH_FP16=0 SIMSEG_BENCH_GELU16_LEG=0
for r in 1 2 3; do
  run xhat_from_output
  SIMSEG_AMD_LN_BWD_FROM_OUTPUT=0 run xhat_from_input
done
