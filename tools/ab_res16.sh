# same-box A/B: the ViT residual-stream gradient handed from LayerNorm backward to LayerNorm backward as 16 bits (default) against fp32
run() { python bench.py --steps 20 --warmup 5 --no-seg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['ms_per_step'],d['step_model']['final_loss'])"; }
export SIMSEG_BENCH_FP16=0 SIMSEG_BENCH_GELU16_LEG=0
for r in 1 2 3; do
  run resgrad_16bit
  SIMSEG_AMD_RESGRAD_BITS=32 run resgrad_fp32
done
