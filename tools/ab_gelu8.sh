# same-box A/B: the saved GELU' as the 8-bit image (default) against the 16-bit one
run() { python bench.py --steps 20 --warmup 5 --no-seg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['ms_per_step'],d['step_model']['final_loss'])"; }
export SIMSEG_BENCH_FP16=0
for r in 1 2 3; do
  run gelu_grad_8bit
  SIMSEG_AMD_GELU_GRAD_BITS=16 run gelu_grad_16bit
done
