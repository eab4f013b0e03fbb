run() { python bench.py --steps 20 --warmup 5 --no-seg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['ms_per_step'],d['step_model']['final_loss'])"; }
export SIMSEG_BENCH_FP16=0
for r in 1 2 3; do
  run opt_streams
  SIMSEG_BENCH_OPT_STREAMS=0 run one_launch
done
