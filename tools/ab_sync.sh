# same-box A/B of the N > 1 gradient exchange machinery forced on one rank (no collective): what the hooks, events and copies cost
run() { python bench.py --steps 20 --warmup 5 --no-seg --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['ms_per_step'],d['config'].get('gradient_sync_detail'))"; }
export SIMSEG_BENCH_FP16=0
for r in 1 2; do
  run none
  SIMSEG_BENCH_FORCE_SYNC=1 run bucket_default
  SIMSEG_BENCH_FORCE_SYNC=1 SIMSEG_BENCH_SYNC_ZERO_COPY=1 run bucket_zero_copy
  SIMSEG_BENCH_FORCE_SYNC=1 SIMSEG_GRADSYNC_EVENTS=all run bucket_all_events
  SIMSEG_BENCH_FORCE_SYNC=1 SIMSEG_BENCH_DP=flat run flat
done
