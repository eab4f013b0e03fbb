run() { python bench.py --no-seg $2 2>/dev/null | tail -1 | python -c "import json,sys;d=json.loads(sys.stdin.read());print('$1',d['ms_per_step'],d['step_model']['final_loss'], (d['step_model'].get('fp16_amp_with_gradscaler') or {}).get('ms_per_step'), (d['roofline'].get('clocks_during_timed_steps') or {}).get('sclk_mhz_avg'))"; }
run default_with_cpu_baseline ""
SIMSEG_BENCH_OPT_STREAMS=0 run one_launch_with_cpu_baseline ""
run default_no_cpu "--no-cpu-baseline"
SIMSEG_BENCH_OPT_STREAMS=0 run one_launch_no_cpu "--no-cpu-baseline"
run default_with_cpu_baseline ""
