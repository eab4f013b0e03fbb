set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/kernel_roofline.py > gpurun_out/r6_kernel_roofline.txt 2>&1
python bench.py > gpurun_out/r6_a_bench_n1.json 2> gpurun_out/r6_a_bench_n1.log
tail -3 gpurun_out/r6_a_bench_n1.log
SIMSEG_FORCE_COLLECTIVES=1 SIMSEG_BENCH_FORCE_SYNC=1 python bench.py --steps 15 --warmup 4 --no-seg --no-cpu-baseline > gpurun_out/r6_rccl_one_rank_collectives_issued.json 2> gpurun_out/r6_rccl_one_rank.log
SIMSEG_DIST_BACKEND=gloo SIMSEG_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 3 --warmup 2 --pairs-per-gpu 128 --no-cpu-baseline --no-seg > gpurun_out/r6_bringup_ws2_gloo_bucket.json 2> gpurun_out/r6_bringup_ws2.log
bash tools/profile_round.sh r6
ls -la gpurun_out | tail -20
