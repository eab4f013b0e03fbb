for lib in libsimseg_hip_prev.so libsimseg_hip.so; do
  export SIMSEG_AMD_LIB=$PWD/simseg_amd/$lib
  echo "=== $lib"
  timeout 200 python tools/gemm_bench.py --iters 10 --only tn 2>&1 | grep -v amdgpu | tail -6
  echo "-- act 3 (fc1 fwd)"; timeout 100 python tools/gemm_bench.py --iters 10 --shapes quick --only nt --act 3 2>&1 | grep -v amdgpu | head -1
  echo "-- act 4 + colsum (dpre)"; timeout 100 python tools/gemm_bench.py --iters 10 --shapes quick --only nn --act 4 --colsum 2>&1 | grep -v amdgpu | head -1
  echo "-- act 5 (proj/fc2 fwd, fp32 out + residual)"; timeout 100 python tools/gemm_bench.py --iters 10 --shapes train --only nt --act 5 2>&1 | grep -v amdgpu | sed -n 2p; timeout 100 python tools/gemm_bench.py --iters 10 --shapes quick --only nt --act 5 2>&1 | grep -v amdgpu | sed -n 2p
done
