# same-box A/B of two builds of the library on the GELU epilogue GEMMs (fc1 forward, dgrad through fc2), three interleaved rounds
for r in 1 2 3; do
for lib in libsimseg_hip_ab.so libsimseg_hip.so   # (build the other tree into simseg_amd/libsimseg_hip_ab.so first; it is gpurun-ignored - copy it under another name to ship it); do
  export SIMSEG_AMD_LIB=$PWD/simseg_amd/$lib
  echo "=== $lib round $r"
  echo -n "act 3 (fc1 fwd, row-major GELU')   "; timeout 100 python tools/gemm_bench.py --iters 20 --shapes quick --only nt --act 3 2>&1 | grep -v amdgpu | head -1
  echo -n "act 5 (fc1 fwd, tile-blocked GELU') "; timeout 100 python tools/gemm_bench.py --iters 20 --shapes quick --only nt --act 7 2>&1 | grep -v amdgpu | head -1
  echo -n "act 6 (dgrad fc2 x blocked GELU')   "; timeout 100 python tools/gemm_bench.py --iters 20 --shapes quick --only nn --act 8 --colsum 2>&1 | grep -v amdgpu | head -1
  echo -n "act 0 (plain qkv fwd)               "; timeout 100 python tools/gemm_bench.py --iters 20 --shapes train --only nt 2>&1 | grep -v amdgpu | head -1
done
done
