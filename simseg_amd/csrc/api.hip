// Error plumbing, version, and a hardware probe for libsimseg_hip.so.
#include <stdarg.h>

#include "common.h"

thread_local char g_simseg_err[512] = {0};

int simseg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_simseg_err, sizeof(g_simseg_err), fmt, ap);
    va_end(ap);
    return -1;
}

extern "C" const char* simseg_last_error(void) { return g_simseg_err; }
extern "C" int simseg_version(void) { return 100; }

// 1 = bf16 (default), 2 = IEEE fp16: which flavour of the 16-bit kernels the calling thread's next calls run (dtype code 1 = "the selected
// 16-bit type" everywhere in the ABI).  Thread-local like the kernel selectors.
thread_local int g_ss_half = 1;
extern "C" int simseg_set_half_type(int t) {
    if (t != 1 && t != 2) return simseg_set_error("simseg_set_half_type: 1 (bf16) or 2 (fp16), got %d", t);
    g_ss_half = t;
    return 0;
}

// ---- probe: records which LDS element every lane receives from ds_read_b64_tr_b16 ---------------
// LDS holds iota (element i = i) as 16-bit values; lane l supplies the byte address  l * 8.
__global__ void tr16_probe_kernel(int* out) {
    __shared__ __attribute__((aligned(16))) short lds[1024];
    const int l = threadIdx.x;
    for (int i = l; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    typedef s16x4 __attribute__((address_space(3))) * lptr;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(reinterpret_cast<char*>(lds) + l * 8));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}

extern "C" int simseg_debug_tr16_probe(int* out_dev, void* stream) {
    hipLaunchKernelGGL(tr16_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_dev);
    SS_LAUNCH_CHECK("tr16_probe");
    return 0;
}
