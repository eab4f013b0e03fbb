import sys, torch
sys.path.insert(0, "/root/repo")
from simseg_amd import ops
def t(fn, it=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
H, T = 12, 1025
for B in (16, 32, 63, 64, 128, 256):
    qkv = torch.randn(B, T, 3 * H * 64, device="cuda").bfloat16()
    ms = t(lambda: ops.attention_fwd(qkv, H, None, scale=0.125))
    fl = 4.0 * B * H * T * T * 64
    print(f"B={B:4d} T=1025 fwd {ms*1e3:8.1f} us  {fl/ms/1e9:6.0f} TFLOP/s  {fl/ms/1e9/2500:.3f} of peak; per window {ms/B*1e3:.2f} us", flush=True)
