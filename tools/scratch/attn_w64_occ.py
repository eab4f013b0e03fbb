"""Round 6: the long-sequence forward at one block per CU (attention variant 8: 40 000 bytes of unused dynamic LDS per block) against two -
Gaussian and zero operands.  What does the second resident wave of every SIMD add?"""
import os, sys, torch
sys.path.insert(0, "/root/repo")
from simseg_amd import ops

def t(fn, it=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

H = 12
c = ops.attention_qscale(0.125)
for variant in (0, 8, 0, 8):
  ops.set_attention_variant(variant)
  for (B, T) in ((256, 1025), (64, 2305)):
    for zero in (0, 1):
      qkv = torch.randn(B, T, 3 * H * 64, device="cuda")
      qkv.view(B, T, 3, H * 64)[:, :, 0] *= c
      qkv = qkv.bfloat16()
      if zero: qkv.zero_()
      ms = min(t(lambda: ops.attention_fwd_qscaled(qkv, H)) for _ in range(3))
      fl = 4.0 * B * H * T * T * 64
      print(f"{'one block' if variant == 8 else 'two blocks'} per CU: B={B} T={T} {'zeros' if zero else 'gaussian'}: {ms*1e3:.1f} us {fl/ms/1e9:.0f} TFLOP/s = {fl/ms/1e9/2500:.3f}", flush=True)
ops.set_attention_variant(0)
