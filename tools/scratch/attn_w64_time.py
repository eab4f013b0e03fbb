"""Round 6: timing only of the long-sequence forward (qscaled entry) - for same-box A/B runs of two builds (SIMSEG_AMD_LIB)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from simseg_amd import ops


def t(fn, it=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


import os
H = 12
ops.set_attention_variant(int(os.environ.get("ATTN_VARIANT", "0")))
c = ops.attention_qscale(0.125)
out = []
for dtype in (torch.bfloat16, torch.float16):
    for (B, T) in ((16, 1025), (64, 1025), (256, 1025), (256, 577)):
        torch.manual_seed(0)
        qkv = torch.randn(B, T, 3 * H * 64, device="cuda")
        qkv.view(B, T, 3, H * 64)[:, :, 0] *= c
        qkv = qkv.to(dtype)
        ms = min(t(lambda: ops.attention_fwd_qscaled(qkv, H)) for _ in range(3))
        fl = 4.0 * B * H * T * T * 64
        out.append(f"{str(dtype)[6:]} B={B} T={T}: {ms * 1e3:.1f} us {fl / ms / 1e9:.0f} TF")
print(" | ".join(out), flush=True)
