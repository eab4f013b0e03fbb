"""Round 6: the ViT-B training shape (B = 512, T = 197) on the resident forward (default) against the long-sequence kernel's two forms."""
import math
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


for dtype in (torch.bfloat16, torch.float16):
    B, T, H = 512, 197, 12
    qkv = torch.randn(B, T, 3 * H * 64, device="cuda").to(dtype)
    ref, lref = ops.attention_fwd(qkv, H, None, save_lse=True)
    for v in (0, 7, 6):
        ops.set_attention_variant(v)
        out, lse = ops.attention_fwd(qkv, H, None, save_lse=True)
        ms = min(t(lambda: ops.attention_fwd(qkv, H, None, save_lse=True)) for _ in range(3))
        ops.set_attention_variant(0)
        byts = B * T * H * 64 * 2 * 4
        print(f"{str(dtype)[6:]} B={B} T={T} variant {v}: {ms * 1e3:7.1f} us  {byts / ms / 1e9:5.2f} TB/s  max diff to resident {(out.float() - ref.float()).abs().max().item():.2e} "
              f"lse {(lse - lref).abs().max().item():.2e}", flush=True)
