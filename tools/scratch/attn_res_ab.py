"""Round 6: timing of the resident attention forward at the training shapes (ViT-B B = 512 T = 197; BERT T = 77 masked + dropout) - for same-box
A/B runs of two builds (SIMSEG_AMD_LIB)."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from simseg_amd import ops


def t(fn, it=30):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


out = []
for dtype in (torch.bfloat16, torch.float16):
    H = 12
    torch.manual_seed(0)
    qkv = torch.randn(512, 197, 3 * H * 64, device="cuda").to(dtype)
    ms = min(t(lambda: ops.attention_fwd(qkv, H, None, save_lse=True)) for _ in range(3))
    out.append(f"{str(dtype)[6:]} T=197: {ms * 1e3:.1f} us")
    qkv = torch.randn(512, 77, 3 * H * 64, device="cuda").to(dtype)
    lens = torch.randint(8, 78, (512,), device="cuda")
    mask = (torch.arange(77, device="cuda")[None] < lens[:, None]).long()
    ms = min(t(lambda: ops.attention_fwd(qkv, H, mask, save_lse=True, drop_seed=3, drop_p=0.1)) for _ in range(3))
    out.append(f"T=77 masked+dropout: {ms * 1e3:.1f} us")
    ms = min(t(lambda: ops.attention_fwd(qkv, H, mask, save_lse=True)) for _ in range(3))
    out.append(f"T=77 masked: {ms * 1e3:.1f} us")
print(" | ".join(out), flush=True)
