"""Round 6: what the attention-probability dropout (HF BertSelfAttention, p = 0.1) costs in the text tower's attention kernels:
the packed ragged form of the step (B = 512 captions, U{8..77} tokens, 12 heads) with and without it."""
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


B, L, H = 512, 77, 12
g = torch.Generator().manual_seed(0)
lens = torch.randint(8, L + 1, (B,), generator=g)
cu = torch.zeros(B + 1, dtype=torch.int32)
cu[1:] = lens.cumsum(0)
rows = int(cu[-1])
rows_p = (rows + 255) // 256 * 256
for dtype in (torch.bfloat16, torch.float16):
    qkv = torch.randn(rows_p, 3 * H * 64, device="cuda").to(dtype)
    cud = cu.cuda()
    for p in (0.1, 0.0):
        out, lse = ops.attention_fwd_rows(qkv, H, cud, L, save_lse=True, drop_seed=7, drop_p=p, n_real=rows)
        do = torch.randn_like(out)
        f = min(t(lambda: ops.attention_fwd_rows(qkv, H, cud, L, save_lse=True, drop_seed=7, drop_p=p, n_real=rows)) for _ in range(3))
        b = min(t(lambda: ops.attention_bwd_rows(qkv, out, do, lse, H, cud, L, drop_seed=7, drop_p=p, n_real=rows)) for _ in range(3))
        print(f"{str(dtype)[6:]} packed captions ({rows} rows) dropout {p}: fwd {f * 1e3:6.1f} us, bwd {b * 1e3:6.1f} us", flush=True)
