import torch, sys, os
sys.path.insert(0, "/root/repo")
from simseg_amd import ops
from simseg_amd.lib import call, ptr, raw, stream
def run(tok, k, sliced):
    B, N, P = tok.shape
    emb = torch.empty(B, P, device="cuda"); idx = torch.empty(B, k, P, device="cuda", dtype=torch.int32); norm = torch.empty(B, device="cuda")
    scratch = torch.empty(raw("simseg_topk_pool_workspace_bytes", B, P, k) // 4, device="cuda") if sliced else None
    f = lambda: call("simseg_topk_pool_l2norm_fwd", ptr(tok), ops.dt(tok), None, ptr(emb), ptr(idx), ptr(norm), ptr(scratch), B, N, P, k, 1e-8, 1, stream())
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3, emb
for B, N in ((1, 1024), (63, 1024), (128, 1024), (256, 1024), (512, 196), (512, 77)):
    for dt_ in (torch.float32, torch.bfloat16):
        tok = torch.randn(B, N, 512, device="cuda").to(dt_)
        a, ea = run(tok, 5, True); b, eb = run(tok, 5, False)
        print(f"B={B} N={N} {dt_}: sliced {a:.0f} us, one block per image {b:.0f} us, same={torch.allclose(ea, eb, atol=1e-6)}")
