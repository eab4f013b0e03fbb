"""Round 6: the long-sequence forward at SMALL batch (the reference tool's one image per call): two / one query block per wave against the ring kernel."""
import sys
import torch
sys.path.insert(0, "/root/repo")
from simseg_amd import ops


def t(fn, it=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


c = ops.attention_qscale(0.125)
for (T, H) in ((1025, 12), (577, 12), (1025, 6)):
    for B in (1, 2, 4, 8, 16):
        qkv = torch.randn(B, T, 3 * H * 64, device="cuda")
        qkv.view(B, T, 3, H * 64)[:, :, 0] *= c
        qkv = qkv.bfloat16()
        r = {}
        for v in (0, 6):
            ops.set_attention_variant(v)
            r[v] = min(t(lambda: ops.attention_fwd_qscaled(qkv, H)) for _ in range(3))
        ops.set_attention_variant(1)
        r[1] = min(t(lambda: ops.attention_fwd(qkv, H, None)) for _ in range(3))
        ops.set_attention_variant(0)
        print(f"T={T} H={H} B={B:3d}: two query blocks per wave {r[0] * 1e3:7.1f} us | one {r[6] * 1e3:7.1f} us | ring kernel {r[1] * 1e3:7.1f} us", flush=True)
