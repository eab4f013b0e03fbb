#!/usr/bin/env python
"""Per-block timeline of the bf16 attention backward (single-kernel form; SIMSEG_ATTN_VARIANT=3: the resident dK/dV pass): how long a
block waits for its operand copies, computes, and stores."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.lib import call, ptr  # noqa: E402

B, T, H = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 197, 12)
qkv = torch.randn(B, T, 3 * H * 64, device="cuda").bfloat16()
dout = torch.randn(B, T, H * 64, device="cuda").bfloat16()
out, lse = ops.attention_fwd(qkv, H, None, save_lse=True)
for _ in range(3):
    ops.attention_bwd(qkv, out, dout, lse, H, None)
buf = torch.zeros(B * H * 4, device="cuda", dtype=torch.int64)
call("simseg_debug_attn_trace", ptr(buf))
ops.attention_bwd(qkv, out, dout, lse, H, None)
call("simseg_debug_attn_trace", None)
torch.cuda.synchronize()
t = buf.cpu().numpy().reshape(-1, 4)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
print(f"B={B} T={T} H={H}: {B * H} blocks, span {us[:, 3].max():.1f} us; per block: waiting for the operand copies {np.mean(us[:, 1] - us[:, 0]):.2f} us "
      f"(p90 {np.percentile(us[:, 1] - us[:, 0], 90):.2f}), tile loop {np.mean(us[:, 2] - us[:, 1]):.2f} us, stores {np.mean(us[:, 3] - us[:, 2]):.2f} us, "
      f"total {np.mean(us[:, 3] - us[:, 0]):.2f} us")
grid = np.arange(0, us[:, 3].max(), 8.0)
print("active blocks / of which waiting for copies, every 8 us:", [(int(((us[:, 0] <= g) & (us[:, 3] > g)).sum()), int(((us[:, 0] <= g) & (us[:, 1] > g)).sum())) for g in grid[:24]])
