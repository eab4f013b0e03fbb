import os, sys
import numpy as np
import torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
os.environ["SIMSEG_AMD_TWO_STREAMS"] = "1"
from oracle import simseg_ref as R
from simseg_amd import towers
from test_gpu_fullsize import _build_vitb, _cos
B, L = 256, 77
torch.manual_seed(5)
m = _build_vitb(224).cuda().eval()
image = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(31)).cuda()
ids, mask = R.synthetic_text(B, L, 30522, seed=32, min_len=8)
batch = {"image": image, "input_ids": ids.cuda(), "attention_mask": mask.cuda()}
grads = {}
for tag, mode, emu in (("fp32", "fp32", False), ("b16", "bf16", False), ("res16", "bf16", True)):
    os.environ["SIMSEG_AMD_COMPUTE"] = mode
    towers._EMU_RES16 = emu
    m.zero_grad(set_to_none=True)
    m(batch)[0]["nce_loss"].backward()
    torch.cuda.synchronize()
    grads[tag] = {n: p.grad.detach().double().cpu() for n, p in m.named_parameters() if p.grad is not None}
rows = []
for n, g in grads["fp32"].items():
    if float(g.norm()) < 1e-6:
        continue
    rows.append((n, 1 - _cos(grads["b16"][n], g), 1 - _cos(grads["res16"][n], g)))
a = np.array([[r[1], r[2]] for r in rows])
print(f"mean 1-cos: fp32 residual-gradient stream {a[:,0].mean():.4e}, 16-bit stream {a[:,1].mean():.4e}; max {a[:,0].max():.4e} / {a[:,1].max():.4e}")
worst = sorted(rows, key=lambda r: r[2] - r[1], reverse=True)[:6]
for n, c0, c1 in worst:
    print(f"  {n:<60} {c0:.3e} -> {c1:.3e}")
# early vs late layers of the image tower
for lay in (0, 5, 11):
    sel = [(c0, c1) for n, c0, c1 in rows if f"blocks.{lay}." in n]
    print(f"  ViT block {lay}: mean {np.mean([s[0] for s in sel]):.3e} -> {np.mean([s[1] for s in sel]):.3e}")
