#!/usr/bin/env python
"""Debug: per-parameter gradient cosine of the HIP BERT tower (bf16) against the fp32 oracle, for several shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SIMSEG_AMD_COMPUTE"] = "bf16"
from oracle import simseg_ref as R  # noqa: E402
from simseg_amd.nn import Bert  # noqa: E402

torch.set_num_threads(min(32, os.cpu_count() or 8))


def cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


ref = R.init_weights_(R.RefBert("bert-base-uncased"), seed=12).eval()
m = Bert("bert-base-uncased")
m.load_state_dict(ref.state_dict(), strict=False)
m = m.cuda().eval()
for (B, L, min_len) in ((6, 77, 8), (6, 25, 3)):
    ids, mask = R.synthetic_text(B, L, 30522, seed=22, min_len=min_len)
    g = torch.randn(B, L, 768, generator=torch.Generator().manual_seed(3)) * mask[:, :, None]
    ref.zero_grad(); m.zero_grad()
    y_ref = ref(ids, mask)
    (y_ref * g).sum().backward()
    y = m(ids.cuda(), mask.cuda()).last_hidden_state
    (y * g.cuda()).sum().backward()
    torch.cuda.synchronize()
    rp = dict(ref.named_parameters())
    rows = []
    for n, p in m.named_parameters():
        if rp[n].grad is None or n.endswith("key.bias"):
            continue
        rows.append((cos(p.grad, rp[n].grad), float(p.grad.float().norm().cpu() / rp[n].grad.norm()), n))
    rows.sort()
    fwd = float((y.float().cpu() - y_ref).norm() / y_ref.norm())
    print(f"B={B} L={L} min_len={min_len}: fwd rel err {fwd:.2e}; worst grads:")
    for r in rows[:8]:
        print(f"    cos {r[0]:.4f} norm ratio {r[1]:.3f}  {r[2]}")


# the whole CLIP model as tests/test_gpu_fullsize.py runs it, with the towers on one stream and on two
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from test_gpu_fullsize import _build_vitb  # noqa: E402
refc = R.init_weights_(R.RefCLIP("vit_base_patch16_224_in21k", "bert-base-uncased", img_size=224), seed=12).eval()
image = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(21))
ids, mask = R.synthetic_text(6, 77, 30522, seed=22, min_len=8)
want, _, _ = refc.forward_loss_local(image, ids, mask)
want.backward()
rp = dict(refc.named_parameters())
for two in ("0", "1", "1", "0"):
    os.environ["SIMSEG_AMD_TWO_STREAMS"] = two
    mc = _build_vitb(224)
    mc.load_state_dict(refc.state_dict(), strict=False)
    mc = mc.cuda().eval()
    loss = mc({"image": image.cuda(), "input_ids": ids.cuda(), "attention_mask": mask.cuda()})[0]["nce_loss"]
    loss.backward()
    torch.cuda.synchronize()
    rows = []
    for n, p in mc.named_parameters():
        if n.endswith("key.bias"):
            continue
        rows.append((cos(p.grad, rp[n].grad), float(p.grad.float().norm().cpu() / rp[n].grad.norm()), n))
    rows.sort()
    print(f"CLIP two_streams={two}: loss {loss.item():.5f} vs {want.item():.5f}; worst grads:")
    for r in rows[:8]:
        print(f"    cos {r[0]:.4f} norm ratio {r[1]:.3f}  {r[2]}")
    del mc
