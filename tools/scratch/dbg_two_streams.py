#!/usr/bin/env python
"""Debug: the word-embedding gradient under the two-stream tower schedule."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
os.environ["SIMSEG_AMD_COMPUTE"] = "bf16"
from oracle import simseg_ref as R  # noqa: E402
from simseg_amd import ops, towers  # noqa: E402
from test_gpu_fullsize import _build_vitb  # noqa: E402

torch.set_num_threads(min(32, os.cpu_count() or 8))


def cos(a, b):
    a, b = a.double().flatten().cpu(), b.double().flatten().cpu()
    return float(a @ b / (a.norm() * b.norm() + 1e-300))


MODE = [""]
orig_bwd = towers.BertEmbedFn.backward
info = {}


def patched(ctx, dy):
    if "A" in MODE[0]:
        torch.cuda.synchronize()
    info["stream"] = torch.cuda.current_stream().cuda_stream
    info["dy_ptr"] = dy.data_ptr()
    out = orig_bwd(ctx, dy)
    if "B" in MODE[0]:
        torch.cuda.synchronize()
    if "C" in MODE[0]:
        torch.cuda.synchronize()
        ids, mask, s, mean, rstd, lnw = ctx.saved_tensors
        info["dword_norm_in_bwd"] = float(out[2].norm())
        info["dword_clone"] = out[2].clone()
    return out


towers.BertEmbedFn.backward = staticmethod(patched)

refc = R.init_weights_(R.RefCLIP("vit_base_patch16_224_in21k", "bert-base-uncased", img_size=224), seed=12).eval()
image = torch.randn(6, 3, 224, 224, generator=torch.Generator().manual_seed(21))
ids, mask = R.synthetic_text(6, 77, 30522, seed=22, min_len=8)
want, _, _ = refc.forward_loss_local(image, ids, mask)
want.backward()
rp = dict(refc.named_parameters())
W = "text_encoder.model.model.embeddings.word_embeddings.weight"
for two, mode in (("0", ""), ("1", ""), ("1", "A"), ("1", "B"), ("1", "C"), ("1", "")):
    os.environ["SIMSEG_AMD_TWO_STREAMS"] = two
    MODE[0] = mode
    info.clear()
    mc = _build_vitb(224)
    mc.load_state_dict(refc.state_dict(), strict=False)
    mc = mc.cuda().eval()
    main = torch.cuda.current_stream().cuda_stream
    loss = mc({"image": image.cuda(), "input_ids": ids.cuda(), "attention_mask": mask.cuda()})[0]["nce_loss"]
    loss.backward()
    torch.cuda.synchronize()
    p = dict(mc.named_parameters())[W]
    extra = ""
    if "dword_clone" in info:
        extra = f" clone-in-bwd cos {cos(info['dword_clone'], rp[W].grad):.4f} same_as_final {bool(torch.equal(info['dword_clone'], p.grad))}"
    print(f"two={two} mode={mode!r}: word-emb grad cos {cos(p.grad, rp[W].grad):.4f} norm ratio {float(p.grad.norm().cpu() / rp[W].grad.norm()):.3f} "
          f"bwd stream is main: {info.get('stream') == main}{extra}", flush=True)
    rows = sorted((cos(q.grad, rp[n].grad), n) for n, q in mc.named_parameters() if not n.endswith("key.bias"))
    print("    worst:", [(round(c, 4), n.split("model.model.")[-1]) for c, n in rows[:3]])
    del mc
