"""End-to-end parity of the drop-in CLIPModel on MI355X against (a) the golden vectors captured from the reference
itself and (b) the CPU oracle on the same seeded inputs.

Tolerances (north star: outputs within 1e-3 fp32):
  * fp32 mode (what the eval tools run): 1e-3 absolute on every reference-method output (measured ~1e-5);
  * bf16 compute mode (training, the reference's autocast path): unit-norm embeddings within 2e-2, loss within 2e-2
    relative, parameter gradients with cosine similarity >= 0.99 to the fp32 reference gradients."""
import os

import numpy as np
import pytest
import torch

from conftest import REPO, tt

pytestmark = pytest.mark.gpu

TINY = ["transforms.input_size=96", "model.image_encoder.tag=vit_test_patch16", "model.image_encoder.embedding_dim=128",
        "model.image_encoder.pretrained=False", "model.text_encoder.tag=bert-test", "model.text_encoder.embedding_dim=128",
        "model.text_encoder.pretrained=False"]


def _build(golden, extra=()):
    from simseg.core.config import update_cfg
    from simseg.models import PIPELINE
    from simseg.tasks.clip.config import task_cfg_init_fn, update_clip_config
    from simseg.utils import build_from_cfg
    cfg = update_cfg(task_cfg_init_fn, os.path.join(REPO, "configs/clip/simseg.vit-s.yaml"), TINY + list(extra), update_clip_config)
    model = build_from_cfg(cfg.model.name, cfg, PIPELINE)
    g = golden("clip_glue")
    sd = {k[3:]: tt(g[k]) for k in g.files if k.startswith("sd.")}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing)
    return model.cuda()


def _maxerr(a, b):
    return (a.float().cpu() - b.float().cpu()).abs().max().item()


def test_clip_methods_fp32_vs_reference_golden(golden, monkeypatch):
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    g = golden("clip_glue")
    m = _build(golden)
    m.eval()
    image, ids, mask = tt(g["image"]).cuda(), tt(g["input_ids"]).cuda(), tt(g["attention_mask"]).cuda()
    with torch.no_grad():
        f = m.forward_image_feature(image)
        assert _maxerr(f, tt(g["img_feat"])) < 1e-3
        assert _maxerr(m.image_projection(f), tt(g["img_tok"])) < 1e-3
        assert _maxerr(m.forward_image_project(f), tt(g["img_emb"])) < 1e-3
        t = m.forward_text_feature(ids, mask)
        assert _maxerr(t, tt(g["txt_feat"])) < 1e-3
        assert _maxerr(m.forward_text_project(t, mask), tt(g["txt_emb"])) < 1e-3
        both = m({"image": image, "input_ids": ids, "attention_mask": mask}, embeddings="all")
        assert _maxerr(both[0], tt(g["img_emb"])) < 1e-3 and _maxerr(both[1], tt(g["txt_emb"])) < 1e-3
        print("fp32 max errs:", _maxerr(f, tt(g["img_feat"])), _maxerr(t, tt(g["txt_feat"])), _maxerr(both[0], tt(g["img_emb"])))


def test_clip_methods_bf16(golden, monkeypatch):
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    g = golden("clip_glue")
    m = _build(golden)
    m.eval()
    batch = {"image": tt(g["image"]).cuda(), "input_ids": tt(g["input_ids"]).cuda(), "attention_mask": tt(g["attention_mask"]).cuda()}
    with torch.no_grad():
        img, txt = m(batch, embeddings="all")
    assert _maxerr(img, tt(g["img_emb"])) < 2e-2 and _maxerr(txt, tt(g["txt_emb"])) < 2e-2
    # autocast is the reference's switch into 16-bit compute (clip_runner.py:226-228)
    monkeypatch.delenv("SIMSEG_AMD_COMPUTE")
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        img2, _ = m(batch, embeddings="all")
    assert torch.equal(img, img2)


def test_bert_tower_vs_hf_golden(golden, monkeypatch):
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    from simseg_amd.nn import Bert
    g = golden("bert_tiny")
    m = Bert("bert-test")
    missing, unexpected = m.load_state_dict({k[3:]: tt(g[k]) for k in g.files if k.startswith("sd.")}, strict=False)
    assert not unexpected
    m = m.cuda().eval()
    for tag in ("a", "b"):      # L = 25 (reference max_length) and L = 77 (BASELINE.json), ragged masks
        with torch.no_grad():
            y = m(tt(g[f"ids_{tag}"]).cuda(), tt(g[f"mask_{tag}"]).cuda()).last_hidden_state
        assert _maxerr(y, tt(g[f"out_{tag}"])) < 1e-3


def _cos(a, b):
    a, b = a.float().flatten().cpu(), b.float().flatten().cpu()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-30))


def test_vit_tower_vs_hf_golden(golden, monkeypatch):
    """The HIP ViT tower in fp32 mode against outputs of transformers' ViTModel (fixture vit_hf_tiny): <= 1e-3 (north star)."""
    from simseg_amd import nn as snn
    from simseg_amd.towers import vit_forward
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    g = golden("vit_hf_tiny")
    vit = snn.ViT("vit_test_patch16", img_size=96)
    sd = {k[3:]: tt(g[k]) for k in g.files if k.startswith("sd.")}
    missing, unexpected = vit.load_state_dict(sd, strict=True)
    vit = vit.cuda().eval()
    for tag in ("a", "b"):
        with torch.no_grad():
            y = vit_forward(vit, tt(g[f"image_{tag}"]).cuda(), torch.float32)
        err = _maxerr(y, tt(g[f"out_{tag}"]))
        assert err < 1e-3, err
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    with torch.no_grad():
        y16 = vit_forward(vit, tt(g["image_a"]).cuda(), torch.bfloat16)
    assert _maxerr(y16, tt(g["out_a"])) < 0.15          # bf16 operands, fp32 residual stream; |y| up to 4.6


def test_train_step_ws1_vs_reference_golden(golden, monkeypatch):
    """forward(batch) -> loss -> backward, world size 1, against the reference's own loss/acc/gradients."""
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    from simseg_amd import towers
    towers.SHADOW_HITS[0] = 0
    g = golden("clip_train_ws1")
    m = _build(golden)
    m.eval()          # the fixture was generated in eval mode (no dropout) so that gradients are comparable
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    loss_dict, a1, a2 = m(batch)
    loss = loss_dict["nce_loss"]
    loss.backward()
    assert abs(loss.item() - float(g["r0.loss"])) < 2e-2 * abs(float(g["r0.loss"]))
    assert abs(a1.item() - float(g["r0.i2t_acc"])) < 1e-6 and abs(a2.item() - float(g["r0.t2i_acc"])) < 1e-6
    from simseg_amd import towers
    assert towers.SHADOW_HITS[0] >= 2         # final LN -> block 1 -> block 0 of the tiny ViT: the bf16 gradient hand-off is the path that ran
    params = dict(m.named_parameters())
    for k in g.files:
        if not k.startswith("r0.grad."):
            continue
        name = k[len("r0.grad."):]
        ours, ref = params[name].grad, tt(g[k])
        c = _cos(ours, ref)
        ratio = float(ours.float().norm().cpu() / (ref.norm() + 1e-30))
        print(f"{name}: cos {c:.5f} norm ratio {ratio:.4f}")
        assert c > 0.99 and 0.9 < ratio < 1.1, (name, c, ratio)


def test_train_mode_dropout_runs_and_differs(golden, monkeypatch):
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    g = golden("clip_train_ws1")
    m = _build(golden)
    m.train()
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    l1 = m(batch)[0]["nce_loss"]
    l2 = m(batch)[0]["nce_loss"]
    l1.backward()
    assert torch.isfinite(l1) and torch.isfinite(l2) and l1.item() != l2.item()     # BERT dropout active (HF default p=0.1)
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_train_step_fp32_exact_vs_reference_golden(golden, monkeypatch):
    """A non-AMP run (cfg.dist.fp16 = False -> plain fp32 loss.backward(), simseg/core/hooks/optimizer.py:76-77): the hand-written
    backward in exact fp32 arithmetic against the REFERENCE's own fp32 gradients of the same batch - every parameter, elementwise."""
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    g = golden("clip_train_ws1")
    m = _build(golden)
    m.eval()          # the fixture was generated in eval mode (no dropout)
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    loss_dict, a1, a2 = m(batch)
    loss = loss_dict["nce_loss"]
    loss.backward()
    np.testing.assert_allclose(loss.item(), float(g["r0.loss"]), rtol=2e-5)
    assert abs(a1.item() - float(g["r0.i2t_acc"])) < 1e-6 and abs(a2.item() - float(g["r0.t2i_acc"])) < 1e-6
    params = dict(m.named_parameters())
    worst = 0.0
    for k in g.files:
        if not k.startswith("r0.grad."):
            continue
        name = k[len("r0.grad."):]
        ours, ref = params[name].grad.float().cpu(), tt(g[k])
        err = float((ours - ref).abs().max() / (ref.abs().max() + 1e-30))
        worst = max(worst, err)
        if "key.bias" in name:       # softmax is invariant to a key bias: the true gradient is 0 and both sides hold rounding noise
            assert float(ours.abs().max()) < 1e-5 * float(params[name.replace("key", "value")].grad.abs().max()) + 1e-9, name
            continue
        assert err < 2e-4, (name, err)
    print("fp32 backward: worst max-abs error relative to the gradient's max", worst)


def test_fp32_train_mode_dropout(golden, monkeypatch):
    """Exact mode with BERT's dropout active: finite, stochastic, and the backward regenerates the forward's masks (a finite
    difference along the gradient direction, same seed, agrees with the analytic directional derivative)."""
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    from simseg_amd import nn as snn
    g = golden("clip_train_ws1")
    m = _build(golden)
    m.train()
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    l1 = m(batch)[0]["nce_loss"]
    l2 = m(batch)[0]["nce_loss"]
    assert torch.isfinite(l1) and torch.isfinite(l2) and l1.item() != l2.item()
    snn.manual_dropout_seed(1234)
    loss = m(batch)[0]["nce_loss"]
    loss.backward()
    w = [p for n, p in m.named_parameters() if n.endswith("encoder.layer.0.intermediate.dense.weight")][0]
    gdir = w.grad / w.grad.norm()
    analytic = float((w.grad * gdir).sum())
    eps = 2e-2
    vals = []
    for sgn in (1.0, -1.0):
        with torch.no_grad():
            w.add_(gdir, alpha=sgn * eps)
        snn.manual_dropout_seed(1234)
        with torch.no_grad():
            vals.append(m(batch)[0]["nce_loss"].item())
        with torch.no_grad():
            w.add_(gdir, alpha=-sgn * eps)
    numeric = (vals[0] - vals[1]) / (2 * eps)
    print("directional derivative: analytic", analytic, "numeric", numeric)
    assert abs(numeric - analytic) < 5e-2 * abs(analytic) + 1e-5


def test_bert_qkv_operand_is_read_in_place_and_gradscaler_protocol(golden, monkeypatch):
    """(a) After the fused AdamW has written its bf16 copies, BERT's [3D, D] query/key/value operand is a view of the optimizer's buffer
    (no per-forward concatenation); before that - and in evaluation - the concatenation is made once and reused.
    (b) The reference's AMP step `scaler.scale(loss).backward(); scaler.step(optimizer); scaler.update()`
    (simseg/core/hooks/optimizer.py:73-82) runs unmodified over this model + optimizer: with bf16 compute the scale is a power of
    two and cancels, so the unscaled gradients equal those of the plain step (to the rounding of the split-K atomics)."""
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    from simseg_amd import towers
    from simseg_amd.optim import AdamW
    g = golden("clip_train_ws1")
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    results = []
    for use_scaler in (False, True):
        m = _build(golden)
        m.eval()
        opt = AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3)
        scaler = torch.amp.GradScaler("cuda", init_scale=65536.0, enabled=use_scaler)
        sa = [mod for n, mod in m.named_modules() if n.endswith("encoder.layer.0.attention.self")][0]
        ws = (sa.query.weight, sa.key.weight, sa.value.weight)
        with torch.no_grad():
            m(batch)
        first = towers._wt_stacked(ws, torch.bfloat16)
        assert towers._wt_stacked(ws, torch.bfloat16) is first                      # evaluation: one concatenation, reused
        for it in range(2):
            opt.zero_grad(set_to_none=True)
            loss = m(batch)[0]["nce_loss"]
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
            if it == 0:     # (compared on the gradients: AdamW's first steps are +-lr whatever the magnitude, so rounding-level noise on
                            #  a near-zero gradient - split-K atomics order - flips whole updates)
                results.append({n: p.grad.detach().clone() for n, p in m.named_parameters()})
            scaler.step(opt)
            scaler.update()
        assert scaler.get_scale() == (65536.0 if use_scaler else 1.0)                  # no inf / nan step was skipped
        w3 = towers._wt_stacked(ws, torch.bfloat16)
        assert w3.data_ptr() == opt.state[sa.query.weight]["p16"].data_ptr() and w3.shape == (3 * ws[0].shape[0], ws[0].shape[1])
        assert torch.equal(w3[ws[0].shape[0]:2 * ws[0].shape[0]], sa.key.weight.detach().bfloat16())      # current values, in place
        assert all(torch.isfinite(p).all() for p in m.parameters())
        torch.cuda.synchronize()
    worst = max(float((results[0][n] - results[1][n]).abs().max() / (results[0][n].abs().max() + 1e-30)) for n in results[0]
                if "key.bias" not in n)
    print("max gradient difference (relative to the tensor's max), GradScaler step vs plain step:", worst)
    assert worst < 1e-4


@pytest.mark.parametrize("Bl,P,smoothing", [(8, 512, 0.0), (512, 512, 0.0), (96, 256, 0.1)])
def test_fused_loss_head_equals_the_two_nce_calls(golden, monkeypatch, Bl, P, smoothing):
    """CLIPModel.forward_loss through the fused head (NCE.both -> heads.ClipLossFn: both directions, one autograd node, 4 + 5 launches)
    against the reference's structure - two NCE calls and 0.5 * (a + b) (pipelines/clip.py:129-140) - on the same embeddings: loss,
    both accuracies and the gradients w.r.t. the image embeddings, the text embeddings and the temperature."""
    from simseg_amd import ops
    m = _build(golden, extra=[f"loss.smoothing={smoothing}"] if smoothing else [])
    nce = m.loss
    assert nce.smoothing == smoothing
    g = torch.Generator(device="cuda").manual_seed(Bl)
    base_i = torch.nn.functional.normalize(torch.randn(Bl, P, device="cuda", generator=g), dim=-1)
    base_t = torch.nn.functional.normalize(base_i + 0.7 * torch.nn.functional.normalize(torch.randn(Bl, P, device="cuda", generator=g), dim=-1), dim=-1)
    res = []
    for fused in ("1", "0"):
        monkeypatch.setenv("SIMSEG_AMD_FUSED_LOSS", fused)
        img, txt = base_i.clone().requires_grad_(True), base_t.clone().requires_grad_(True)
        nce.temperature.grad = None
        n0 = len(ops.LAUNCHES) if hasattr(ops, "LAUNCHES") else 0
        loss_dict, a1, a2 = m.forward_loss(img, txt)
        (loss_dict["nce_loss"] * 3.0).backward()
        res.append((loss_dict["nce_loss"].item(), a1.item(), a2.item(), img.grad.clone(), txt.grad.clone(), nce.temperature.grad.clone()))
    (l1, p1, q1, gi1, gt1, gT1), (l0, p0, q0, gi0, gt0, gT0) = res
    assert abs(l1 - l0) <= 1e-6 * abs(l0) and p1 == p0 and q1 == q0, (l1, l0, p1, p0, q1, q0)
    for a, b, what in ((gi1, gi0, "image embeddings"), (gt1, gt0, "text embeddings"), (gT1, gT0, "temperature")):
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
        assert err < 2e-5, (what, err)


def test_training_step_with_host_caption_lengths_has_no_host_synchronisation(golden, monkeypatch):
    """batch["caption_lengths"] (host numbers, what a loader's tokenizer returned): the text tower sizes its packed rows from them, one
    kernel builds the row maps, and NOTHING in forward + loss + backward reads the device - asserted with torch's sync-debug mode (nonzero /
    item / cpu raise).  Loss and gradients equal those of the same step without the lengths (count read back)."""
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    monkeypatch.setenv("SIMSEG_AMD_TWO_STREAMS", "1")
    from simseg_amd import towers
    g = golden("clip_train_ws1")
    res = []
    for with_lengths in (False, True):
        m = _build(golden).eval()                       # (eval: no dropout, so the two runs are the same function)
        batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
        if with_lengths:
            batch["caption_lengths"] = tt(g["r0.attention_mask"]).sum(1)
        m(batch)[0]["nce_loss"].backward()              # warm-up: weight copies, side stream
        m.zero_grad(set_to_none=True)
        batch = {k: (v.clone() if v.is_cuda else v) for k, v in batch.items()}      # fresh tensors: nothing cached on the mask
        torch.cuda.synchronize()
        if with_lengths:
            torch.cuda.set_sync_debug_mode("error")
        try:
            loss_dict, a1, a2 = m(batch)
            loss_dict["nce_loss"].backward()
        finally:
            torch.cuda.set_sync_debug_mode("default")
        res.append((loss_dict["nce_loss"].item(), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    towers._poll_length_checks(block=True)
    assert abs(res[0][0] - res[1][0]) < 1e-6 * abs(res[0][0])
    for n, gd in res[0][1].items():          # (split-K atomics: the accumulation order of a weight gradient is not reproducible to the bit)
        assert float((gd - res[1][1][n]).abs().max()) <= 1e-3 * float(gd.abs().max()) + 1e-12, n


def test_packed_text_tower_equals_dense(golden, monkeypatch):
    """Ragged captions: inside CLIPModel.forward the text tower drops the padded token rows (GEMMs / LayerNorms on the real tokens only,
    the attention kernels on the dense layout with zero rows put back).  Loss, accuracies and every parameter gradient equal those of the
    dense computation (SIMSEG_AMD_PACKED_TEXT=0), in exact fp32 to rounding; the plain forward_text_feature API stays dense."""
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    from simseg_amd import towers
    g = golden("clip_train_ws1")
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    assert 0 < int(batch["attention_mask"].sum()) < batch["attention_mask"].numel()          # the fixture batch is ragged
    res = {}
    for packed in ("0", "1"):
        monkeypatch.setenv("SIMSEG_AMD_PACKED_TEXT", packed)
        m = _build(golden)
        m.eval()
        calls = []
        orig = towers.ragged_plan
        monkeypatch.setattr(towers, "ragged_plan", lambda mask, *a, **k: (calls.append(1), orig(mask, *a, **k))[1])
        loss_dict, a1, a2 = m(batch)
        loss_dict["nce_loss"].backward()
        monkeypatch.setattr(towers, "ragged_plan", orig)
        assert len(calls) == (1 if packed == "1" else 0)                                       # the packed path is the one that ran
        res[packed] = (loss_dict["nce_loss"].item(), a1.item(), a2.item(), {n: p.grad.clone() for n, p in m.named_parameters()})
        with torch.no_grad():
            feat = m.forward_text_feature(batch["input_ids"], batch["attention_mask"])
        assert _maxerr(feat, tt(g["r0.txt_feat"])) < 1e-3 if "r0.txt_feat" in g.files else True
    np.testing.assert_allclose(res["1"][0], res["0"][0], rtol=1e-6)
    assert res["1"][1:3] == res["0"][1:3]
    for n, gd in res["0"][3].items():
        err = float((res["1"][3][n] - gd).abs().max() / (gd.abs().max() + 1e-30))
        if "key.bias" in n:
            continue
        assert err < 1e-4, (n, err)
    # dropout active (train mode, bf16): runs, finite
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    monkeypatch.setenv("SIMSEG_AMD_PACKED_TEXT", "1")
    m = _build(golden)
    m.train()
    loss = m(batch)[0]["nce_loss"]
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)


def test_full_size_vs_oracle(monkeypatch):
    """ViT-S @224 + BERT-base on config-1-shaped input (4 images, 20 prompts): kernels vs the CPU oracle."""
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    from oracle import simseg_ref as R
    from simseg_amd.nn import Bert, ViT
    torch.manual_seed(0)
    ref = R.init_weights_(R.RefCLIP("vit_small_patch16_224_in21k", "bert-base-uncased", img_size=224), seed=2)
    ref.eval()
    vit, bert = ViT("vit_small_patch16_224_in21k", 224), Bert("bert-base-uncased")
    vit.load_state_dict(ref.vit.state_dict(), strict=False)
    bert.load_state_dict(ref.bert.state_dict(), strict=False)
    vit, bert = vit.cuda().eval(), bert.cuda().eval()
    image = torch.randn(4, 3, 224, 224, generator=torch.Generator().manual_seed(0))
    ids, mask = R.synthetic_text(20, 25, 30522, seed=1, min_len=3)
    with torch.no_grad():
        want_i = ref.vit(image)
        want_t = ref.bert(ids, mask)
        got_i = vit(image.cuda())
        got_t = bert(ids.cuda(), mask.cuda()).last_hidden_state
    assert _maxerr(got_i, want_i) < 1e-3 and _maxerr(got_t, want_t) < 1e-3
    print("full-size fp32 max errs:", _maxerr(got_i, want_i), _maxerr(got_t, want_t))


def test_retrieval_metric_mirror_vs_reference_golden(golden):
    """IndexedEmbInfo.unique + RetrievalMetric both directions == the reference's numbers (hooks/utils.py)."""
    from simseg.tasks.clip.hooks.utils import IndexedEmbInfo, RetrievalMetric
    g = golden("retrieval")
    img = IndexedEmbInfo("image", tt(g["gid_rows"]).cuda(), tt(g["img_rows"]).cuda()).unique()
    assert torch.equal(img.group_idx.cpu(), tt(g["uni_gid"])) and torch.equal(img.emb_mat.cpu(), tt(g["uni_emb"]))
    txt = IndexedEmbInfo("text", tt(g["gid_txt"]).cuda(), tt(g["txt"]).cuda())
    m = RetrievalMetric()
    from simseg_amd import ops
    ops.PROFILE = []                       # every simseg_gemm launch is recorded
    try:
        i2t = m(img, txt)
        n_first = len(ops.PROFILE)
        t2i = m(txt, img)                  # the swapped call of tools/retrieval_evaluation.py:44-45: answered from the first call's matrix
        n_second = len(ops.PROFILE) - n_first
        again = m(txt, img)                # a memo entry is used once: an unpaired call computes
        n_third = len(ops.PROFILE) - n_first - n_second
    finally:
        ops.PROFILE = None
    assert (n_first, n_second, n_third) == (1, 0, 1), (n_first, n_second, n_third)
    assert again == t2i
    np.testing.assert_allclose([i2t[f"[image] to [text]: R@{k}"] for k in (1, 5, 10)], g["i2t"], atol=1e-7)
    np.testing.assert_allclose([t2i[f"[text] to [image]: R@{k}"] for k in (1, 5, 10)], g["t2i"], atol=1e-7)
    # tensors changed in place between the two calls miss the memo (version counter) and are recomputed on the new values
    m2 = RetrievalMetric()
    m2(img, txt)
    txt.emb_mat.mul_(1.0)
    ops.PROFILE = []
    try:
        r = m2(txt, img)
        assert len(ops.PROFILE) == 1
    finally:
        ops.PROFILE = None
    assert r == t2i


def test_seg_similarity_map_vs_reference_golden(golden, monkeypatch):
    """Dense patch x class-text similarity (tools/seg_evaluation.py:112,136) for all classes at once."""
    from simseg_amd.heads import patch_text_similarity
    g = golden("seg_block")
    sim = patch_text_similarity(tt(g["proj"]).cuda(), tt(g["text"]).cuda())           # [B, N, C]
    maps = sim.transpose(1, 2).reshape(2, 21, 18, 18)
    assert _maxerr(maps, tt(g["maps"])) < 1e-5
    sim16 = patch_text_similarity(tt(g["proj"]).cuda(), tt(g["text"]).cuda(), compute_dtype=torch.bfloat16)
    assert _maxerr(sim16, sim) < 2e-2


def test_class_text_embeddings_prompt_ensemble(golden, monkeypatch):
    """Batched zero-shot classifier == the per-class loop of tools/seg_evaluation.py:57-75 evaluated with the oracle."""
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    from oracle import simseg_ref as R
    from simseg_amd.heads import class_text_embeddings
    m = _build(golden).eval()
    g = golden("clip_glue")
    ref = R.RefCLIP("vit_test_patch16", "bert-test", img_size=96)
    ref.load_state_dict({k[3:]: tt(g[k]) for k in g.files if k.startswith("sd.")}, strict=False)
    ref.eval()
    C, P, L = 5, 7, 25
    ids, mask = R.synthetic_text(C * P, L, 1000, seed=9)
    got = class_text_embeddings(m, ids.view(C, P, L).cuda(), mask.view(C, P, L).cuda(), chunk=16)
    want = []
    with torch.no_grad():
        for c in range(C):
            sl = slice(c * P, (c + 1) * P)
            e = ref.forward_text_project(ref.forward_text_feature(ids[sl], mask[sl]), mask[sl]).mean(dim=0)
            want.append(e / e.norm())
    assert _maxerr(got, torch.stack(want)) < 1e-3


def test_trainer_iteration_and_checkpoint_roundtrip(golden, tmp_path):
    """A few reference-style training iterations (stateless LR, autocast forward, AdamW) reduce the loss on a fixed batch;
    the checkpoint has the reference's layout and restores model + optimizer exactly."""
    from simseg_amd.trainer import Trainer
    g = golden("clip_train_ws1")
    m = _build(golden, ["epoch=1", "optim.lr.init=1e-3"])
    cfg = m.cfg
    tr = Trainer(m, cfg, steps_per_epoch=40)
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    m.eval()                               # deterministic (no dropout) so that the resume check below is exact
    losses = [float(tr.train_step(batch)["loss"]) for _ in range(12)]
    assert losses[-1] < losses[0] - 0.05, losses
    assert abs(tr.optimizer.param_groups[0]["lr"] - 1e-3 * (0.1 + 0.9 * 0.5 * (1 + np.cos(np.pi * (11 - 1) / 39)))) < 1e-9
    ck = tr.checkpoint()
    assert set(ck) == {"state_dict", "optimizer", "meta", "scaler"} and ck["meta"]["step"] == 12
    assert "image_encoder.model.model.blocks.0.attn.qkv.weight" in ck["state_dict"] and "loss.temperature" in ck["state_dict"]
    path = tmp_path / "step_checkpoint.pth"
    torch.save(ck, path)
    nxt = float(tr.train_step(batch)["loss"])
    m2 = _build(golden, ["epoch=1", "optim.lr.init=1e-3"])
    m2.eval()
    tr2 = Trainer(m2, m2.cfg, steps_per_epoch=40)
    tr2.load_checkpoint(torch.load(path, weights_only=False))
    assert tr2.step == 12 and tr2.optimizer._step == 12              # the bias-correction counter travels with the checkpoint
    assert abs(float(tr2.train_step(batch)["loss"]) - nxt) < 1e-6
    # the step AFTER the resume depends on the restored moments and step counter: the resumed run stays on the original one
    after, after2 = float(tr.train_step(batch)["loss"]), float(tr2.train_step(batch)["loss"])
    assert abs(after - after2) < 1e-5 * max(1.0, abs(after)), (after, after2)
    for (n, a), (_, b) in zip(m.named_parameters(), m2.named_parameters()):
        assert _maxerr(a, b) <= 1e-6 * (1 + float(a.abs().max())), n
    # torch.optim.AdamW's state layout: the reference's optimizer checkpoints load here and ours load there
    st = ck["optimizer"]["state"]
    assert all(set(v) == {"step", "exp_avg", "exp_avg_sq"} for v in st.values()) and len(st) == len(list(m.parameters()))
    topt = torch.optim.AdamW([{"params": [p]} for p in m.parameters()], lr=1e-3)
    topt.load_state_dict(ck["optimizer"])
    assert float(next(iter(topt.state.values()))["step"]) == 12.0
    # one launch for all ~100 single-parameter groups (the reference's ClipOptimizerHook layout)
    assert len(tr.optimizer.param_groups) > 50 and len(tr.optimizer._plans) == 1


def test_bf16_weight_copy_cache_coherence(golden, monkeypatch):
    """The towers reuse a cached bf16 copy of each weight; every way the reference's code paths change a weight must be seen:
    the fused AdamW kernel (refreshes the copy itself), load_state_dict / copy_ (version bump)."""
    from simseg_amd import towers
    from simseg_amd.optim import AdamW
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    g = golden("clip_train_ws1")
    m = _build(golden)
    m.eval()
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    w = m.image_encoder.model.model.blocks[0].mlp.fc1.weight
    e0 = m(batch, embeddings="all")[0].clone()
    assert torch.equal(towers._wt(w, torch.bfloat16), w.detach().bfloat16())
    c0 = towers._wt(w, torch.bfloat16)
    assert towers._wt(w, torch.bfloat16) is c0                     # second use: no cast
    with torch.no_grad():
        w.mul_(0.5)                                                # torch-visible write
    assert torch.equal(towers._wt(w, torch.bfloat16), w.detach().bfloat16())
    e1 = m(batch, embeddings="all")[0]
    assert (e1 - e0).abs().max() > 1e-4
    opt = AdamW(m.parameters(), lr=1e-2)
    loss = m(batch)[0]["nce_loss"]
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    for p in (w, m.text_encoder.model.model.encoder.layer[0].attention.self.query.weight, m.image_projection.linear.weight):
        ent = towers._W16[id(p)]
        assert ent[3].data_ptr() == opt.state[p]["p16"].data_ptr()      # the optimizer's copy is the one the towers use
        assert torch.equal(towers._wt(p, torch.bfloat16), p.detach().bfloat16())
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m.load_state_dict({k: v * 0 + 0.01 for k, v in sd.items()}, strict=False)
    assert torch.equal(towers._wt(w, torch.bfloat16), w.detach().bfloat16())


def test_two_stream_towers_match_single_stream(golden, monkeypatch):
    """The text tower on a second HIP stream (default outside DDP training) must not change anything: same loss trajectory and
    gradients as the single-stream schedule over several optimizer steps (races would show up as drift)."""
    from simseg_amd.optim import AdamW
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    g = golden("clip_train_ws1")
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    runs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("SIMSEG_AMD_TWO_STREAMS", mode)
        m = _build(golden)
        m.eval()                                   # no dropout: the two schedules are comparable step by step
        opt = AdamW(m.parameters(), lr=1e-3)
        losses = []
        for _ in range(6):
            opt.zero_grad(set_to_none=True)
            loss = m(batch)[0]["nce_loss"]
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        runs[mode] = (losses, {n: p.detach().clone() for n, p in m.named_parameters()})
    l0, l1 = runs["0"][0], runs["1"][0]
    assert max(abs(a - b) for a, b in zip(l0, l1)) < 2e-3 * max(abs(x) for x in l0), (l0, l1)
    assert l0[-1] < l0[0]
    worst = max(float((runs["0"][1][n] - runs["1"][1][n]).abs().max() / (runs["0"][1][n].abs().max() + 1e-12)) for n in runs["0"][1])
    assert worst < 5e-2, worst                     # bf16 compute + fp32 atomics ordering; a race would be O(1)


def test_graphed_forward_matches_eager(golden, monkeypatch):
    """hipGraph replay of the per-image segmentation forward (towers -> projection -> similarity map -> candidate selection)
    returns what the eager launches return, for inputs it was not captured with."""
    from simseg_amd.graph import GraphedCall
    from simseg_amd.heads import patch_text_similarity
    from simseg_amd import ops
    for mode in ("fp32", "bf16"):
        monkeypatch.setenv("SIMSEG_AMD_COMPUTE", mode)
        m = _build(golden).eval()
        g = torch.Generator().manual_seed(3)
        text = torch.nn.functional.normalize(torch.randn(21, 512, generator=g), dim=-1).cuda()

        def pipeline(image):
            feats = m.forward_image_feature(image)
            pooled = m.forward_image_project(feats)
            sim = patch_text_similarity(m.image_projection(feats), text)
            idx, score, thr = ops.seg_select(ops.gemm(pooled.float(), text), 10, 5)
            return sim, idx, score

        imgs = [torch.randn(1, 3, 96, 96, generator=g).cuda() for _ in range(3)]
        gc = GraphedCall(pipeline, imgs[0])
        for im in imgs[1:]:
            with torch.no_grad():
                want = [t.clone() for t in pipeline(im)]
            got = gc(im)
            torch.cuda.synchronize()
            assert _maxerr(got[0], want[0]) < 1e-6 and torch.equal(got[1], want[1]) and _maxerr(got[2], want[2]) < 1e-6, mode


def test_training_trajectory_follows_oracle(golden, monkeypatch):
    """40 AdamW steps on a fixed batch (tiny towers, eval mode = no dropout): the bf16 HIP path's loss curve stays on the curve the
    fp32 CPU oracle produces with torch.optim.AdamW from the same initial weights, and both memorise the batch."""
    from oracle import simseg_ref as R
    from simseg_amd.optim import AdamW
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    g = golden("clip_train_ws1")
    gw = golden("clip_glue")
    ref = R.RefCLIP("vit_test_patch16", "bert-test", img_size=96)
    ref.load_state_dict({k[3:]: tt(gw[k]) for k in gw.files if k.startswith("sd.")}, strict=False)
    ref.eval()
    m = _build(golden)
    m.eval()
    image, ids, mask = tt(g["r0.image"]), tt(g["r0.input_ids"]), tt(g["r0.attention_mask"])
    batch = {"image": image.cuda(), "input_ids": ids.cuda(), "attention_mask": mask.cuda()}
    hp = dict(lr=5e-4, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3)
    opt = AdamW(m.parameters(), **hp)
    opt_ref = torch.optim.AdamW(ref.parameters(), **hp)
    ours, want = [], []
    torch.set_num_threads(8)
    for _ in range(40):
        opt.zero_grad(set_to_none=True)
        loss = m(batch)[0]["nce_loss"]
        loss.backward()
        opt.step()
        ours.append(float(loss.detach()))
        opt_ref.zero_grad(set_to_none=True)
        lr_, _, _ = ref.forward_loss_local(image, ids, mask)
        lr_.backward()
        opt_ref.step()
        want.append(float(lr_.detach()))
    print("ours", [round(x, 3) for x in ours[::5]], "oracle", [round(x, 3) for x in want[::5]])
    assert want[-1] < 0.5 * want[0], want                      # the oracle memorises the batch ...
    assert ours[-1] < 0.5 * ours[0], ours                      # ... and so does the HIP path
    assert abs(ours[0] - want[0]) < 2e-2 * want[0]
    for a, b in zip(ours[:10], want[:10]):                     # early steps: same curve (later ones diverge chaotically in any precision)
        assert abs(a - b) < 0.1 * max(abs(b), 0.1), (ours[:10], want[:10])


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bf16", "fp32"])
def test_fused_image_head_equals_slice_project_pool(golden, monkeypatch, mode):
    """CLIPModel._image_embeddings hands the image tower's WHOLE output to ProjectPoolFn (skip=1: the [cls] token is masked out of the top-k
    pooling; the 16-bit copy written by the tower's last LayerNorm is the GEMM operand) instead of slicing feats[:, 1:] first (clip.py:65-84,
    87-93).  Same embeddings bit for bit; same gradients up to the order of the weight-gradient atomics."""
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", mode)
    g = golden("clip_train_ws1")
    image = tt(g["r0.image"]).cuda()
    res = {}
    for fused in ("1", "0"):
        monkeypatch.setenv("SIMSEG_AMD_FUSED_IMAGE_HEAD", fused)
        m = _build(golden).eval()
        emb = m._image_embeddings(image)
        (emb * torch.linspace(-1, 1, emb.numel(), device="cuda").view_as(emb)).sum().backward()
        torch.cuda.synchronize()
        res[fused] = (emb.detach().clone(), {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
    assert torch.equal(res["1"][0], res["0"][0])
    assert set(res["1"][1]) == set(res["0"][1]) and len(res["1"][1]) > 10
    for n, ga in res["1"][1].items():
        gb = res["0"][1][n]
        assert float((ga - gb).abs().max()) <= 1e-5 * float(gb.abs().max()) + 1e-12, n
    # ... and it is what forward() runs
    m = _build(golden).eval()
    monkeypatch.setenv("SIMSEG_AMD_FUSED_IMAGE_HEAD", "1")
    from simseg_amd import towers
    hits = []
    real = towers.ProjectPoolFn.forward
    monkeypatch.setattr(towers.ProjectPoolFn, "forward", staticmethod(lambda ctx, *a: (hits.append(a[5] if len(a) > 5 else 0), real(ctx, *a))[1]))
    batch = {"image": image, "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    m(batch)
    assert 1 in hits


@pytest.mark.gpu
def test_text_tower_update_on_its_own_stream_matches_single_launch(golden, monkeypatch):
    """AdamW.set_param_streams: the text tower's parameters are updated on the text tower's stream (right behind its backward) instead of in
    the one launch behind everything.  Same trajectory as the single launch over several steps, with fresh batches' memory recycled in
    between (a missing ordering would show up as drift or garbage), and with a single-stream step in the middle (param_streams=False)."""
    from simseg.models.pipelines.clip import _side_stream
    from simseg_amd.optim import AdamW
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "bf16")
    monkeypatch.setenv("SIMSEG_AMD_TWO_STREAMS", "1")
    g = golden("clip_train_ws1")
    base = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    runs = {}
    for own in (False, True):
        m = _build(golden).eval()
        opt = AdamW(m.parameters(), lr=1e-3)
        if own:
            opt.set_param_streams({p: _side_stream(torch.device("cuda", 0)) for p in list(m.text_encoder.parameters()) + list(m.text_projection.parameters())})
        losses = []
        for it in range(7):
            batch = {k: v.clone() for k, v in base.items()}
            if it == 3:
                monkeypatch.setenv("SIMSEG_AMD_TWO_STREAMS", "0")
            opt.zero_grad(set_to_none=True)
            loss = m(batch)[0]["nce_loss"]
            loss.backward()
            opt.step(param_streams=m.two_streams_used)
            if it == 3:
                assert not m.two_streams_used
                monkeypatch.setenv("SIMSEG_AMD_TWO_STREAMS", "1")
            losses.append(float(loss.detach()))
            del batch
            junk = torch.full((1 << 22,), float("nan"), device="cuda")      # recycled memory is poisoned
            del junk
        torch.cuda.synchronize()
        runs[own] = (losses, {n: p.detach().clone() for n, p in m.named_parameters()})
    l0, l1 = runs[False][0], runs[True][0]
    assert all(np.isfinite(l1)) and max(abs(a - b) for a, b in zip(l0, l1)) < 2e-3 * max(abs(x) for x in l0), (l0, l1)
    worst = max(float((runs[False][1][n] - runs[True][1][n]).abs().max() / (runs[False][1][n].abs().max() + 1e-12)) for n in runs[False][1])
    assert worst < 5e-2, worst
