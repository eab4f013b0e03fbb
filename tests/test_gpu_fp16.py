"""The fp16 flavour of the 16-bit kernels - the type of the reference's AMP mode (torch.cuda.amp.autocast() + GradScaler,
simseg/tasks/clip/clip_runner.py:226-230, simseg/core/hooks/optimizer.py:73-82).  Same sources as the bf16 kernels (compiled with
-DSS_HALF: v_mfma_f32_32x32x16_f16), selected per call by the tensors' dtype.  Kernels against fp64 / fp32 references; the model's
fp16 gradients against its exact-fp32 gradients; the trainer's live loss scaling (scaled backward, unscale, overflow -> skipped step +
back-off, growth, checkpointed state)."""
import os

import numpy as np
import pytest
import torch

from conftest import tt
from test_gpu_model import _build

pytestmark = pytest.mark.gpu
F16, BF16 = torch.float16, torch.bfloat16


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm() + 1e-300))


@pytest.mark.parametrize("kind,M,N,K", [("nt", 4096, 768, 768), ("nn", 1024, 3072, 768), ("tn", 768, 768, 8192), ("nt", 333, 130, 72), ("nt", 25856, 2304, 768)])
def test_gemm_fp16_flavour(kind, M, N, K):
    from simseg_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    a32 = torch.randn((K, M) if kind == "tn" else (M, K), device="cuda", generator=g)
    b32 = torch.randn((K, N) if kind in ("nn", "tn") else (N, K), device="cuda", generator=g) * 0.1
    err = {}
    for hd in (BF16, F16):
        a, b = a32.to(hd), b32.to(hd)
        A = a.double().T if kind == "tn" else a.double()
        B = b.double() if kind in ("nn", "tn") else b.double().T
        want = A @ B
        if kind == "tn":
            out = torch.zeros(M, N, device="cuda")
            ops.gemm(a, b, trans_a=True, trans_b=True, out=out, accumulate=True, splitk=4)
        else:
            out = ops.gemm(a, b, trans_b=(kind == "nn"), out_dtype=torch.float32)
        e32 = float((out.double() - want).abs().max() / want.abs().max())
        assert e32 < 1e-5, (hd, e32)                             # fp32 accumulation of exact 16-bit products
        if kind != "tn":
            o16 = ops.gemm(a, b, trans_b=(kind == "nn"))         # 16-bit output: one rounding of the result
            assert o16.dtype == hd
            err[hd] = float((o16.double() - want).abs().max() / want.abs().max())
    if err:
        assert err[F16] < 1e-3 and err[BF16] < 8e-3 and err[F16] < err[BF16], err      # 11 vs 8 significant bits


def test_fp16_values_beyond_bf16_precision_survive():
    """The two flavours really are different types: 1 + 2^-10 is an fp16 number and not a bf16 one."""
    from simseg_amd import ops
    x = torch.full((256, 64), 1.0 + 2.0 ** -10, device="cuda")
    eye = torch.eye(64, device="cuda")
    for hd, exact in ((F16, True), (BF16, False)):
        h = ops.cast(x, hd)
        assert h.dtype == hd and torch.equal(h, x.to(hd))
        y = ops.gemm(h, eye.to(hd), out_dtype=torch.float32)
        assert bool((y == 1.0 + 2.0 ** -10).all()) == exact
        assert torch.equal(ops.cast(h, torch.float32), h.float())


@pytest.mark.parametrize("T,drop", [(197, 0.0), (77, 0.1), (300, 0.0)])
def test_attention_fp16_fwd_bwd(T, drop):
    from simseg_amd import ops
    B, H = 3, 4
    g = torch.Generator(device="cuda").manual_seed(1)
    qkv32 = torch.randn(B, T, 3 * H * 64, device="cuda", generator=g)
    do32 = torch.randn(B, T, H * 64, device="cuda", generator=g)
    mask = None
    if T == 77:
        lens = torch.tensor([77, 40, 9], device="cuda")
        mask = (torch.arange(T, device="cuda")[None] < lens[:, None]).long()
    outs = {}
    for hd in (BF16, F16):
        qkv = qkv32.to(hd)
        out, lse = ops.attention_fwd(qkv, H, mask, scale=0.125, save_lse=True, drop_seed=7, drop_p=drop)
        dqkv = ops.attention_bwd(qkv, out, do32.to(hd), lse, H, mask, scale=0.125, drop_seed=7, drop_p=drop)
        assert out.dtype == hd and dqkv.dtype == hd
        outs[hd] = (out.float(), dqkv.float())
    if drop == 0.0:
        q, k, v = [t.view(B, T, H, 64).transpose(1, 2).double() for t in qkv32.to(F16).chunk(3, -1)]
        s = q @ k.transpose(-1, -2) * 0.125
        if mask is not None:
            s = s.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
        want = (s.softmax(-1) @ v).transpose(1, 2).reshape(B, T, H * 64)
        assert float((outs[F16][0].double() - want).abs().max()) < 3e-3
    # both flavours compute the same function of (their rounding of) the same inputs: bf16-level agreement, fp16 the finer one
    valid = slice(None) if mask is None else None
    for i in range(2):
        a, b = outs[F16][i], outs[BF16][i]
        if mask is not None:
            keep = mask.bool()[:, :, None].expand_as(a)
            a, b = a[keep], b[keep]
        assert _cos(a, b) > (0.999 if drop == 0.0 else 0.99), (i, _cos(a, b))
    assert torch.isfinite(outs[F16][1]).all()


def test_layernorm_and_rowops_fp16():
    from simseg_amd import ops
    g = torch.Generator(device="cuda").manual_seed(2)
    x = torch.randn(1000, 768, device="cuda", generator=g) * 3 + 0.5
    w, b = torch.randn(768, device="cuda", generator=g), torch.randn(768, device="cuda", generator=g)
    want = torch.nn.functional.layer_norm(x, (768,), w, b, 1e-6)
    y, _, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-6, out_dtype=F16, save_stats=True)
    assert y.dtype == F16 and float((y.float() - want).abs().max()) < 2e-2
    y32, y16, _, _ = ops.layernorm_fwd(x, w, b, 1e-6, want_bf16_copy=F16)
    assert y16.dtype == F16 and torch.equal(y16, y32.to(F16))
    dy = torch.randn(1000, 768, device="cuda", generator=g)
    xr = x.clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xr, (768,), w, b, 1e-6).backward(dy.to(F16).float())
    dg, db = torch.zeros(768, device="cuda"), torch.zeros(768, device="cuda")
    dx32, dx16 = ops.layernorm_bwd(x, mean, rstd, w, dg, db, dy16=dy.to(F16))
    assert dx16.dtype == F16 and float((dx32 - xr.grad).abs().max()) < 1e-3 * float(xr.grad.abs().max()) + 1e-4
    assert torch.equal(dx16, dx32.to(F16))
    s = torch.zeros(768, device="cuda")
    ops.colsum_accum(dy.to(F16), s)
    assert float((s - dy.to(F16).float().sum(0)).abs().max()) < 1e-2


def test_model_fp16_gradients_vs_exact_fp32(golden, monkeypatch):
    """The drop-in CLIPModel under torch.autocast(dtype=float16): loss and every parameter gradient against the exact-fp32 run of the
    same step (the reference's own gradients are pinned against that mode), next to the bf16 autocast run of the same step.  Bars: loss
    within 2e-3; every gradient cosine >= 0.99 (the bf16 bar of tests/test_gpu_model.py) and within 3 % in norm; and - fp16 carries three
    more bits than bf16 - closer to the fp32 gradients than the bf16 run on average."""
    g = golden("clip_train_ws1")
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    m = _build(golden).eval()
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    loss32 = m(batch)[0]["nce_loss"]
    loss32.backward()
    ref = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    monkeypatch.delenv("SIMSEG_AMD_COMPUTE")
    cos, SCALE = {}, 65536.0
    for hd in (torch.bfloat16, torch.float16):
        m.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=hd):
            loss16 = m(batch)[0]["nce_loss"]
        (loss16 * SCALE).backward()                                # the loss scale the reference's GradScaler starts from (fp16 gradients
                                                                   # below 6e-5 are subnormal: at a scale of 1024 the query-weight gradients of this model lose most of their bits)
        l16, l32 = float(loss16.detach()), float(loss32.detach())
        assert abs(l16 - l32) < (2e-3 if hd == torch.float16 else 2e-2) * abs(l32), (hd, l16, l32)
        cos[hd] = {}
        for n, p in m.named_parameters():
            gr = p.grad / SCALE
            assert torch.isfinite(gr).all(), n
            if "key.bias" in n or float(ref[n].abs().max()) < 1e-7:     # (the null space of the loss: pure rounding noise)
                continue
            cos[hd][n] = _cos(gr, ref[n])
            if hd == torch.float16:
                assert cos[hd][n] > 0.99, (n, cos[hd][n])
                assert abs(float(gr.norm() / ref[n].norm()) - 1) < 3e-2, n
    m16, mb = np.mean([1 - c for c in cos[torch.float16].values()]), np.mean([1 - c for c in cos[torch.bfloat16].values()])
    print(f"mean (1 - cosine) to the exact-fp32 gradients: fp16 autocast {m16:.2e} (worst cosine {min(cos[torch.float16].values()):.4f}), "
          f"bf16 autocast {mb:.2e} (worst {min(cos[torch.bfloat16].values()):.4f})")
    assert m16 < mb


def test_trainer_fp16_amp_with_live_gradscaler(golden, tmp_path):
    """The reference's AMP iteration on this engine: fp16 autocast forward, scaler.scale(loss).backward(), scaler.step (unscale, overflow
    check), scaler.update; an overflowing step is skipped and halves the scale; the scale grows after growth_interval clean steps; the
    scaler's state travels in the checkpoint; the 16-bit weight copies the optimizer kernel writes are fp16."""
    from simseg_amd.trainer import Trainer
    g = golden("clip_train_ws1")
    batch = {"image": tt(g["r0.image"]).cuda(), "input_ids": tt(g["r0.input_ids"]).cuda(), "attention_mask": tt(g["r0.attention_mask"]).cuda()}
    m = _build(golden, ["epoch=1", "optim.lr.init=1e-3", "dist.fp16=True"])
    m.eval()
    tr = Trainer(m, m.cfg, steps_per_epoch=40, amp_dtype="fp16")
    assert tr.scaler.is_enabled() and tr.amp_dtype == torch.float16
    tr.scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 14, growth_interval=4)
    losses = [float(tr.train_step(batch)["loss"]) for _ in range(10)]
    assert losses[-1] < losses[0] - 0.05, losses
    assert tr.scaler.get_scale() >= 2.0 ** 15                         # grew at least once over ten clean steps (interval 4)
    p16 = tr.optimizer.state[next(iter(m.parameters()))]["p16"]
    assert p16.dtype == torch.float16
    w = m.image_projection.linear.weight
    assert torch.equal(tr.optimizer.state[w]["p16"], w.detach().to(torch.float16))      # refreshed by the optimizer kernel
    # an overflow: a scale fp16 gradients cannot carry -> inf in the scaled gradients -> the step is skipped, the scale backs off
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    from simseg_amd.optim import GradScaler
    assert isinstance(tr.scaler, torch.amp.GradScaler) and not isinstance(tr.scaler, GradScaler)      # (the ten steps above: torch's own scaler, as the reference constructs it)
    tr.scaler = GradScaler("cuda", init_scale=2.0 ** 40, growth_interval=1000)      # the same scaler with this package's one-kernel overflow check
    step_no = tr.optimizer.steps_taken()
    assert step_no == 10
    tr.train_step(batch)
    assert tr.scaler.get_scale() == 2.0 ** 39
    assert tr.optimizer.steps_taken() == step_no                      # the kernel skipped the update on the device (no host read in scaler.step)
    assert all(torch.equal(before[n], p.detach()) for n, p in m.named_parameters())
    ck = tr.checkpoint()
    assert ck["scaler"]["scale"] == 2.0 ** 39
    m2 = _build(golden, ["epoch=1", "optim.lr.init=1e-3", "dist.fp16=True"])
    tr2 = Trainer(m2, m2.cfg, steps_per_epoch=40, amp_dtype="fp16")
    tr2.load_checkpoint(ck)
    assert tr2.scaler.get_scale() == 2.0 ** 39
    # no host synchronisation anywhere in the AMP iteration (forward, scaled backward, overflow check, unscale + update / skip, scale update);
    # the scale has come down to where steps are taken again, so this also covers a TAKEN step with either scaler
    for scaler in (GradScaler("cuda", init_scale=2.0 ** 14), torch.amp.GradScaler("cuda", init_scale=2.0 ** 14)):
        tr.scaler = scaler
        tr.train_step(batch)
        torch.cuda.synchronize()
        n0 = tr.optimizer.steps_taken()
        torch.cuda.set_sync_debug_mode("error")
        try:
            tr.train_step(batch)
        finally:
            torch.cuda.set_sync_debug_mode("default")
        assert tr.optimizer.steps_taken() == n0 + 1
