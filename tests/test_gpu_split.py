"""Exact-mode forward GEMMs on the bf16 matrix pipe (simseg_split_bf16x3 + one bf16 simseg_gemm over the six leading piece products):
the pieces are exact, the product is as close to the fp64 product as the fp32 MFMA kernel's, and the ViT tower evaluated through it
agrees with the fp32-kernel tower and with the CPU oracle (north star: <= 1e-3 fp32)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_split_pieces_are_exact_and_laid_out_along_k():
    from simseg_amd import ops
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.randn(300, 128, device="cuda", generator=g) * torch.logspace(-6, 4, 128, device="cuda")      # ten decades of magnitudes
    x[5, 7] = 0.0
    x[6, :8] = torch.tensor([1.0, -1.0, 3.0e-39, 65504.0, 1e30, -1e-30, 0.333333343, 16777217.0], device="cuda")     # incl. a subnormal
    for bpat, order in ((False, (0, 0, 0, 1, 1, 2)), (True, (0, 1, 2, 0, 1, 0))):
        s = ops.split_bf16x3(x, b_pattern=bpat)
        assert s.shape == (300, 768) and s.dtype == torch.bfloat16
        seg = s.view(300, 6, 128)
        hi = x.bfloat16()
        mid = (x - hi.float()).bfloat16()
        lo = (x - hi.float() - mid.float()).bfloat16()
        pieces = (hi, mid, lo)
        for j, k in enumerate(order):
            assert torch.equal(seg[:, j], pieces[k]), (bpat, j)
        total = hi.double() + mid.double() + lo.double()
        # three 8-bit pieces carry fp32's 24 significant bits: the sum is x itself (normal numbers)
        normal = x.abs() > 1e-36
        assert torch.equal(total[normal], x.double()[normal])


@pytest.mark.parametrize("M,N,K", [(16384, 768, 768), (8192, 2304, 768), (16384, 768, 3072)])
def test_split_gemm_is_fp32_accurate(M, N, K):
    from simseg_amd import ops
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(M, K, device="cuda", generator=g)
    b = torch.randn(N, K, device="cuda", generator=g) * 0.05
    bias = torch.randn(N, device="cuda", generator=g)
    want = a.double() @ b.double().T + bias.double()
    scale = (a.abs().double() @ b.abs().double().T)               # sum |a||b| per output: what rounding errors are relative to
    native = ops.gemm(a, b, bias=bias)
    split = ops.gemm(ops.split_bf16x3(a), ops.split_bf16x3(b, b_pattern=True), bias=bias, out_dtype=torch.float32)
    assert ops.raw("simseg_gemm_last_variant") == 3               # the 256x256 bf16 ping-pong kernel did it
    e_native = float(((native.double() - want).abs() / scale).max())
    e_split = float(((split.double() - want).abs() / scale).max())
    print(f"{M}x{N}x{K}: max |err| / sum|a||b|  fp32 MFMA kernel {e_native:.2e}   split-bf16 {e_split:.2e}")
    assert e_split < 1e-6, e_split
    assert e_split < 4 * e_native + 1e-8


def test_vit_tower_through_split_gemms_matches_fp32_kernels_and_oracle(monkeypatch):
    from oracle import simseg_ref as R
    from simseg_amd import towers
    from simseg_amd.nn import ViT
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    ref = R.init_weights_(R.RefViT("vit_small_patch16_224_in21k", 224), seed=4).eval()
    m = ViT("vit_small_patch16_224_in21k", 224)
    m.load_state_dict(ref.state_dict(), strict=False)
    m = m.cuda().eval()
    x = torch.randn(8, 3, 224, 224, generator=torch.Generator().manual_seed(1))        # 8 x 197 = 1576 token rows: 7 row tiles of 256
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    with torch.no_grad():
        want = ref(x)
        monkeypatch.setattr(towers, "_SPLIT_FP32", "0")
        native = m(x.cuda())
        monkeypatch.setattr(towers, "_SPLIT_FP32", "1")
        n0 = towers.SPLIT_CALLS[0]
        split = m(x.cuda())
        assert towers.SPLIT_CALLS[0] - n0 == 1 + 5 * 12          # patch embedding + four linear layers and the attention per block
    scale = float(want.abs().max())
    e_native = float((native.cpu() - want).abs().max()) / scale
    e_split = float((split.cpu() - want).abs().max()) / scale
    d = float((split - native).abs().max()) / scale
    print(f"ViT-S tower vs oracle: fp32 kernels {e_native:.2e}, split-bf16 {e_split:.2e} of the output scale; split vs fp32 kernels {d:.2e}")
    assert e_split * scale < 1e-3                                 # the north-star bound on fp32 outputs
    assert e_split < 2e-5 and d < 2e-5
    # a gradient-carrying forward keeps the fp32 kernels (their operands are what the backward re-reads)
    n0 = towers.SPLIT_CALLS[0]
    xg = x[:2].cuda()
    m(xg).sum().backward()
    assert towers.SPLIT_CALLS[0] == n0


def test_split_weight_copies_follow_the_optimizer(monkeypatch):
    """train -> exact-mode eval -> train -> eval: simseg_amd.optim.AdamW rewrites the fp32 masters through raw pointers (no `_version`
    bump), so the cached split-bf16 copies of the evaluation path must be dropped by the optimizer step; the second evaluation has to see
    the UPDATED weights (round-3 advisor finding: it used the first evaluation's copies)."""
    from simseg_amd import towers
    from simseg_amd.nn import ViT
    from simseg_amd.optim import AdamW
    monkeypatch.setenv("SIMSEG_AMD_COMPUTE", "fp32")
    torch.manual_seed(3)
    m = ViT("vit_test_patch16", 64).cuda()
    opt = AdamW(m.parameters(), lr=5e-2, weight_decay=0.0)
    x = torch.randn(64, 3, 64, 64, device="cuda")                 # 64 x 17 token rows

    def evals():
        m.eval()
        with torch.no_grad():
            monkeypatch.setattr(towers, "_SPLIT_FP32", "1")
            n0 = towers.SPLIT_CALLS[0]
            a = m(x)
            assert towers.SPLIT_CALLS[0] > n0                     # the split path ran
            monkeypatch.setattr(towers, "_SPLIT_FP32", "0")
            b = m(x)                                              # fp32 kernels on the live masters
        return a, b

    a0, b0 = evals()
    assert float((a0 - b0).abs().max()) < 2e-5 * float(b0.abs().max())
    m.train()
    monkeypatch.setattr(towers, "_SPLIT_FP32", "0")
    m(x).square().mean().backward()
    opt.step()
    a1, b1 = evals()
    assert float((b1 - b0).abs().max()) > 1e-2 * float(b0.abs().max())      # the step moved the weights
    assert float((a1 - b1).abs().max()) < 2e-5 * float(b1.abs().max()), "the split path evaluated stale weights"


@pytest.mark.parametrize("B,T,H", [(2, 1025, 12), (3, 325, 6), (2, 197, 4), (1, 64, 2), (1, 40, 1)])
def test_attention_through_bf16_pieces_is_fp32_accurate(B, T, H):
    """simseg_attention_fwd_x3 against the fp64 softmax(Q K^T / 8) V, next to the fp32 MFMA kernel on the same inputs."""
    from simseg_amd import ops
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.randn(B, T, 3 * H * 64, device="cuda", generator=g) * 1.5
    qkv[0, 0] *= 6.0                                          # a row with large scores: sharp softmax, a raised running maximum
    q, k, v = [t.view(B, T, H, 64).transpose(1, 2).double() for t in qkv.chunk(3, -1)]
    want = ((q @ k.transpose(-1, -2) * 0.125).softmax(-1) @ v).transpose(1, 2).reshape(B, T, H * 64)
    native, _ = ops.attention_fwd(qkv, H, None, scale=0.125)
    got = ops.attention_fwd_x3(qkv, H, scale=0.125)
    scale = float(want.abs().max())
    e_native = float((native.double() - want).abs().max()) / scale
    e_x3 = float((got.double() - want).abs().max()) / scale
    print(f"B={B} T={T} H={H}: max |err| / max|out|  fp32 MFMA kernel {e_native:.2e}   bf16 pieces {e_x3:.2e}")
    assert torch.isfinite(got).all()
    assert e_x3 < 2e-6 and e_x3 < 4 * e_native + 2e-7
