"""Kernel-level parity on MI355X: every C-ABI entry point against a plain torch fp32 restatement of the same op
on the same seeded inputs.  fp32 kernels: tolerance 1e-4 relative to the output scale (well inside the north-star
1e-3); bf16-operand kernels: compared with torch fp32 on the SAME bf16-rounded inputs, tolerance 2e-2 of scale
(bf16 output rounding) -- both written next to each check."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from simseg_amd import ops as o
    assert torch.cuda.is_available()
    return o


def _rand(*shape, seed=0, scale=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).cuda()


def _close(got, want, tol, what=""):
    got, want = got.float().cpu(), want.float().cpu()
    scale = want.abs().max().item() + 1e-12
    err = (got - want).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} (tol {tol})"


def test_tr16_probe_layout(ops):
    """ds_read_b64_tr_b16 with lane l addressing bytes [8l, 8l+8): every 16-lane group holds a [4][16] block and
    lane a of the group receives column a (4 consecutive rows)."""
    m = ops.tr16_probe().numpy()
    want = np.zeros((64, 4), dtype=np.int32)
    for l in range(64):
        for j in range(4):
            want[l, j] = (l >> 4) * 64 + j * 16 + (l & 15)
    assert np.array_equal(m, want), f"unexpected tr16 layout:\n{m[:16]}"


@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (256, 384, 96), (788, 1152, 384), (325, 21, 512), (4, 512, 128), (1000, 264, 40),
                                   (3, 512, 3), (130, 7, 45), (512, 3, 1)])
def test_gemm_f32_nt(ops, M, N, K):
    a, b = _rand(M, K, seed=1), _rand(N, K, seed=2)
    _close(ops.gemm(a, b), a @ b.T, 1e-5, f"f32 NT {M}x{N}x{K}")


def test_gemm_f32_epilogues(ops):
    M, N, K = 300, 200, 64
    a, b = _rand(M, K, seed=1), _rand(N, K, seed=2)
    bias, rs, res = _rand(N, seed=3), _rand(M, seed=4).abs() + 0.5, _rand(M, N, seed=5)
    ref = (a @ b.T) * 0.37 * rs[:, None] + bias
    _close(ops.gemm(a, b, alpha=0.37, bias=bias, rowscale=rs), ref, 1e-5, "alpha/rowscale/bias")
    pre = torch.empty(M, N, device="cuda")
    _close(ops.gemm(a, b, bias=bias, act=1, aux_out=pre, residual=res), F.gelu(a @ b.T + bias) + res, 1e-5, "gelu+res")
    _close(pre, a @ b.T + bias, 1e-5, "saved pre-activation")
    # training pair: act=3 stores GELU'(pre-activation), act=4 multiplies by the stored derivative
    x32 = (a @ b.T + bias).detach().requires_grad_(True)
    F.gelu(x32).backward(torch.ones_like(x32))
    dact = torch.empty(M, N, device="cuda")
    _close(ops.gemm(a, b, bias=bias, act=3, aux_out=dact), F.gelu(a @ b.T + bias), 1e-5, "gelu (derivative-saving)")
    _close(dact, x32.grad, 1e-5, "saved GELU'")
    _close(ops.gemm(a, b, act=4, aux=dact), (a @ b.T) * dact, 1e-5, "times saved derivative")
    # ViT token-row remap: rows of G patches land behind a [cls] row, residual = pos_embed[1 + r % G]
    G, Bn = 25, 12
    pos = _rand(G + 1, N, seed=6)
    out = torch.zeros(Bn * (G + 1), N, device="cuda")
    ops.gemm(a, b, bias=bias, residual=pos, row_group=G, res_mod=True, out=out)
    ref = torch.zeros(Bn, G + 1, N, device="cuda")
    ref[:, 1:] = (a @ b.T + bias).view(Bn, G, N) + pos[1:]
    _close(out.view(Bn, G + 1, N), ref, 1e-5, "row_group")
    acc = _rand(M, N, seed=7)
    want = acc + a @ b.T
    _close(ops.gemm(a, b, out=acc, accumulate=True), want, 1e-5, "accumulate")
    cs = torch.ones(N, device="cuda")
    y = ops.gemm(a, b, bias=bias, colsum=cs)
    _close(cs, 1 + y.sum(0), 1e-5, "epilogue column sums")
    cs = torch.zeros(N, device="cuda")
    y16 = ops.gemm(a.bfloat16(), b.bfloat16(), bias=bias, colsum=cs)
    _close(cs, (a.bfloat16().float() @ b.bfloat16().float().T + bias).sum(0), 1e-4, "epilogue column sums (bf16, pre-rounding values)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(325, 384, 384), (325, 1152, 384), (1025, 768, 768), (1025, 768, 3072), (1025, 3072, 768), (325, 21, 512),
                                   (1, 512, 768), (7, 70, 64), (64, 64, 128), (33, 65, 1024)])
def test_gemm_small_problem_kernel(ops, dtype, M, N, K):
    """The batch-1 shapes of the reference's segmentation tool (one 288^2 / 512^2 image: 325 / 1025 token rows): 32x64 / 64x64 tiles with the
    K slab split between wave groups.  Auto-dispatch takes it for these shapes; results and every fused epilogue against fp64 torch,
    and the dropout mask is the 128x128 kernel's (same element hash)."""
    from simseg_amd.lib import raw
    a, b = _rand(M, K, seed=1, dtype=dtype), _rand(N, K, seed=2, scale=K ** -0.5, dtype=dtype)      # products of O(1)
    ref = (a.double() @ b.double().T).float()
    tol = 1e-5
    ops.gemm(a, b, out_dtype=torch.float32)
    auto = raw("simseg_gemm_last_variant")
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    assert auto == (4 if t128 < (200 if dtype == torch.float32 else 160) else (5 if (dtype == torch.bfloat16 and t128 <= 256) else 1))
    ops.set_gemm_variant(4)
    try:
        out = ops.gemm(a, b, out_dtype=torch.float32)
        assert raw("simseg_gemm_last_variant") == 4
        _close(out, ref, tol, f"small {M}x{N}x{K}")
        bias, res, rs = _rand(N, seed=3), _rand(M, N, seed=4), _rand(M, seed=5).abs() + 0.5
        dact = torch.empty(M, N, device="cuda", dtype=dtype)
        got = ops.gemm(a, b, bias=bias, act=3, aux_out=dact)
        x = (ref + bias).requires_grad_(True)
        F.gelu(x).backward(torch.ones_like(x))
        _close(got, F.gelu(ref + bias), tol if dtype == torch.float32 else 1e-2, "gelu")
        _close(dact, x.grad, tol if dtype == torch.float32 else 1e-2, "saved GELU'")
        cs = torch.zeros(N, device="cuda")
        y = ops.gemm(a, b, alpha=0.5, rowscale=rs, bias=bias, residual=res, out_dtype=torch.float32, colsum=cs)
        _close(y, ref * 0.5 * rs[:, None] + bias + res, tol, "alpha/rowscale/bias/residual")
        _close(cs, y.sum(0), 1e-4, "column sums")
        acc = _rand(M, N, seed=7)
        want = acc + ref
        _close(ops.gemm(a, b, out=acc, accumulate=True), want, tol, "accumulate")
        d4 = ops.gemm(a, b, bias=bias, out_dtype=torch.float32, drop_seed=99, drop_p=0.25)
        ops.set_gemm_variant(1)
        d1 = ops.gemm(a, b, bias=bias, out_dtype=torch.float32, drop_seed=99, drop_p=0.25)
        assert raw("simseg_gemm_last_variant") == 1
        o1 = ops.gemm(a, b, out_dtype=torch.float32)
    finally:
        ops.set_gemm_variant(0)
    assert float(((d4 == 0) != (d1 == 0)).float().mean()) < 1e-4      # the same mask (an exactly-zero sum aside)
    _close(d4, d1, tol, "dropout epilogue, both kernels")
    _close(out, o1, tol, "both kernels")


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (384, 256, 200), (768, 776, 1000), (72, 40, 24)])
def test_gemm_bf16_layouts(ops, ta, tb, M, N, K):
    if not ta:
        K = (K + 7) // 8 * 8
    a = _rand(*((K, M) if ta else (M, K)), seed=1, dtype=torch.bfloat16)
    b = _rand(*((K, N) if tb else (N, K)), seed=2, dtype=torch.bfloat16)
    A = a.float().T if ta else a.float()
    Bm = b.float() if tb else b.float().T
    ref = A @ Bm
    _close(ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32), ref, 1e-5, f"bf16 ta={ta} tb={tb} f32 out")
    _close(ops.gemm(a, b, trans_a=ta, trans_b=tb), ref, 1e-2, f"bf16 ta={ta} tb={tb} bf16 out")


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
@pytest.mark.parametrize("M,N,K", [(512, 256, 128), (1000, 520, 192), (256, 136, 64), (2048, 768, 768), (16640, 768, 192)])
def test_gemm_bf16_large_tile_kernel(ops, variant, ta, tb, M, N, K):
    """The 256x256 direct-to-LDS kernels (swizzled LDS image, counted vmcnt ring) on interior and edge tiles."""
    ops.set_gemm_variant(variant)
    try:
        a = _rand(*((K, M) if ta else (M, K)), seed=1, dtype=torch.bfloat16)
        b = _rand(*((K, N) if tb else (N, K)), seed=2, dtype=torch.bfloat16)
        ref = (a.float().T if ta else a.float()) @ (b.float() if tb else b.float().T)
        _close(ops.gemm(a, b, trans_a=ta, trans_b=tb, out_dtype=torch.float32), ref, 1e-5, f"large v{variant} ta={ta} tb={tb}")
        bias, res = _rand(N, seed=3), _rand(M, N, seed=4)
        _close(ops.gemm(a, b, trans_a=ta, trans_b=tb, bias=bias, residual=res, out_dtype=torch.float32), ref + bias + res, 1e-5, "large + epilogue")
        acc = torch.zeros(M, N, device="cuda")
        ops.gemm(a, b, trans_a=ta, trans_b=tb, out=acc, accumulate=True, splitk=3)
        _close(acc, ref, 2e-5, "large split-K")
    finally:
        ops.set_gemm_variant(0)


@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("M,N,K", [(512, 512, 768), (1024, 768, 128)])
def test_gemm_pingpong_accumulator_layout_epilogues(ops, tb, M, N, K):
    """Full 256x256 tiles on the ping-pong kernel finish in the accumulator layout (bf16: transposed staging + ds_read_b64_tr_b16;
    fp32 + residual: direct row-segment stores).  Every such configuration against fp64 torch AND against the staged epilogue of the
    128x128 kernel (same hash -> same dropout mask)."""
    from simseg_amd.lib import raw
    a = _rand(M, K, seed=1, dtype=torch.bfloat16)
    b = _rand(*((K, N) if tb else (N, K)), seed=2, scale=K ** -0.5, dtype=torch.bfloat16)
    ref = (a.double() @ (b.double() if tb else b.double().T)).float()
    bias, res = _rand(N, seed=3), _rand(M, N, seed=4)
    aux = _rand(M, N, seed=5, dtype=torch.bfloat16)
    x = (ref + bias).requires_grad_(True)
    F.gelu(x).backward(torch.ones_like(x))
    results = {}
    for variant in (3, 1):
        ops.set_gemm_variant(variant)
        try:
            r = {}
            r["plain"] = ops.gemm(a, b, trans_b=tb)
            assert raw("simseg_gemm_last_variant") == variant
            r["bias_alpha"] = ops.gemm(a, b, trans_b=tb, bias=bias, alpha=0.5)
            pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            r["gelu"] = ops.gemm(a, b, trans_b=tb, bias=bias, act=1, aux_out=pre)
            r["gelu_pre"] = pre
            d = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            r["gelu3"] = ops.gemm(a, b, trans_b=tb, bias=bias, act=3, aux_out=d)
            r["gelu3_d"] = d
            r["gelu3_noaux"] = ops.gemm(a, b, trans_b=tb, bias=bias, act=3)
            cs = torch.zeros(N, device="cuda")
            r["times_aux"] = ops.gemm(a, b, trans_b=tb, act=4, aux=aux, colsum=cs)
            r["times_aux_colsum"] = cs
            r["f32_res"] = ops.gemm(a, b, trans_b=tb, bias=bias, residual=res, out_dtype=torch.float32)
            r["f32_res_drop"] = ops.gemm(a, b, trans_b=tb, bias=bias, residual=res, out_dtype=torch.float32, drop_seed=7, drop_p=0.1)
            results[variant] = r
        finally:
            ops.set_gemm_variant(0)
    want = {"plain": ref, "bias_alpha": ref * 0.5 + bias, "gelu": F.gelu(ref + bias), "gelu_pre": ref + bias, "gelu3": F.gelu(ref + bias),
            "gelu3_d": x.grad, "gelu3_noaux": F.gelu(ref + bias), "times_aux": ref * aux.float(), "f32_res": ref + bias + res}
    for k, w in want.items():
        _close(results[3][k], w, 1e-5 if k.startswith("f32") else 1e-2, f"ping-pong {k}")
    _close(results[3]["times_aux_colsum"], (ref * aux.float()).sum(0), 4e-3, "column sums of the stored products")       # (the ping-pong epilogue rounds the matmul result to bf16 before the multiplication, as autocast does: 2^-9 per term)
    for k in results[3]:
        tol = 1e-5 if k.startswith("f32") else (4e-3 if k.endswith("colsum") else 1e-2)
        _close(results[3][k], results[1][k].float(), tol, f"ping-pong vs staged epilogue: {k}")
    d3, d1 = results[3]["f32_res_drop"] - res, results[1]["f32_res_drop"] - res
    assert float(((d3.abs() < 1e-6) != (d1.abs() < 1e-6)).float().mean()) < 1e-3          # the same dropout mask


@pytest.mark.parametrize("tb", [False, True])
@pytest.mark.parametrize("M,N,K", [(256 * 90, 768, 768), (256 * 33, 2304, 192), (256 * 131, 768, 320), (256 * 9, 256 * 30, 128)])
def test_gemm_persistent_pingpong_kernel(ops, tb, M, N, K):
    """The persistent 256x256 ping-pong kernel (one workgroup per CU walks its XCD's tile list, operand copies running across tile
    boundaries, epilogue scratch in the ring slots of the K-tile consumed last): every epilogue it carries, on tile counts that leave
    blocks with 1, 2, ... tiles and an uneven split over the XCDs, odd and even K-tile counts (the ring parity at a tile boundary),
    against fp64 torch and - bit for bit, same MFMA order - against the per-tile kernel."""
    from simseg_amd.lib import raw
    a = _rand(M, K, seed=1, dtype=torch.bfloat16)
    b = _rand(*((K, N) if tb else (N, K)), seed=2, scale=K ** -0.5, dtype=torch.bfloat16)
    ref = (a.double() @ (b.double() if tb else b.double().T)).float()
    bias, res = _rand(N, seed=3), _rand(M, N, seed=4)
    aux = _rand(M, N, seed=5, dtype=torch.bfloat16)
    x = (ref + bias).requires_grad_(True)
    F.gelu(x).backward(torch.ones_like(x))
    results = {}
    for variant in (10, 3):
        ops.set_gemm_variant(variant)
        try:
            r = {}
            r["plain"] = ops.gemm(a, b, trans_b=tb)
            assert raw("simseg_gemm_last_variant") == variant, (variant, raw("simseg_gemm_last_variant"))
            if variant == 10:
                for sched in (11, 12, 13):          # the schedule experiments compute the same thing
                    ops.set_gemm_variant(sched)
                    assert torch.equal(ops.gemm(a, b, trans_b=tb), r["plain"]), sched
                ops.set_gemm_variant(variant)
            r["bias_alpha"] = ops.gemm(a, b, trans_b=tb, bias=bias, alpha=0.5)
            pre = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            r["gelu"] = ops.gemm(a, b, trans_b=tb, bias=bias, act=1, aux_out=pre)
            r["gelu_pre"] = pre
            d = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
            r["gelu3"] = ops.gemm(a, b, trans_b=tb, bias=bias, act=3, aux_out=d)
            r["gelu3_d"] = d
            r["gelu3_noaux"] = ops.gemm(a, b, trans_b=tb, bias=bias, act=3)
            cs = torch.zeros(N, device="cuda")
            r["times_aux"] = ops.gemm(a, b, trans_b=tb, act=4, aux=aux, colsum=cs)
            r["times_aux_colsum"] = cs
            r["f32_res"] = ops.gemm(a, b, trans_b=tb, bias=bias, residual=res, out_dtype=torch.float32)
            assert raw("simseg_gemm_last_variant") == variant
            r["f32_res_drop"] = ops.gemm(a, b, trans_b=tb, bias=bias, residual=res, out_dtype=torch.float32, drop_seed=7, drop_p=0.1)
            results[variant] = r
        finally:
            ops.set_gemm_variant(0)
    want = {"plain": ref, "bias_alpha": ref * 0.5 + bias, "gelu": F.gelu(ref + bias), "gelu_pre": ref + bias, "gelu3": F.gelu(ref + bias),
            "gelu3_d": x.grad, "gelu3_noaux": F.gelu(ref + bias), "times_aux": ref * aux.float(), "f32_res": ref + bias + res}
    for k, w in want.items():
        _close(results[10][k], w, 1e-5 if k.startswith("f32") else 1e-2, f"persistent {k}")
    _close(results[10]["times_aux_colsum"], (ref * aux.float()).sum(0), 4e-3, "column sums")
    for k in results[10]:
        if k.endswith("colsum"):
            continue                                                          # (atomic order differs)
        assert torch.equal(results[10][k], results[3][k]), f"persistent vs per-tile kernel: {k} differs"
    # the automatic choice takes the persistent kernel from 600 full tiles on (test_gemm_auto_takes_the_persistent_kernel_for_many_tiles); these are smaller
    ops.gemm(a, b, trans_b=tb)
    assert raw("simseg_gemm_last_variant") == (3 if K >= 768 else 1)       # (short contractions stay on the 128x128 kernel)
    # repeated launches are independent (no state carried in the ring / scratch between launches)
    again = ops.gemm(a, b, trans_b=tb, bias=bias, alpha=0.5)
    assert torch.equal(again, results[10]["bias_alpha"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gather_rows(ops, dtype):
    """Row gather with -1 -> zero row (packing / unpacking the real tokens of a ragged caption batch)."""
    src = _rand(1000, 768, seed=1, dtype=dtype)
    idx = torch.randint(-1, 1000, (2500,), generator=torch.Generator().manual_seed(2)).to(torch.int32).cuda()
    out = ops.gather_rows(src, idx)
    want = torch.where((idx >= 0)[:, None], src[idx.clamp(min=0).long()], torch.zeros((), device="cuda", dtype=dtype))
    assert torch.equal(out, want)
    with pytest.raises(TypeError):
        ops.gather_rows(src, idx.long())


def test_gemm_bf16_splitk_and_dgelu(ops):
    K, M, N = 5000, 256, 384          # wgrad shape: contraction over rows
    dy, x = _rand(K, M, seed=1, dtype=torch.bfloat16), _rand(K, N, seed=2, dtype=torch.bfloat16)
    acc = torch.zeros(M, N, device="cuda")
    ops.gemm(dy, x, trans_a=True, trans_b=True, out=acc, accumulate=True, splitk=16)
    _close(acc, dy.float().T @ x.float(), 2e-5, "split-K wgrad")
    # dgrad with the GELU' epilogue
    a, w = _rand(300, 256, seed=3, dtype=torch.bfloat16), _rand(256, 512, seed=4, dtype=torch.bfloat16)
    pre = _rand(300, 512, seed=5, dtype=torch.bfloat16)
    x32 = pre.float().requires_grad_(True)
    F.gelu(x32).backward(torch.ones_like(x32))
    # aux shares the output dtype (bf16 activations on the training path)
    _close(ops.gemm(a, w, trans_b=True, act=2, aux=pre), (a.float() @ w.float()) * x32.grad, 1e-2, "dgelu")


def test_gemm_dropout_epilogue(ops):
    a, b = _rand(512, 64, seed=1), _rand(256, 64, seed=2)
    y0 = ops.gemm(a, b)
    y = ops.gemm(a, b, drop_seed=1234, drop_p=0.1)
    kept = y != 0
    frac = 1 - kept.float().mean().item()
    assert abs(frac - 0.1) < 0.01, frac
    _close(y[kept], y0[kept] / 0.9, 1e-5, "kept values scaled")
    g = torch.ones_like(y0)
    ops.dropout_apply_(g, 1234, 0.1)
    assert torch.equal(g != 0, kept)          # backward regenerates the same mask
    assert not torch.equal(ops.gemm(a, b, drop_seed=99, drop_p=0.1) != 0, kept)


@pytest.mark.parametrize("D", [128, 384, 768, 1024])
def test_layernorm_fwd_bwd(ops, D):
    rows = 333
    x = _rand(rows, D, seed=1, scale=3.0) + 0.5
    g, b = _rand(D, seed=2) * 0.2 + 1.0, _rand(D, seed=3) * 0.1
    for eps in (1e-6, 1e-12):
        y, y16, mean, rstd = ops.layernorm_fwd(x, g, b, eps, want_bf16_copy=True, save_stats=True)
        ref = F.layer_norm(x, (D,), g, b, eps)
        _close(y, ref, 1e-5, "ln fwd")
        _close(y16, ref, 1e-2, "ln fwd bf16 copy")
    yb, *_ = ops.layernorm_fwd(x, g, b, 1e-6, out_dtype=torch.bfloat16)
    _close(yb, ref, 1e-2, "ln fwd bf16 out")
    # backward:  dy = dy16 + dy32,  dx = LN'(dy) + dres
    dy32, dy16, dres = _rand(rows, D, seed=4), _rand(rows, D, seed=5, dtype=torch.bfloat16), _rand(rows, D, seed=6)
    xr = x.clone().requires_grad_(True); gr = g.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    F.layer_norm(xr, (D,), gr, br, 1e-12).backward(dy32 + dy16.float())
    dgam = torch.zeros(D, device="cuda"); dbet = torch.zeros(D, device="cuda")
    dx32, dx16 = ops.layernorm_bwd(x, mean, rstd, g, dgam, dbet, dy16=dy16, dy32=dy32, dres=dres)
    _close(dx32, xr.grad + dres, 2e-5, "ln bwd dx")
    _close(dx16, xr.grad + dres, 1e-2, "ln bwd dx bf16")
    _close(dgam, gr.grad, 1e-4, "ln bwd dgamma")
    _close(dbet, br.grad, 1e-4, "ln bwd dbeta")
    # fused extras: column sums of the bf16 output and re-application of a dense layer's dropout mask to it
    dgam.zero_(); dbet.zero_()
    dsum = torch.zeros(D, device="cuda")
    dx32b, dx16b = ops.layernorm_bwd(x, mean, rstd, g, dgam, dbet, dy16=dy16, dy32=dy32, dres=dres, dxsum=dsum, drop_seed=77, drop_p=0.1)
    assert torch.equal(dx32b, dx32)                      # the fp32 stream is never masked
    keep = torch.ones(rows, D, device="cuda")
    ops.dropout_apply_(keep, 77, 0.1)
    _close(dx16b, dx32 * keep, 1e-2, "ln bwd masked bf16 output")
    _close(dsum, (dx32 * keep).sum(0), 1e-4, "ln bwd column sums")


@pytest.mark.parametrize("D", [384, 768])
def test_layernorm_bwd_16bit_residual_and_xhat_from_the_saved_output(ops, D):
    """Round 4 inputs of simseg_layernorm_bwd: the residual gradient as a 16-bit tensor (dres_bf16, no fp32 image out) and the normalised value
    taken from the layer's saved 16-bit OUTPUT (y_bf16 + beta) instead of the fp32 input - for the chunks whose gains allow it; chunks with a
    small gain or a large offset read x as before.  Against the fp64 derivative."""
    rows = 517
    x = _rand(rows, D, seed=1, scale=3.0) + 0.5
    g, b = torch.exp(_rand(D, seed=2) * 0.7), _rand(D, seed=3) * 0.3
    g[5] = 0.01; g[40] = -0.02; b[77] = 9.0; g[77] = 1.0          # chunks that must fall back to x (tiny gain, offset >> gain)
    y, y16, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, want_bf16_copy=True, save_stats=True)
    dy16, dres16 = _rand(rows, D, seed=5, dtype=torch.bfloat16), _rand(rows, D, seed=6, dtype=torch.bfloat16)
    xr = x.double().requires_grad_(True); gr = g.double().requires_grad_(True); br = b.double().requires_grad_(True)
    F.layer_norm(xr, (D,), gr, br, 1e-6).backward(dy16.double())
    want = xr.grad + dres16.double()
    res = {}
    for tag, kw in (("x", {}), ("y", dict(y16=y16, beta=b))):
        dgam = torch.zeros(D, device="cuda"); dbet = torch.zeros(D, device="cuda")
        dx32, dx16 = ops.layernorm_bwd(x, mean, rstd, g, dgam, dbet, dy16=dy16, dres16=dres16, want_f32=False, **kw)
        assert dx32 is None
        res[tag] = (dx16.double(), dgam.double(), dbet.double())
    rel = lambda a, w: float((a - w).norm() / w.norm())       # noqa: E731
    ex, ey = rel(res["x"][0], want), rel(res["y"][0], want)
    assert ex < 3e-3 and ey < 1.3 * ex + 2e-4, (ex, ey)        # the 16-bit rounding of dx dominates both
    _close(res["y"][1], gr.grad, 5e-3, "dgamma with xhat from y")
    _close(res["y"][2], br.grad, 1e-4, "dbeta")
    # the fall-back chunks read x like the x path (their rows' two reductions still see the other chunks' xhat from y: close, not bit-equal)
    for c in (5, 40, 77):
        lo = c // 4 * 4
        assert float((res["x"][0][:, lo:lo + 4] - res["y"][0][:, lo:lo + 4]).abs().max()) <= 2e-2 * float(want.abs().max())


@pytest.mark.parametrize("D", [256, 384, 768, 1024, 1280])
@pytest.mark.parametrize("mixed", [False, True])
def test_layernorm_bwd_16bit_step_form_many_rows(ops, D, mixed):
    """Round 5's kernel of the 16-bit step form (ln_bwd16_kernel: dy16 + y16 + dres16 -> dx16): more rows than one resident round of waves (the
    software-pipelined row loop runs 2-3 times per wave and ends on its self-re-read), rows exactly MAXC * 256 wide and ragged ones, every
    gain eligible for the y16 read and a mix that needs x, with the fused column sums and the dropout mask.  Against the fp64 derivative and
    against the generic kernel (x path) on the same inputs."""
    rows = 9001
    x = _rand(rows, D, seed=1, scale=2.0) + 0.3
    g, b = torch.exp(_rand(D, seed=2) * 0.5), _rand(D, seed=3) * 0.2
    if mixed:
        g[3] = 0.01; g[D - 2] = -0.03; b[D // 2] = 30.0
    y, y16, mean, rstd = ops.layernorm_fwd(x, g, b, 1e-6, want_bf16_copy=True, save_stats=True)
    dy16, dres16 = _rand(rows, D, seed=5, dtype=torch.bfloat16), _rand(rows, D, seed=6, dtype=torch.bfloat16)
    xr = x.double().requires_grad_(True); gr = g.double().requires_grad_(True); br = b.double().requires_grad_(True)
    F.layer_norm(xr, (D,), gr, br, 1e-6).backward(dy16.double())
    want = xr.grad + dres16.double()
    out = {}
    for tag, kw in (("x", {}), ("y", dict(y16=y16, beta=b))):
        dgam = torch.zeros(D, device="cuda"); dbet = torch.zeros(D, device="cuda"); dsum = torch.zeros(D, device="cuda")
        _, dx16 = ops.layernorm_bwd(x, mean, rstd, g, dgam, dbet, dy16=dy16, dres16=dres16, want_f32=False, dxsum=dsum, **kw)
        out[tag] = (dx16.double(), dgam.double(), dbet.double(), dsum.double())
    rel = lambda a, w: float((a - w).norm() / w.norm())       # noqa: E731
    ex, ey = rel(out["x"][0], want), rel(out["y"][0], want)
    assert ex < 3e-3 and ey < 1.3 * ex + 2e-4, (ex, ey)
    ok = torch.ones(D, dtype=torch.bool, device="cuda")
    if mixed:                                                  # the two tiny gains' dgamma is only as good as xhat from x is: compared on the x path below
        ok[3] = ok[D - 2] = False
    assert rel(out["y"][1][ok], gr.grad[ok]) < 5e-3
    assert rel(out["y"][1], out["x"][1]) < 5e-3                # ... and the fall-back chunks match the generic kernel's
    assert rel(out["y"][2], br.grad) < 1e-5
    assert rel(out["y"][3], want.sum(0)) < 2e-2 and rel(out["y"][3], out["y"][0].sum(0)) < 1e-2   # column sums of what was written (taken before its 16-bit rounding)
    # dropout mask of the producing dense layer re-applied to the written copy, column sums of the masked values
    dgam = torch.zeros(D, device="cuda"); dbet = torch.zeros(D, device="cuda"); dsum = torch.zeros(D, device="cuda")
    _, dxm = ops.layernorm_bwd(x, mean, rstd, g, dgam, dbet, dy16=dy16, dres16=dres16, want_f32=False, dxsum=dsum, drop_seed=77, drop_p=0.1, y16=y16, beta=b)
    keep = torch.ones(rows, D, device="cuda")
    ops.dropout_apply_(keep, 77, 0.1)
    assert rel(dxm.double(), out["y"][0] * keep.double()) < 5e-3
    assert torch.equal(dxm == 0, (keep == 0) | (dxm == 0))
    assert rel(dsum.double(), dxm.double().sum(0)) < 1e-2


def test_colsum_transpose_cast(ops):
    x = _rand(1001, 776, seed=1)
    out = torch.ones(776, device="cuda")
    ops.colsum_accum(x, out)
    _close(out, 1 + x.sum(0), 1e-5, "colsum f32")
    xb = x.bfloat16()
    out = torch.zeros(776, device="cuda")
    ops.colsum_accum(xb, out)
    _close(out, xb.float().sum(0), 1e-5, "colsum bf16")
    assert torch.equal(ops.transpose_f32(x), x.T.contiguous())
    assert torch.equal(ops.cast(x, torch.bfloat16), x.bfloat16())
    assert torch.equal(ops.cast(xb, torch.float32), xb.float())


def test_vit_patch_embed(ops):
    """im2col + GEMM(row_group) + cls rows == Conv2d patch embed, cls concat, +pos (vit_builder.py:14-17)."""
    B, D, S = 3, 128, 96
    N = (S // 16) ** 2
    img = _rand(B, 3, S, S, seed=1)
    w, bias = _rand(D, 3, 16, 16, seed=2, scale=0.05), _rand(D, seed=3)
    cls, pos = _rand(1, 1, D, seed=4), _rand(1, 1 + N, D, seed=5)
    ref = F.conv2d(img, w, bias, stride=16).flatten(2).transpose(1, 2)
    ref = torch.cat([cls.expand(B, -1, -1), ref], 1) + pos
    x = torch.empty(B, 1 + N, D, device="cuda")
    cols = ops.vit_im2col(img, torch.float32)
    ops.gemm(cols, w.view(D, 768), bias=bias, residual=pos.view(1 + N, D), row_group=N, res_mod=True, out=x.view(-1, D))
    ops.vit_cls_rows(cls.view(-1), pos.view(-1), x)
    _close(x, ref, 1e-5, "patch embed")


def test_bert_embed(ops):
    B, L, D, V = 5, 25, 128, 1000
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(0, V, (B, L), generator=g).cuda()
    mask = (torch.rand(B, L, generator=g) < 0.7).long().cuda()
    word, pos, typ = _rand(V, D, seed=1), _rand(128, D, seed=2), _rand(2, D, seed=3)
    out = ops.bert_embed_fwd(ids, word, pos, typ[0].contiguous())
    ref = word[ids] + typ[0] + pos[:L][None]
    _close(out, ref, 1e-6, "bert embed")
    dsum = _rand(B, L, D, seed=4)
    dword = torch.zeros(V, D, device="cuda")
    ops.bert_embed_bwd(ids, mask, dsum, dword)
    refw = torch.zeros(V, D, device="cuda").index_add_(0, ids.view(-1), (dsum * mask[..., None]).view(-1, D))
    _close(dword, refw, 1e-5, "bert embed bwd")


def test_gemm_randomised_shapes_and_epilogues(ops):
    """40 seeded random problems through whatever kernel the dispatcher picks (128x128 register-staged, 256x256 direct-to-LDS,
    split-K): ragged M / N, every operand orientation, bf16 and fp32, with bias / GELU / fp32 residual / column sums."""
    import random
    rng = random.Random(1234)
    for case in range(40):
        bf = rng.random() < 0.7
        ta = bf and rng.random() < 0.25
        tb = bf and (ta or rng.random() < 0.5)  # transposed operands are a bf16 (training) feature; (ta, !tb) is not on the path
        M = rng.choice([1, 7, 64, 200, 257, 777, 1025, 2304, 4100])
        N = rng.choice([8, 24, 128, 264, 768, 1000, 1536]) if bf else rng.choice([3, 21, 128, 171, 512])
        K = rng.choice([64, 128, 320, 768, 1536]) if bf else rng.choice([1, 40, 96, 512])
        if bf and (ta or tb):
            N = (N + 7) // 8 * 8
        if ta:
            M = (M + 7) // 8 * 8
        dt_ = torch.bfloat16 if bf else torch.float32
        a = _rand(*((K, M) if ta else (M, K)), seed=case, dtype=dt_)
        b = _rand(*((K, N) if tb else (N, K)), seed=case + 100, dtype=dt_)
        A = a.float().t() if ta else a.float()
        Bm = b.float() if tb else b.float().t()
        ref = A @ Bm
        kw, what = {}, f"case {case}: M={M} N={N} K={K} ta={ta} tb={tb} {'bf16' if bf else 'f32'}"
        mode = rng.choice(["plain", "bias", "gelu", "residual", "colsum"])
        if mode in ("bias", "gelu", "residual"):
            bias = _rand(N, seed=case + 200)
            kw["bias"] = bias
            ref = ref + bias.float().cpu().cuda()
        if mode == "gelu":
            kw["act"] = 1
            ref = F.gelu(ref)
        if mode == "residual":
            res = _rand(M, N, seed=case + 300)
            kw.update(residual=res, out_dtype=torch.float32)
            ref = ref + res
        cs = None
        if mode == "colsum":
            cs = torch.zeros(N, device="cuda")
            kw["colsum"] = cs
        out = ops.gemm(a, b, trans_a=ta, trans_b=tb, **kw)
        tol = 2e-2 if bf and out.dtype == torch.bfloat16 else (2e-5 if not bf else 1e-4)
        _close(out, ref, tol, what + " " + mode)
        if cs is not None:
            _close(cs, ref.sum(0), 1e-4, what + " column sums (of the fp32 values, before the output is rounded)")


def test_topk_pool_sliced_equals_single_pass(ops):
    """Small batches scan token slices in parallel and merge; the result - selected indices included, with the many ties of
    bf16 data - is the single-pass kernel's."""
    from simseg_amd.lib import call, ptr, raw, stream
    for dtype, (B, N, P, k) in ((torch.bfloat16, (3, 1024, 512, 5)), (torch.float32, (1, 333, 128, 3)), (torch.bfloat16, (2, 300, 64, 1))):
        tok = (_rand(B, N, P, seed=N, dtype=torch.float32) * 2).round().div(2).to(dtype)        # coarse values: ties everywhere
        mask = None
        outs = []
        for sliced in (False, True):
            emb = torch.empty(B, P, device="cuda"); idx = torch.empty(B, k, P, device="cuda", dtype=torch.int32); norm = torch.empty(B, device="cuda")
            scratch = torch.empty(raw("simseg_topk_pool_workspace_bytes", B, P, k) // 4, device="cuda") if sliced else None
            call("simseg_topk_pool_l2norm_fwd", ptr(tok), 0 if dtype == torch.float32 else 1, None, ptr(emb), ptr(idx), ptr(norm), ptr(scratch),
                 B, N, P, k, 1e-8, 1, stream())
            outs.append((emb, idx, norm))
        assert torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_topk_pool_l2norm(ops, golden, dtype):
    from conftest import tt
    g = golden("heads")
    tok = tt(g["tok"]).cuda().to(dtype)
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    emb, idx, norm = ops.topk_pool_l2norm_fwd(tok, 5)
    x = tok.float().requires_grad_(True)
    pooled = x.topk(5, dim=1)[0].mean(1)
    ref = pooled / (pooled.pow(2).sum(-1, keepdim=True).sqrt() + 1e-8)
    _close(emb, ref, 1e-5, "loda image pool")
    if dtype == torch.float32:
        _close(emb, tt(g["emb"]), 1e-5, "loda image pool vs reference fixture")
    gy = tt(g["gy"]).cuda()
    ref.backward(gy)
    dtok = ops.topk_pool_l2norm_bwd(gy, emb, norm, idx, tok.shape[1], dtype)
    if dtype == torch.float32:
        _close(dtok, x.grad, tol, "loda pool bwd")
    else:
        # bf16 token values tie; torch.topk's tie-break is unspecified (SURVEY 7 "hard parts"), so compare the
        # tie-invariant quantities: gradient mass per channel and the values of the selected tokens
        _close(dtok.float().sum(1), x.grad.sum(1), tol, "loda pool bwd (mass per channel)")
        sel = (dtok != 0)
        assert int(sel.sum(1).max()) <= 5
        _close((tok.float() * sel).sum(1), tok.float().topk(5, dim=1)[0].sum(1), 1e-5, "selected tokens are the top-5")
    # masked text pooling k=1 (pipelines/clip.py:111-120)
    t, mask = tt(g["t"]).cuda().to(dtype), tt(g["mask"]).cuda()
    temb, tidx, tnorm = ops.topk_pool_l2norm_fwd(t, 1, mask)
    if dtype == torch.float32:
        _close(temb, tt(g["temb"]), 1e-5, "masked text pool vs reference fixture")
        dt_ = ops.topk_pool_l2norm_bwd(tt(g["gt"]).cuda(), temb, tnorm, tidx, t.shape[1], dtype)
        _close(dt_, tt(g["gt_in"]), 1e-5, "masked text pool bwd vs reference fixture")
    _close(ops.topk_pool_l2norm_fwd(t, 2, tt(g["mask2"]).cuda())[0],
           (lambda p: p / (p.norm(dim=-1, keepdim=True) + 1e-8))(
               torch.where(tt(g["mask2"]).cuda()[..., None] == 0, torch.full_like(t.float(), -10000.), t.float()).topk(2, dim=1)[0].mean(1)),
           1e-5, "masked pool k=2")


def test_row_rnorm(ops):
    x = _rand(777, 512, seed=1, scale=4.0)
    _close(ops.row_rnorm(x), 1.0 / x.norm(dim=-1).clamp_min(1e-12), 1e-5, "rnorm")
    _close(ops.row_rnorm(x.bfloat16()), 1.0 / x.bfloat16().float().norm(dim=-1).clamp_min(1e-12), 1e-5, "rnorm bf16")


def _attn_ref(qkv, H, mask, scale):
    B, T, _ = qkv.shape
    q, k, v = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) * scale
    if mask is not None:
        s = s + (1.0 - mask[:, None, None, :].float()) * -10000.0
    return (s.softmax(-1) @ v).transpose(1, 2).reshape(B, T, H * 64)


@pytest.mark.parametrize("T", [25, 77, 197, 325])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_fwd(ops, T, dtype):
    B, H = 3, 2
    qkv = _rand(B, T, 3 * H * 64, seed=T, scale=1.5, dtype=dtype)
    mask = None
    if T <= 77:       # text shapes: ragged key-padding mask
        mask = torch.zeros(B, T, dtype=torch.long)
        for b, n in enumerate((T, 7, 3)):
            mask[b, :n] = 1
        mask = mask.cuda()
    out, lse = ops.attention_fwd(qkv, H, mask, save_lse=True)
    ref = _attn_ref(qkv, H, mask, 0.125)
    _close(out, ref, 1e-5 if dtype == torch.float32 else 1.5e-2, f"attention fwd T={T}")
    q, k, _ = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = q @ k.transpose(-1, -2) * 0.125
    if mask is not None:
        s = s.masked_fill(mask[:, None, None, :] == 0, -1e30)
    _close(lse, torch.logsumexp(s, -1) / math.log(2), 1e-5 if dtype == torch.float32 else 1e-2, "lse (log2)")


@pytest.mark.parametrize("T", [25, 77, 197, 300])
def test_attention_bwd(ops, T):
    B, H = 2, 2
    qkv = _rand(B, T, 3 * H * 64, seed=T, scale=1.2, dtype=torch.bfloat16)
    dout = _rand(B, T, H * 64, seed=T + 1, dtype=torch.bfloat16)
    mask = None
    if T <= 77:
        mask = torch.zeros(B, T, dtype=torch.long)
        for b, n in enumerate((T, 9)):
            mask[b, :n] = 1
        mask = mask.cuda()
    out, lse = ops.attention_fwd(qkv, H, mask, save_lse=True)
    dqkv = ops.attention_bwd(qkv, out, dout, lse, H, mask)
    x = qkv.float().requires_grad_(True)
    _attn_ref(x, H, mask, 0.125).backward(dout.float())
    _close(dqkv, x.grad, 2e-2, f"attention bwd T={T}")
    if mask is not None:      # padded keys get an exactly-zero K/V gradient
        g = dqkv.view(B, T, 3, H, 64)
        assert float(g[1, 9:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("T", [1, 25, 77, 197, 300])
def test_attention_bwd_fp32_exact(ops, T):
    """Exact mode (a non-AMP run of the reference trains in fp32): forward and backward in fp32 MFMA against fp64 torch autograd, with
    the ragged key-padding mask of text batches and with attention dropout (mask regenerated from the hash)."""
    B, H = 3, 2
    qkv = _rand(B, T, 3 * H * 64, seed=T, scale=1.2)
    dout = _rand(B, T, H * 64, seed=T + 1)
    mask = None
    if T <= 77:
        mask = torch.zeros(B, T, dtype=torch.long)
        for b, n in enumerate((T, min(9, T), 1)):
            mask[b, :n] = 1
        mask = mask.cuda()
    for p, seed in ((0.0, 0), (0.1, 1234)):
        out, lse = ops.attention_fwd(qkv, H, mask, save_lse=True, drop_seed=seed, drop_p=p)
        dqkv = ops.attention_bwd(qkv, out, dout, lse, H, mask, drop_seed=seed, drop_p=p)
        keep = torch.ones(B * H * T * T, device="cuda")
        if p > 0:
            ops.dropout_apply_(keep, seed, p)
        keep = keep.view(B, H, T, T).double()
        x = qkv.double().requires_grad_(True)
        q, k, v = x.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
        sc = (q @ k.transpose(-1, -2)) * 0.125
        if mask is not None:
            sc = sc + (1.0 - mask[:, None, None, :].double()) * -10000.0
        ref = ((sc.softmax(-1) * keep) @ v).transpose(1, 2).reshape(B, T, H * 64)
        _close(out, ref.detach().float(), 1e-5, f"fp32 attention fwd T={T} p={p}")
        ref.backward(dout.double())
        _close(dqkv, x.grad.float(), 2e-5, f"fp32 attention bwd T={T} p={p}")
        if mask is not None and T > 9:      # padded keys get an exactly-zero K/V gradient
            assert float(dqkv.view(B, T, 3, H, 64)[1, 9:, 1:].abs().max()) == 0.0


@pytest.mark.parametrize("T", [25, 77, 200])
def test_attention_skip_padded_rows(ops, T):
    """skip_padded_rows (the packed text tower): the kernels work on the rows up to each sequence's last unmasked key only.  On those
    rows forward output, log-sum-exp and every gradient equal the dense call's bit for bit (same tiles, same order; with and without
    dropout); lengths cover one row, tile edges, a hole inside a mask and a fully masked sequence (which falls back to the dense work)."""
    B, H = 7, 2
    qkv = _rand(B, T, 3 * H * 64, seed=T, scale=1.2, dtype=torch.bfloat16)
    dout = _rand(B, T, H * 64, seed=T + 1, dtype=torch.bfloat16)
    lens = [T, 1, min(9, T), min(32, T), min(33, T), T - 1, 0]
    mask = torch.zeros(B, T, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    if T > 20:
        mask[2, 3] = 0                          # a hole: still masked as a key, still computed as a query
    for b, n in enumerate(lens):
        if n > 0:
            dout[b, n:] = 0                     # the caller's contract: rows past the last unmasked key carry no gradient
    mask = mask.cuda()
    for p, seed in ((0.0, 0), (0.1, 99)):
        out_d, lse_d = ops.attention_fwd(qkv, H, mask, save_lse=True, drop_seed=seed, drop_p=p)
        out_s, lse_s = ops.attention_fwd(qkv, H, mask, save_lse=True, drop_seed=seed, drop_p=p, skip_padded_rows=True)
        g_d = ops.attention_bwd(qkv, out_d, dout, lse_d, H, mask, drop_seed=seed, drop_p=p)
        g_s = ops.attention_bwd(qkv, out_s, dout, lse_s, H, mask, drop_seed=seed, drop_p=p, skip_padded_rows=True)
        for b, n in enumerate(lens):
            n = n if n > 0 else T                # nothing unmasked: dense
            assert torch.equal(out_s[b, :n], out_d[b, :n]) and torch.equal(lse_s[b, :, :n], lse_d[b, :, :n]), (T, b, p)
            assert torch.equal(g_s[b, :n], g_d[b, :n]), (T, b, p)
            # the public op's contract: rows the kernels skip come back as zeros (their true gradient), never as uninitialised memory
            assert not out_s[b, n:].any() and not g_s[b, n:].any() and not lse_s[b, :, n:].any(), (T, b, p)


@pytest.mark.parametrize("T", [25, 77, 200])
def test_attention_on_packed_rows_equals_dense(ops, T):
    """simseg_attention_fwd_rows / _bwd_rows: the ragged batch stored without its padding (sequence b = rows [row_start[b], row_start[b+1]))
    against the dense, prefix-masked call with skip_padded_rows: forward output, log-sum-exp, every gradient row and the q/k/v bias
    gradient agree to bf16 rounding, with and without dropout (the same dropout mask: the hash is indexed by sequence / query / key); lengths cover one token, tile edges, the full
    length and an EMPTY sequence; rows behind the last sequence (a caller's tile padding) come back as zeros."""
    B, H = 7, 2
    lens = [T, 1, min(9, T), min(32, T), min(33, T), 0, T - 1]
    mask = torch.zeros(B, T, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, :n] = 1
    mask = mask.cuda()
    qkv = _rand(B, T, 3 * H * 64, seed=T, scale=1.2, dtype=torch.bfloat16)
    dout = _rand(B, T, H * 64, seed=T + 1, dtype=torch.bfloat16)
    idx = mask.view(-1).nonzero().flatten()
    nv = idx.numel()
    pad = 5
    qp = torch.cat([qkv.view(B * T, -1)[idx], torch.full((pad, 3 * H * 64), float("nan"), device="cuda", dtype=torch.bfloat16)])
    dp = torch.cat([dout.view(B * T, -1)[idx], torch.zeros(pad, H * 64, device="cuda", dtype=torch.bfloat16)])
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.tensor(lens).cumsum(0)
    cu = cu.cuda()
    for p, seed in ((0.0, 0), (0.1, 99)):
        # the empty sequence: the dense call treats a fully masked sequence as unmasked - leave it out of the comparison
        out_d, lse_d = ops.attention_fwd(qkv, H, mask, save_lse=True, drop_seed=seed, drop_p=p, skip_padded_rows=True)
        cs_d = torch.zeros(3 * H * 64, device="cuda")
        dmask = dout.clone()
        dmask[5] = 0                                   # (its gradient rows do not exist in the packed layout)
        for b, n in enumerate(lens):
            dmask[b, n:] = 0
        g_d = ops.attention_bwd(qkv, out_d, dmask, lse_d, H, mask, drop_seed=seed, drop_p=p, skip_padded_rows=True)
        out_r, lse_r = ops.attention_fwd_rows(qp, H, cu, T, save_lse=True, drop_seed=seed, drop_p=p, n_real=nv)
        cs_r = torch.zeros(3 * H * 64, device="cuda")
        g_r = ops.attention_bwd_rows(qp, out_r, dp, lse_r, H, cu, T, drop_seed=seed, drop_p=p, colsum=cs_r, n_real=nv)
        assert not out_r[nv:].any() and not g_r[nv:].any()
        # (the unmasked forward instantiation tracks its running maximum over the real keys only, the masked one over the raw scores of
        #  a whole tile: the same function, roundings one bf16 ulp apart)
        _close(out_r[:nv], out_d.view(B * T, -1)[idx], 1e-2, f"packed rows forward T={T} p={p}")
        _close(g_r[:nv], g_d.view(B * T, -1)[idx], 1.5e-2, f"packed rows backward T={T} p={p}")
        for b, n in enumerate(lens):
            if n:
                _close(lse_r[b, :, :n], lse_d[b, :, :n], 1e-5, f"packed rows lse T={T} b={b} p={p}")
        want_cs = g_d.view(B * T, -1)[idx].float().sum(0)
        _close(cs_r, want_cs, 2e-3, f"qkv bias gradient from the packed rows T={T} p={p}")


@pytest.mark.parametrize("T", [77, 197, 600])
def test_attention_deferred_rescale_branch(ops, T):
    """The online softmax raises its running maximum (and rescales O, l) only when a tile's maximum exceeds it by more than a
    threshold - a rare, data-dependent branch that bounded random scores never take after the first tile.  Force it: a few
    late keys are made collinear with a few queries so that those rows' maxima jump by far more than the threshold in the second,
    third, ... tile, in some lanes of a wave only.  Forward (both kernels: resident for T <= 256, ring above / when forced) and the
    log-sum-exp the backward consumes, against fp32 torch."""
    B, H = 2, 3
    qkv = _rand(B, T, 3 * H * 64, seed=T, scale=0.6, dtype=torch.bfloat16)
    v5 = qkv.view(B, T, 3, H, 64)
    for j, (qi, ki, amp) in enumerate(((3, T - 2, 6.0), (T // 2, 70 if T > 70 else T - 1, 9.0), (T - 1, T // 2 + 1, 4.0), (40, 66 if T > 66 else 5, 12.0))):
        v5[j % B, ki, 1, j % H] = (v5[j % B, qi, 0, j % H].float() * amp).bfloat16()          # score ~ amp * |q|^2 / 8 >> threshold
    mask = None
    if T <= 77:
        mask = torch.ones(B, T, dtype=torch.long, device="cuda")
        mask[1, T - 9:] = 0
    ref = _attn_ref(qkv, H, mask, 0.125)
    q, k, _ = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    sc = q @ k.transpose(-1, -2) * 0.125
    if mask is not None:
        sc = sc.masked_fill(mask[:, None, None, :] == 0, -1e30)
    assert float(sc.max()) > 25.0                                   # the spikes are there (log2 units: > 36 >> 6)
    want_lse = torch.logsumexp(sc, -1) / math.log(2)
    for variant in (4, 1):                                           # 4: resident kernel for every T <= 256, 1: ring kernel
        ops.set_attention_variant(variant)
        try:
            out, lse = ops.attention_fwd(qkv, H, mask, save_lse=True)
        finally:
            ops.set_attention_variant(0)
        _close(out, ref, 1.5e-2, f"attention fwd with late maxima T={T} variant={variant}")
        _close(lse, want_lse, 1e-2, "lse (log2)")
    dout = _rand(B, T, H * 64, seed=T + 1, dtype=torch.bfloat16)
    x = qkv.float().requires_grad_(True)
    _attn_ref(x, H, mask, 0.125).backward(dout.float())
    _close(ops.attention_bwd(qkv, out, dout, lse, H, mask), x.grad, 2e-2, f"attention bwd with late maxima T={T}")


def _lse_ref(qkv, H, scale=0.125):
    B, T, _ = qkv.shape
    q, k, _ = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    return torch.logsumexp(q @ k.transpose(-1, -2) * scale, -1) / math.log(2)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,H", [(3, 512, 2), (2, 513, 3), (1, 576, 3), (2, 577, 2), (3, 600, 3), (11, 1025, 1), (1, 1024, 3), (2, 1072, 2), (2, 1089, 2), (1, 1153, 2),
                                   (1, 2305, 1)])
def test_attention_fwd_long_sequences(ops, B, T, H, dtype):
    """The 64-queries-per-wave forward (T >= 512, no mask / dropout; csrc/attn.hip attn_fwd_w64_kernel): class-token key as the softmax's
    start, full tiles from key 1, the masked last tile ((T - 1) % 64 != 0), blocks with idle waves, the key-split block for <= 64 leftover
    rows (513, 1025, 2305: one row; 576, 1072: both row groups of the key-split block in use; 577, 1089 -> one main block more), (batch x head)
    counts that are not multiples of the 8 XCDs -
    against fp32 torch on the same 16-bit operands, and against the ring kernel it replaces (variant 1)."""
    qkv = _rand(B, T, 3 * H * 64, seed=T, scale=1.3, dtype=dtype)
    want = _attn_ref(qkv, H, None, 0.125)
    tol = 1.5e-2 if dtype == torch.bfloat16 else 2.5e-3
    for variant in (7, 6, 0):          # two query blocks per wave (+ key-split blocks), one (small launches), the dispatcher's choice
        ops.set_attention_variant(variant)
        try:
            out, lse = ops.attention_fwd(qkv, H, None, save_lse=True)
        finally:
            ops.set_attention_variant(0)
        _close(out, want, tol, f"attention fwd T={T} {dtype} variant {variant}")
        _close(lse, _lse_ref(qkv, H), 1e-5, "lse (log2)")
    ops.set_attention_variant(1)
    try:
        ring, _ = ops.attention_fwd(qkv, H, None)
    finally:
        ops.set_attention_variant(0)
    _close(out, ring, tol, "w64 against the ring kernel")


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,T,H", [(2, 1025, 3), (1, 577, 2), (2, 700, 2), (3, 512, 1)])
def test_attention_fwd_qscaled(ops, B, T, H, dtype):
    """simseg_attention_fwd_qscaled: the q columns carry scale * log2(e) BEFORE their rounding to 16 bits (the towers fold the factor into
    timm Attention.qkv's q rows).  Reference: exact attention of the rounded operands with the factor divided out again - the kernel's
    exponent is its MFMA output, nothing is rounded twice, so the log-sum-exp agrees to fp32 rounding."""
    c = ops.attention_qscale(0.125)
    q32 = _rand(B, T, 3, H, 64, seed=T + 7, scale=1.3)
    q32[:, :, 0] *= c
    qkv = q32.view(B, T, 3 * H * 64).to(dtype)
    back = qkv.float().view(B, T, 3, H, 64).clone()
    back[:, :, 0] /= c
    back = back.view(B, T, 3 * H * 64)
    for variant in (7, 6):
        ops.set_attention_variant(variant)
        try:
            out, lse = ops.attention_fwd_qscaled(qkv, H, save_lse=True)
        finally:
            ops.set_attention_variant(0)
        _close(out, _attn_ref(back, H, None, 0.125), 1.5e-2 if dtype == torch.bfloat16 else 2.5e-3, f"attention fwd qscaled T={T} {dtype} variant {variant}")
        _close(lse, _lse_ref(back, H), 1e-5, "lse (log2)")
    with pytest.raises(Exception):
        ops.attention_fwd_qscaled(qkv[:, :300].contiguous(), H)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("T", [1025, 700])
def test_attention_fwd_long_recentre_branch(ops, T, dtype):
    """The long-sequence forward keeps a per-row offset that only has to keep the probabilities representable, and raises it - O, l rescaled,
    P recomputed from re-formed scores - when a half-row's sum over a key block leaves [0, 2^14] (fp16) / [0, 2^30] (bf16): a wave-uniform
    branch that bounded random scores never take.  Force it in every phase: keys made collinear with a few queries so that those rows'
    scores jump by 40-120 (log2 units) above the class-token score in the first, a middle and the last tile, in the first and the second
    32-key block, for rows of a main block, of a key-split block (T = 1025: row 1024) and of the masked last tile (T = 700); one row's
    scores overflow fp32 exponentials outright (2^200)."""
    B, H = 2, 2
    qkv = _rand(B, T, 3 * H * 64, seed=T + 11, scale=0.7, dtype=dtype)
    v5 = qkv.view(B, T, 3, H, 64)
    spikes = ((3, 5, 9.0), (100, 40, 14.0), (T - 1, T // 2 + 3, 11.0), (T - 1, T - 2, 16.0), (300, T - 40, 20.0), (511, 700 - 37 if T > 700 else 600, 12.0),
              (64, 33 + 64 * 7, 40.0))
    for j, (qi, ki, amp) in enumerate(spikes):
        v5[j % B, ki, 1, j % H] = (v5[j % B, qi, 0, j % H].float() * amp).to(dtype)
    q, k, _ = qkv.float().view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    sc = q @ k.transpose(-1, -2) * 0.125 / math.log(2)
    assert float(sc.max()) > 150.0 and float((sc.amax(-1) - sc[..., 0]).max()) > 100.0       # far above the class-token score, beyond 2^127
    want = _attn_ref(qkv, H, None, 0.125)
    tol = 1.5e-2 if dtype == torch.bfloat16 else 2.5e-3
    for variant in (7, 6):
        ops.set_attention_variant(variant)
        try:
            out, lse = ops.attention_fwd(qkv, H, None, save_lse=True)
        finally:
            ops.set_attention_variant(0)
        assert torch.isfinite(out.float()).all()
        _close(out, want, tol, f"long forward with late maxima T={T} variant {variant}")
        _close(lse, _lse_ref(qkv, H), 1e-5, "lse (log2)")
    ops.set_attention_variant(7)
    c = ops.attention_qscale(0.125)
    qs = qkv.clone()
    qs.view(B, T, 3, H, 64)[:, :, 0] = (v5[:, :, 0].float() * c).to(dtype)
    back = qs.float().view(B, T, 3, H, 64).clone()
    back[:, :, 0] /= c
    try:
        out2, lse2 = ops.attention_fwd_qscaled(qs, H, save_lse=True)
    finally:
        ops.set_attention_variant(0)
    _close(out2, _attn_ref(back.view(B, T, -1), H, None, 0.125), tol, f"qscaled forward with late maxima T={T}")
    _close(lse2, _lse_ref(back.view(B, T, -1), H), 1e-5, "lse (log2)")


@pytest.mark.parametrize("T", [1, 33, 64, 65, 129, 1025])
def test_attention_edge_lengths(ops, T):
    """Tile edges (one token, one over a 32 / 64 boundary, the 512^2 window) with a (batch x head) count that is not a multiple of
    the 8 XCDs, forward in both dtypes and backward."""
    B, H = 11, 1
    for dtype, tol in ((torch.float32, 1e-5), (torch.bfloat16, 1.5e-2)):
        qkv = _rand(B, T, 3 * H * 64, seed=T + 3, scale=1.3, dtype=dtype)
        out, lse = ops.attention_fwd(qkv, H, None, save_lse=True)
        _close(out, _attn_ref(qkv, H, None, 0.125), tol, f"attention fwd T={T} {dtype}")
    dout = _rand(B, T, H * 64, seed=T + 4, dtype=torch.bfloat16)
    dqkv = ops.attention_bwd(qkv, out, dout, lse, H, None)
    x = qkv.float().requires_grad_(True)
    _attn_ref(x, H, None, 0.125).backward(dout.float())
    _close(dqkv, x.grad, 2e-2, f"attention bwd T={T}")


@pytest.mark.parametrize("T", [33, 77, 197, 224, 256])
def test_attention_bwd_one_kernel_vs_two_passes(ops, T):
    """bf16 backward for T <= 256: the one-kernel form (default) against the two resident passes (variant 3) and the fp32 reference, with a
    ragged key mask and dropout; the one-kernel form hands dS tiles between waves in a fixed order, so two runs are bit-identical."""
    B, H = 3, 2
    qkv = _rand(B, T, 3 * H * 64, seed=T + 11, scale=1.2, dtype=torch.bfloat16)
    dout = _rand(B, T, H * 64, seed=T + 12, dtype=torch.bfloat16)
    mask = torch.zeros(B, T, dtype=torch.long)
    for b, n in enumerate((T, max(1, T // 3), 1)):
        mask[b, :n] = 1
    mask = mask.cuda()
    for m, p in ((None, 0.0), (mask, 0.0), (mask, 0.1)):
        out, lse = ops.attention_fwd(qkv, H, m, save_lse=True, drop_seed=9, drop_p=p)
        one = ops.attention_bwd(qkv, out, dout, lse, H, m, drop_seed=9, drop_p=p)
        again = ops.attention_bwd(qkv, out, dout, lse, H, m, drop_seed=9, drop_p=p)
        assert torch.equal(one, again)
        ops.set_attention_variant(3)
        try:
            two = ops.attention_bwd(qkv, out, dout, lse, H, m, drop_seed=9, drop_p=p)
        finally:
            ops.set_attention_variant(0)
        _close(one, two.float(), 6e-3, f"one kernel vs two passes T={T} mask={m is not None} p={p}")
        if p == 0.0:
            x = qkv.float().requires_grad_(True)
            _attn_ref(x, H, m, 0.125).backward(dout.float())
            _close(two, x.grad, 2e-2, f"two-pass attention bwd T={T}")
            _close(one, x.grad, 2e-2, f"one-kernel attention bwd T={T}")


@pytest.mark.parametrize("T", [25, 77, 197, 256, 300])
def test_attention_bwd_qkv_bias_gradient(ops, T):
    """colsum += column sums of dqkv (the q/k/v bias gradient): formed inside the one-kernel backward for T <= 256, by a column-sum pass
    otherwise (T = 300, fp32); accumulates into the caller's buffer; with skip_padded_rows only the rows the kernel works on count."""
    B, H = 5, 2
    qkv = _rand(B, T, 3 * H * 64, seed=T + 21, scale=1.2, dtype=torch.bfloat16)
    dout = _rand(B, T, H * 64, seed=T + 22, dtype=torch.bfloat16)
    out, lse = ops.attention_fwd(qkv, H, None, save_lse=True)
    cs = torch.full((3 * H * 64,), 2.0, device="cuda")
    dqkv = ops.attention_bwd(qkv, out, dout, lse, H, None, colsum=cs)
    _close(cs - 2.0, dqkv.float().sum((0, 1)), 2e-5, f"qkv bias gradient T={T}")
    assert torch.equal(dqkv, ops.attention_bwd(qkv, out, dout, lse, H, None))                  # dqkv itself does not change
    x32 = qkv.float()
    o32, l32 = ops.attention_fwd(x32, H, None, save_lse=True)
    cs32 = torch.zeros(3 * H * 64, device="cuda")
    d32 = ops.attention_bwd(x32, o32, dout.float(), l32, H, None, colsum=cs32)
    _close(cs32, d32.sum((0, 1)), 2e-5, f"qkv bias gradient fp32 T={T}")
    if T <= 256:
        mask = torch.zeros(B, T, dtype=torch.long)
        for b, n in enumerate((T, max(1, T // 2), 1, 7, T - 1)):
            mask[b, :min(n, T)] = 1
        mask = mask.cuda()
        dm = dout * mask[:, :, None].to(dout.dtype)
        for skip in (False, True):
            out, lse = ops.attention_fwd(qkv, H, mask, save_lse=True, drop_seed=3, drop_p=0.1, skip_padded_rows=skip)
            cs = torch.zeros(3 * H * 64, device="cuda")
            dqkv = ops.attention_bwd(qkv, out, dm, lse, H, mask, drop_seed=3, drop_p=0.1, skip_padded_rows=skip, colsum=cs)
            want = torch.where(mask[:, :, None].bool(), dqkv.float(), torch.zeros((), device="cuda")).sum((0, 1))      # (untouched rows may hold anything)
            _close(cs, want, 2e-5, f"qkv bias gradient, ragged mask, skip_padded_rows={skip} T={T}")


@pytest.mark.parametrize("B,T,H", [(3, 197, 4), (70, 145, 12), (40, 224, 3), (2, 130, 2)])
def test_attention_fwd_persistent_loader_wave_kernel(ops, B, T, H):
    """The opt-in persistent forward (attention variant 4: one block per CU walks its heads, a loader wave copies the next head's K / V
    into the other LDS stage, the compute waves software-pipeline their key tiles) computes the bits of the default resident kernel - more
    heads than blocks included - and both agree with the fp64 softmax attention."""
    qkv = _rand(B, T, 3 * H * 64, seed=T, dtype=torch.bfloat16)
    try:
        ops.set_attention_variant(4)
        out4, lse4 = ops.attention_fwd(qkv, H, None, scale=0.125, save_lse=True)
        ops.set_attention_variant(0)
        out0, lse0 = ops.attention_fwd(qkv, H, None, scale=0.125, save_lse=True)
    finally:
        ops.set_attention_variant(0)
    assert torch.equal(lse4, lse0)
    # (the persistent kernel forms all eight score MFMAs of the partial last tile; the masked block's probabilities are exactly 0 either way)
    assert torch.equal(out4, out0)
    q, k, v = [t.view(B, T, H, 64).transpose(1, 2).double() for t in qkv.chunk(3, -1)]
    want = ((q @ k.transpose(-1, -2) * 0.125).softmax(-1) @ v).transpose(1, 2).reshape(B, T, H * 64)
    assert float((out4.double() - want).abs().max()) < 2e-2


@pytest.mark.parametrize("B,T,H,ragged,drop", [(5, 197, 12, False, 0.0), (9, 77, 4, True, 0.1), (3, 224, 2, False, 0.0), (6, 33, 3, True, 0.0)])
def test_attention_on_plane_major_operands_equals_packed_rows(ops, B, T, H, ragged, drop):
    """simseg_attention_{fwd,bwd}_planes: qkv / dqkv as [3 H][rows][64] (a head's operand rows one contiguous run) through the same resident
    forward / one-kernel backward, addressed by two stride parameters - the bits of the packed-row entry points, dense and ragged batches,
    dropout included, with the q/k/v bias gradient."""
    g = torch.Generator(device="cuda").manual_seed(T)
    if ragged:
        lens = torch.randint(1, T + 1, (B,), device="cuda", generator=g)
        rs = torch.zeros(B + 1, dtype=torch.int32, device="cuda")
        rs[1:] = lens.cumsum(0)
        n = int(rs[-1])
        rows = (n + 255) // 256 * 256
    else:
        rs, n, rows = None, B * T, B * T
    qkv = torch.randn(rows, 3 * H * 64, device="cuda", generator=g).to(torch.bfloat16)
    qkvp = qkv.view(rows, 3 * H, 64).permute(1, 0, 2).contiguous()
    kw = dict(drop_seed=77, drop_p=drop)
    if ragged:
        oa, la = ops.attention_fwd_rows(qkv, H, rs, T, save_lse=True, n_real=n, **kw)
    else:
        oa, la = ops.attention_fwd(qkv.view(B, T, -1), H, None, save_lse=True)
        oa = oa.reshape(rows, -1)
    ob, lb = ops.attention_fwd_planes(qkvp, H, B, T, rs, save_lse=True, n_real=n, **kw)
    assert torch.equal(oa[:n], ob[:n])
    if not ragged:
        assert torch.equal(la, lb)
    do = torch.randn(rows, H * 64, device="cuda", generator=g).to(torch.bfloat16)
    ca, cb = torch.zeros(3 * H * 64, device="cuda"), torch.zeros(3 * H * 64, device="cuda")
    if ragged:
        da = ops.attention_bwd_rows(qkv, oa, do, la, H, rs, T, n_real=n, colsum=ca, **kw)
    else:
        da = ops.attention_bwd(qkv.view(B, T, -1), oa.view(B, T, -1), do.view(B, T, -1), la, H, None, colsum=ca).reshape(rows, -1)
    db = ops.attention_bwd_planes(qkvp, ob, do, lb, H, B, T, rs, n_real=n, colsum=cb, **kw)
    assert torch.equal(da.view(rows, 3 * H, 64)[:n], db.permute(1, 0, 2)[:n])
    assert torch.equal(ca, cb)


def test_attention_masked_length_limit(ops):
    qkv = _rand(1, 1100, 3 * 64, seed=1, dtype=torch.bfloat16)
    mask = torch.ones(1, 1100, dtype=torch.long, device="cuda")
    with pytest.raises(RuntimeError, match="limited to"):
        ops.attention_fwd(qkv, 1, mask)
    out, _ = ops.attention_fwd(qkv.float(), 1, mask)         # the fp32 kernel has no such limit
    _close(out, _attn_ref(qkv.float(), 1, mask, 0.125), 1e-5, "masked fp32 T=1100")


def test_attention_dropout(ops):
    """Dropout on attention probabilities: mean preserved, same mask regenerated by the backward."""
    B, H, T = 2, 2, 77
    qkv = _rand(B, T, 3 * H * 64, seed=5, dtype=torch.bfloat16)
    base, lse = ops.attention_fwd(qkv, H, None, save_lse=True)
    outs = torch.stack([ops.attention_fwd(qkv, H, None, drop_seed=s, drop_p=0.1)[0].float() for s in range(200)])
    _close(outs.mean(0), base, 0.1, "dropout keeps the expectation")
    assert not torch.equal(outs[0], outs[1])
    # backward with dropout == autograd through an explicit masked softmax using the regenerated mask
    seed, p = 77, 0.1
    out, lse = ops.attention_fwd(qkv, H, None, save_lse=True, drop_seed=seed, drop_p=p)
    keep = torch.ones(B * H * T * T, device="cuda")
    ops.dropout_apply_(keep, seed, p)                      # same hash, same linear index (bh*T + q)*T + key
    keep = keep.view(B, H, T, T)
    x = qkv.float().requires_grad_(True)
    q, k, v = x.view(B, T, 3, H, 64).permute(2, 0, 3, 1, 4)
    ref = (((q @ k.transpose(-1, -2)) * 0.125).softmax(-1) * keep @ v).transpose(1, 2).reshape(B, T, H * 64)
    _close(out, ref, 1.5e-2, "dropout fwd vs explicit mask")
    dout = _rand(B, T, H * 64, seed=6, dtype=torch.bfloat16)
    ref.backward(dout.float())
    _close(ops.attention_bwd(qkv, out, dout, lse, H, None, drop_seed=seed, drop_p=p), x.grad, 2e-2, "dropout bwd")


@pytest.mark.parametrize("smoothing", [0.0, 0.1])
def test_nce_rows(ops, smoothing):
    N1, N2, rank = 24, 96, 2
    f1 = F.normalize(_rand(N1, 64, seed=1), dim=-1)
    f2 = F.normalize(_rand(N2, 64, seed=2), dim=-1)
    ign = torch.zeros(N1, device="cuda"); ign[3] = 1.0
    temp = torch.tensor(0.05, device="cuda")
    sims = ops.gemm(f1, f2)
    s_ref = (f1 @ f2.T).requires_grad_(True)
    t_ref = temp.clone().requires_grad_(True)
    z = s_ref / torch.clamp(t_ref, 0.001, 0.5)
    logp = F.log_softmax(z, -1)
    tgt = torch.arange(rank * N1, (rank + 1) * N1, device="cuda")
    loss_rows = (1 - smoothing) * -logp.gather(1, tgt[:, None])[:, 0] + smoothing * -logp.mean(-1)
    loss = (loss_rows * (1 - ign)).mean()
    loss.backward()
    keep = ign < 1
    acc = (z[keep].argmax(1) == tgt[keep]).float().mean()
    out3 = ops.nce_rows(sims, temp, rank * N1, ign, smoothing)
    _close(out3[0], loss.detach(), 1e-5, "nce loss")
    _close(out3[1], acc, 1e-6, "nce acc")
    _close(out3[2], t_ref.grad, 1e-4, "nce dtemp")
    _close(sims, s_ref.grad, 1e-4, "nce dsims")
    # clamped temperature passes no gradient
    out3 = ops.nce_rows(ops.gemm(f1, f2), torch.tensor(0.7, device="cuda"), rank * N1, None, 0.0)
    assert float(out3[2]) == 0.0


def test_retrieval_rank(ops, golden):
    from conftest import tt
    g = golden("retrieval")
    img, gi = tt(g["uni_emb"]).cuda(), tt(g["uni_gid"]).cuda()
    txt, gt = tt(g["txt"]).cuda(), tt(g["gid_txt"]).cuda()
    for left, lg, right, rg, key in ((img, gi, txt, gt, "i2t"), (txt, gt, img, gi, "t2i")):
        sim = ops.gemm(left, right)
        has, rank = ops.retrieval_rank(sim, lg, rg)
        c = ops.recall_counts(has, rank).cpu().numpy()
        np.testing.assert_allclose(c[1:] / c[0], g[key], atol=1e-7)
        s = left @ right.T
        match = lg[:, None] == rg[None, :]
        best = torch.where(match, s, torch.full_like(s, -float("inf"))).max(1)[0]
        assert torch.equal(rank.long(), (s > best[:, None]).sum(1))
        # the reverse direction from the same matrix: columns rank their rows; against the reference's recalls of the swapped call
        hasc, rankc = ops.retrieval_rank_cols(sim, lg, rg)
        cc = ops.recall_counts(hasc, rankc).cpu().numpy()
        np.testing.assert_allclose(cc[1:] / cc[0], g["t2i" if key == "i2t" else "i2t"], atol=1e-7)
        bestc = torch.where(match, s, torch.full_like(s, -float("inf"))).max(0)[0]
        assert torch.equal(hasc.bool(), bestc > -float("inf"))
        assert torch.equal(rankc.long()[hasc.bool()], (s > bestc[None, :]).sum(0)[hasc.bool()])


def test_adamw_step(ops):
    n = 4096 * 3
    p, g = _rand(n, seed=1), _rand(n, seed=2)
    ref = p.clone().requires_grad_(True)
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-3)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    p16 = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    for step in range(1, 4):
        ref.grad = g * step
        opt.step()
        ops.adamw_step(p, g * step, m, v, p16, 1e-3, (0.9, 0.98), 1e-6, 1e-3, step)
    _close(p, ref.detach(), 1e-6, "adamw")
    assert torch.equal(p16, p.bfloat16())


def test_adamw_multi_tensor_optimizer():
    """simseg_amd.optim.AdamW (one launch for all tensors) == torch.optim.AdamW over several steps and shapes."""
    from simseg_amd.optim import AdamW
    shapes = [(768, 768), (3072,), (1, 197, 768), (), (70000, 3)]
    ours = [torch.nn.Parameter(_rand(*s, seed=i) if s else torch.tensor(0.02, device="cuda")) for i, s in enumerate(shapes)]
    ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
    o1 = AdamW(ours, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-2)
    o2 = torch.optim.AdamW(ref, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-2)
    for step in range(4):
        for i, (a, b) in enumerate(zip(ours, ref)):
            g = _rand(*shapes[i], seed=100 * step + i) if shapes[i] else torch.tensor(0.3 * (step + 1), device="cuda")
            a.grad, b.grad = g.clone(), g.clone()
        o1.step(); o2.step()
    for a, b in zip(ours, ref):
        _close(a.detach(), b.detach(), 1e-5, "multi-tensor adamw")


def test_adamw_under_a_gradscaler_unscales_and_skips_on_the_device():
    """torch.amp.GradScaler's contract for optimizers that handle the scale themselves (`_step_supports_amp_scaling`): scaler.step() hands
    over the loss scale and the overflow flag as DEVICE tensors; the kernel unscales the gradients, skips the whole update when the flag is
    set, and counts on the device the steps it really took (bias corrections: skipped steps do not count, as in torch's fused Adam).  The
    trajectory equals torch.optim.AdamW + torch's scaler on the same scaled gradients, for torch's scaler and for this package's (one
    read-only kernel as the overflow check); no call reads the device."""
    from simseg_amd.optim import AdamW, GradScaler
    shapes = [(768, 768), (3072,), (), (70000, 3)]
    for Scaler in (torch.amp.GradScaler, GradScaler):
        ours = [torch.nn.Parameter(_rand(*s, seed=i) if s else torch.tensor(0.02, device="cuda")) for i, s in enumerate(shapes)]
        ref = [torch.nn.Parameter(p.detach().clone()) for p in ours]
        o1 = AdamW(ours, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-2)
        o2 = torch.optim.AdamW(ref, lr=1e-3, betas=(0.9, 0.98), eps=1e-6, weight_decay=1e-2)
        s1, s2 = Scaler("cuda", init_scale=1024.0, growth_interval=3), torch.amp.GradScaler("cuda", init_scale=1024.0, growth_interval=3)
        for sc in (s1, s2):
            sc.scale(torch.zeros((), device="cuda"))          # (lazy init of the scale tensor)
        for step in range(7):
            scale = s2.get_scale()
            for i, (a, b) in enumerate(zip(ours, ref)):
                g = (_rand(*shapes[i], seed=100 * step + i) if shapes[i] else torch.tensor(0.3 * (step + 1), device="cuda")) * scale
                if step in (1, 4) and i == (step % len(shapes)):
                    g = g.clone()
                    g.view(-1)[g.numel() // 2] = float("inf") if step == 1 else float("nan")       # an overflowing step
                a.grad, b.grad = g.clone(), g.clone()
            torch.cuda.synchronize()
            if step > 0:                                      # (the first call builds the launch plan: host -> device copies of its index tables)
                torch.cuda.set_sync_debug_mode("error")
            try:
                s1.step(o1)
                s1.update()
            finally:
                torch.cuda.set_sync_debug_mode("default")
            s2.step(o2); s2.update()
            assert s1.get_scale() == s2.get_scale(), (step, s1.get_scale(), s2.get_scale())
        assert o1.steps_taken() == 5                              # seven calls, two skipped
        for a, b in zip(ours, ref):
            _close(a.detach(), b.detach(), 1e-5, f"adamw under {Scaler.__module__}.GradScaler")
        # back to plain steps: the host counter picks up where the device one stood
        for a, b in zip(ours, ref):
            a.grad, b.grad = torch.ones_like(a), torch.ones_like(b)
        o1.step(); o2.step()
        assert o1.steps_taken() == 6
        for a, b in zip(ours, ref):
            _close(a.detach(), b.detach(), 1e-5, "plain step after the scaled ones")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,C,K", [(324, 21, 512), (1024, 171, 512), (777, 60, 512), (130, 81, 384), (4096, 256, 512), (64, 1, 128)])
def test_patch_text_sim_fused(ops, dtype, M, C, K):
    """Fused K14 kernel (row L2-normalise + all-class contraction) vs torch on the same inputs."""
    x = (_rand(M, K, seed=1) * 3.0).to(dtype)
    t = F.normalize(_rand(C, K, seed=2), dim=-1).to(dtype)
    ref = F.normalize(x.float(), dim=-1) @ t.float().T
    got = ops.patch_text_sim(x, t)
    assert got.shape == (M, C)
    _close(got, ref, 1e-5 if dtype == torch.float32 else 1e-5, f"fused sim map {M}x{C}x{K}")
    plain = ops.patch_text_sim(x, t, normalize=False)
    _close(plain, x.float() @ t.float().T, 1e-5, "plain contraction")


def test_gemm_auto_takes_the_persistent_kernel_for_many_tiles(ops):
    """From 600 full 256x256 tiles on (the image tower's training GEMMs) the automatic dispatch launches the persistent ping-pong kernel;
    same bits as the per-tile kernel; problems with a ragged edge or fewer tiles stay on the per-tile kernel."""
    from simseg_amd.lib import raw
    M, N, K = 256 * 210, 768, 768
    a = _rand(M, K, seed=1, dtype=torch.bfloat16)
    b = _rand(N, K, seed=2, scale=K ** -0.5, dtype=torch.bfloat16)
    bias, res = _rand(N, seed=3), _rand(M, N, seed=4)
    y = ops.gemm(a, b, bias=bias)
    assert raw("simseg_gemm_last_variant") == 10
    y32 = ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32)
    assert raw("simseg_gemm_last_variant") == 10
    ops.set_gemm_variant(3)
    try:
        assert torch.equal(ops.gemm(a, b, bias=bias), y) and raw("simseg_gemm_last_variant") == 3
        assert torch.equal(ops.gemm(a, b, bias=bias, residual=res, out_dtype=torch.float32), y32)
    finally:
        ops.set_gemm_variant(0)
    ops.gemm(a[:M - 7], b, bias=bias)
    assert raw("simseg_gemm_last_variant") == 3                       # a ragged last row tile
    ops.gemm(a[:256 * 150], b, bias=bias)
    assert raw("simseg_gemm_last_variant") == 3                       # 450 tiles


@pytest.mark.parametrize("M,N,K,variant", [(256 * 40, 768, 768, 0), (256 * 210, 768, 768, 0), (256 * 210, 768, 768, 3), (256 * 33, 3072, 768, 0)])
def test_gemm_saved_derivative_as_tile_blocked_image(ops, M, N, K, variant):
    """act 5 / 6: the fc1 forward stores GELU' as it lies in the accumulator registers of its tile, the dgrad through fc2 reads it back the
    same way - same results as the row-major pair (act 3 / 4), on the per-tile and the persistent kernel, and across the two (a tensor
    written by one kernel is read by the other: the layout is a function of the tile, not of the kernel)."""
    from simseg_amd.lib import raw
    assert raw("simseg_gemm_aux_blocked_ok", M, N, K) == 1 and raw("simseg_gemm_aux_blocked_ok", M + 8, N, K) == 0
    a = _rand(M, K, seed=1, dtype=torch.bfloat16)
    b = _rand(N, K, seed=2, scale=K ** -0.5, dtype=torch.bfloat16)
    g = _rand(M, K, seed=6, dtype=torch.bfloat16)
    w2 = _rand(K, N, seed=7, scale=K ** -0.5, dtype=torch.bfloat16)          # [out = K, in = N]: the dgrad through it is g @ w2 -> [M, N]
    bias = _rand(N, seed=3)
    ops.set_gemm_variant(variant)
    try:
        d_row = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        y_row = ops.gemm(a, b, bias=bias, act=3, aux_out=d_row)
        d_blk = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        y_blk = ops.gemm(a, b, bias=bias, act=5, aux_out=d_blk)
        assert raw("simseg_gemm_last_variant") in (3, 10)
        assert torch.equal(y_row, y_blk)
        assert torch.equal(d_row.flatten().float().sort().values, d_blk.flatten().float().sort().values)       # the same values, tile-blocked order
        assert not torch.equal(d_row, d_blk)
        cs_row, cs_blk = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        dx_row = ops.gemm(g, w2, trans_b=True, act=4, aux=d_row, colsum=cs_row)
        dx_blk = ops.gemm(g, w2, trans_b=True, act=6, aux=d_blk, colsum=cs_blk)
        want = (g.float() @ w2.float()) * d_row.float()
        _close(dx_blk, want, 1e-2, "times the tile-blocked derivative")
        _close(dx_row, want, 1e-2, "times the row-major derivative")
        _close(cs_blk, want.sum(0), 2e-4, "column sums (fp32 products)")
        # act 7 / 8: the same image with one byte per element (uniform grid of 1.264 / 255 over GELU''s range): same forward output; the
        # backward product is as accurate against the exact derivative as with the 16-bit image (both measured against fp64)
        d8 = torch.empty(M, N, device="cuda", dtype=torch.uint8)
        y8 = ops.gemm(a, b, bias=bias, act=7, aux_out=d8)
        assert torch.equal(y8, y_row)
        cs8 = torch.zeros(N, device="cuda")
        dx8 = ops.gemm(g, w2, trans_b=True, act=8, aux=d8, colsum=cs8)
        pre = a.double() @ b.double().t() + bias.double()
        gexact = 0.5 * (1 + torch.erf(pre / 2 ** 0.5)) + pre * torch.exp(-pre * pre / 2) / (2 * 3.141592653589793) ** 0.5
        exact = (g.double() @ w2.double()) * gexact
        rms = lambda t: float(((t.double() - exact) ** 2).mean().sqrt() / (exact ** 2).mean().sqrt())     # noqa: E731
        e16, e8 = rms(dx_blk), rms(dx8)
        assert e8 < 1.25 * e16 + 1e-4 and e8 < 5e-3, (e16, e8)
        assert float((dx8.float() - want).abs().max()) <= 4e-3 * float((g.float() @ w2.float()).abs().max()) + 1e-2 * float(want.abs().max())
        _close(cs8, dx8.float().sum(0), 2e-3, "column sums of the 8-bit product")
        # the other kernel reads what this one wrote
        ops.set_gemm_variant(3 if variant == 0 else 0)
        _close(ops.gemm(g, w2, trans_b=True, act=6, aux=d_blk), dx_blk.float(), 1e-6, "blocked image across kernels")
        _close(ops.gemm(g, w2, trans_b=True, act=8, aux=d8), dx8.float(), 1e-6, "8-bit image across kernels")
    finally:
        ops.set_gemm_variant(0)
    with pytest.raises(RuntimeError, match="act 5"):
        ops.gemm(a[:M - 8], b, bias=bias, act=5, aux_out=d_blk[:M - 8])


# ---- ragged caption batches: the row maps of the packed text tower, built by one kernel (round 4) ---------------------------------------
@pytest.mark.parametrize("B,L,kind", [(6, 11, "holes"), (512, 77, "prefix"), (2048, 25, "prefix"), (37, 200, "holes"), (1, 5, "prefix"),
                                      (1500, 3, "empty_rows"), (8, 64, "full")])
def test_ragged_maps_kernel_equals_the_index_ops(B, L, kind):
    from simseg_amd import towers
    g = torch.Generator().manual_seed(B * 1000 + L)
    lens = torch.randint(1, L + 1, (B,), generator=g)
    if kind == "empty_rows":
        lens[::7] = 0
    if kind == "full":
        lens[:] = L
    mask = (torch.arange(L)[None] < lens[:, None]).long()
    if kind == "holes":
        mask = mask * (torch.rand(B, L, generator=g) > 0.2).long()
    want = towers._plan_torch(mask, 256)
    m = mask.cuda()
    got = towers.ragged_plan(m, 256)                           # count + hole flag read back
    assert got.nv == want.nv and torch.equal(got.idx.cpu(), want.idx) and torch.equal(got.inv.cpu(), want.inv)
    assert (got.cu is None) == (want.cu is None)
    if want.cu is not None:
        assert torch.equal(got.cu.cpu(), want.cu)
    assert towers.ragged_plan(m, 256) is got                   # cached on the tensor
    m2 = mask.cuda()
    host_lens = mask.sum(1)                                    # what a loader knows before the host->device copy
    got2 = towers.ragged_plan(m2, 256, lengths=host_lens)      # nothing read back
    assert got2.nv == want.nv and torch.equal(got2.idx.cpu(), want.idx) and torch.equal(got2.inv.cpu(), want.inv)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(host_lens, 0).to(torch.int32)
    assert torch.equal(got2.cu.cpu(), cu)
    towers._poll_length_checks(block=True)                     # the deferred comparison with the device count: clean


def test_ragged_maps_wrong_host_lengths_are_reported_one_step_late():
    from simseg_amd import towers
    towers._poll_length_checks(block=True)
    mask = (torch.arange(20)[None] < torch.tensor([5, 20, 1, 9])[:, None]).long().cuda()
    towers.ragged_plan(mask, 8, lengths=[5, 20, 1, 8])          # one token short: sized without a read, so it cannot fail here
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="caption_lengths do not describe the attention_mask"):
        towers.ragged_plan(torch.ones(2, 4, dtype=torch.long).cuda(), 8, lengths=[4, 4])
    with pytest.raises(ValueError):
        towers.ragged_plan(torch.ones(2, 4, dtype=torch.long).cuda(), 8, lengths=[4])
