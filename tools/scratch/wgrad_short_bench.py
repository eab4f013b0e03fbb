import sys, torch
sys.path.insert(0, '.')
from simseg_amd import ops
def t(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps * 1e3
for M, N, K in [(768, 768, 21760), (768, 768, 12288), (2304, 768, 21760), (768, 3072, 21760), (512, 768, 21760)]:
    dy = torch.randn(K, M, device='cuda').bfloat16(); x = torch.randn(K, N, device='cuda').bfloat16()
    out = torch.zeros(M, N, device='cuda')
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    sk = max(1, min((K + 63) // 64, (1024 + tiles - 1) // tiles, 64))
    us = t(lambda: ops.gemm(dy, x, trans_a=True, trans_b=True, out=out, accumulate=True, splitk=sk))
    from simseg_amd.lib import raw
    print(M, N, K, f"{us:.1f} us", 2.0 * M * N * K / us / 1e6, "TFLOP/s variant", raw("simseg_gemm_last_variant"))
