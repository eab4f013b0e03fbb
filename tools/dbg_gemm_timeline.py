import sys, torch
sys.path.insert(0, '/root/repo')
from simseg_amd import ops
for (M, N, K) in ((256, 256, 128), (256, 256, 768), (16384, 768, 128), (100864, 768, 768)):
    a = torch.randn(M, K, device='cuda').bfloat16(); b = torch.randn(N, K, device='cuda').bfloat16()
    ops.set_gemm_variant(102)
    for _ in range(3):
        out = ops.gemm(a, b)
    torch.cuda.synchronize()
    d = out.view(-1).view(torch.int64)[:4].cpu().tolist()
    print(M, N, K, "cycles: pre-loop", d[0], "first wait+barrier", d[1], "k-loop rest", d[2], "epilogue(skipped)", d[3])
ops.set_gemm_variant(0)
