#!/usr/bin/env python
"""Achieved TFLOP/s and fraction of the dense MFMA peak for the two kernels BASELINE.json's target names -- the dense
patch x class-text similarity map (K14) and fused attention (K5) -- at the ViT-B shapes of configs 3 and 4."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from simseg_amd import ops  # noqa: E402
from simseg_amd.heads import patch_text_similarity  # noqa: E402

PEAK = {"bf16": 2.5e15, "fp32": 157.3e12}


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    out = {}
    g = torch.Generator(device="cuda").manual_seed(0)
    for dt, tdt in (("fp32", torch.float32), ("bf16", torch.bfloat16)):
        # K14: 64 windows of 512x512 -> [64*1024, 512] x [171, 512]^T (+ row normalisation)
        proj = torch.randn(64, 1024, 512, device="cuda", generator=g).to(tdt)
        text = torch.nn.functional.normalize(torch.randn(171, 512, device="cuda", generator=g), dim=-1)
        s = timeit(lambda: patch_text_similarity(proj, text, compute_dtype=tdt))
        fl = 2.0 * 64 * 1024 * 512 * 171
        byts = 64 * 1024 * 512 * proj.element_size() + 171 * 512 * proj.element_size() + 64 * 1024 * 171 * 4
        out[f"sim_map_{dt}"] = {"shape": "[65536,512]x[171,512]^T", "ms": round(s * 1e3, 4), "tflops": round(fl / s / 1e12, 1),
                                "frac_mfma_peak": round(fl / s / PEAK[dt], 4), "GBps": round(byts / s / 1e9, 1), "frac_hbm_8TBps": round(byts / s / 8e12, 4)}
        for name, B, T in (("vitb_224", 512, 197), ("bert_77", 512, 77), ("vitb_512", 16, 1025)):
            H = 12
            qkv = torch.randn(B, T, 3 * H * 64, device="cuda", generator=g).to(tdt)
            s = timeit(lambda: ops.attention_fwd(qkv, H, None, save_lse=(dt == "bf16")))
            fl = 4.0 * B * H * T * T * 64
            out[f"attn_fwd_{dt}_{name}"] = {"B": B, "T": T, "H": H, "ms": round(s * 1e3, 4), "tflops": round(fl / s / 1e12, 1),
                                            "frac_mfma_peak": round(fl / s / PEAK[dt], 4)}
            if dt == "bf16":
                o, lse = ops.attention_fwd(qkv, H, None, save_lse=True)
                do = torch.randn_like(o)
                s = timeit(lambda: ops.attention_bwd(qkv, o, do, lse, H, None))
                fl = 10.0 * B * H * T * T * 64
                out[f"attn_bwd_bf16_{name}"] = {"B": B, "T": T, "H": H, "ms": round(s * 1e3, 4), "tflops": round(fl / s / 1e12, 1),
                                                "frac_mfma_peak": round(fl / s / PEAK[dt], 4)}
    # the evaluation towers' form of the long-sequence forward (round 6): q columns carry scale * log2(e) (towers._wt_qscaled),
    # simseg_attention_fwd_qscaled; both 16-bit types, the seg-eval batch sizes; fp16 runs through the same source compiled with -DSS_HALF
    c = ops.attention_qscale(0.125)
    for dt, tdt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        for name, B, T, H in (("vitb_512", 16, 1025, 12), ("vitb_512", 63, 1025, 12), ("vitb_512", 256, 1025, 12), ("vitb_384", 256, 577, 12),
                              ("vits_512", 256, 1025, 6)):
            qkv = torch.randn(B, T, 3 * H * 64, device="cuda", generator=g)
            qkv.view(B, T, 3, H * 64)[:, :, 0] *= c
            qkv = qkv.to(tdt)
            s = timeit(lambda: ops.attention_fwd_qscaled(qkv, H))
            fl = 4.0 * B * H * T * T * 64
            out[f"attn_fwd_qscaled_{dt}_{name}_B{B}"] = {"B": B, "T": T, "H": H, "ms": round(s * 1e3, 4), "tflops": round(fl / s / 1e12, 1),
                                                          "frac_mfma_peak": round(fl / s / PEAK["bf16"], 4)}
    for k, v in out.items():
        print(k, json.dumps(v))


if __name__ == "__main__":
    main()
