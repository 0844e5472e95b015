#!/usr/bin/env python3
"""Yardstick only (never used by the product): torch scaled_dot_product_attention (the ROCm flash-attention backend) at the shape of tools/gemm_bench.py's plain causal
attention row - S sequences x 32 heads x L = 1087 x head_dim 128, bf16 - forward and forward+backward, next to this library's kernels at the same shape."""
import json
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
dev = torch.device("cuda:0")
S, nh, L, hd = int(os.environ.get("AV_S", 8)), 32, int(os.environ.get("AV_L", 1087)), 128
q, k, v = (torch.randn(S, nh, L, hd, device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3))
fl_f = 4.0 * S * nh * hd * (L * L / 2)


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


res = {}
for name, ctx in (("flash", torch.nn.attention.SDPBackend.FLASH_ATTENTION), ("efficient", torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION)):
    try:
        with torch.nn.attention.sdpa_kernel(ctx):
            with torch.no_grad():
                t_f = timeit(lambda: F.scaled_dot_product_attention(q, k, v, is_causal=True))
            o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
            do = torch.randn_like(o)
            t_fb = timeit(lambda: torch.autograd.grad(F.scaled_dot_product_attention(q, k, v, is_causal=True), (q, k, v), do))
        res[name] = dict(fwd_ms=round(t_f, 4), fwd_TF=round(fl_f / t_f / 1e9, 1), fwd_bwd_ms=round(t_fb, 4), bwd_TF=round(2.5 * fl_f / max(t_fb - t_f, 1e-6) / 1e9, 1))
    except Exception as e:  # noqa: BLE001
        res[name] = "unavailable: " + str(e)[:100]
print(json.dumps(dict(S=S, L=L, nh=nh, hd=hd, **res)))
