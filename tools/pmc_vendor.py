#!/usr/bin/env python3
"""A few launches of one GEMM shape on the library's default dispatch, on the one-tile-per-workgroup 256x256 kernel and on torch.matmul (hipBLASLt) - the
workload of tools/pmc_vendor.sh (rocprofv3 --pmc passes: what does the vendor kernel do differently on the deep-K products?).  PV_SHAPE = name of the shape."""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import lib as L  # noqa: E402

BF = torch.bfloat16
L.load()
dev = torch.device("cuda:0")
M = int(os.environ.get("PV_M", 24576))
shapes = {"qkv": (12288, 4352), "lm_head": (32000, 4096), "down": (4096, 11264), "dgrad_gu": (4096, 22016), "o": (4096, 4352), "gate_up": (22016, 4352), "dgrad_qkv": (4096, 12288)}
N, K = shapes[os.environ.get("PV_SHAPE", "down")]
a = torch.randn(M, K, device=dev).to(BF)
b = (torch.randn(N, K, device=dev) * 0.02).to(BF)
out = torch.empty(M, N, dtype=BF, device=dev)
for v in [int(x) for x in os.environ.get('PV_VARIANTS', '31,-1').split(',')]:
    if v >= 0:
        L.set_flags(v, True)
    for _ in range(int(os.environ.get("PV_ITERS", 4))):
        if v < 0:
            torch.matmul(a, b.t(), out=out)
        else:
            L.gemm_nt(a, b, out)
torch.cuda.synchronize()
