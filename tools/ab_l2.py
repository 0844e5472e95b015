#!/usr/bin/env python3
"""Is the 256x256 GEMM's gap to the vendor library on deep-K products memory latency?  Timing-only experiment: the same launch with the A rows, the B rows or
both ALIASED (row stride 0: every row of the operand is the same memory, so its panel is 2 K bytes and always an L2 / L1 hit) against the normal launch, on a
deep-K shape and - as the control for the power effect of repeated operand values - on a K = 4352 shape.  Alternating blocks."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import lib as L  # noqa: E402

BF = torch.bfloat16
L.load()
dev = torch.device("cuda:0")
M = int(os.environ.get("AB_M", 24576))
rounds, iters = int(os.environ.get("AB_ROUNDS", 6)), int(os.environ.get("AB_ITERS", 20))
for name, N, K in (("down", 4096, 11264), ("dgrad_gu", 4096, 22016), ("o", 4096, 4352), ("gate_up", 22016, 4352)):
    a = torch.randn(M, K, device=dev).to(BF)
    b = (torch.randn(N, K, device=dev) * 0.02).to(BF)
    out = torch.empty(M, N, dtype=BF, device=dev)
    modes = {"normal": (a, b), "A_aliased": (a[:1].expand(M, K), b), "B_aliased": (a, b[:1].expand(N, K)), "both_aliased": (a[:1].expand(M, K), b[:1].expand(N, K))}
    tot = {m: 0.0 for m in modes}
    for m, (x, y) in modes.items():
        for _ in range(5):
            L.gemm_nt(x, y, out)
    for r in range(rounds):
        keys = list(modes) if r % 2 == 0 else list(modes)[::-1]
        for m in keys:
            x, y = modes[m]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            L.gemm_nt(x, y, out)
            e0.record()
            for _ in range(iters):
                L.gemm_nt(x, y, out)
            e1.record()
            torch.cuda.synchronize()
            tot[m] += e0.elapsed_time(e1)
    fl = 2.0 * M * N * K
    print(json.dumps(dict(name=name, M=M, N=N, K=K, **{m + "_TF": round(fl / (tot[m] / (rounds * iters)) / 1e9, 1) for m in modes})), flush=True)
