#!/usr/bin/env python3
"""Default dispatch (variant 10: the streaming walk where the launcher chooses it) against the one-tile-per-workgroup 256x256 kernel (variant 31) per shape,
ALTERNATING the two in short blocks so that clock / power drift hits both alike (tools/gemm_bench.py measures each variant in one block: its first row is also
the process's first kernel).  AB_M = row counts (default: 24576 and the bench's 32362), AB_ROUNDS x AB_ITERS launches per variant and shape."""
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
shapes = [("qkv", 12288, 4096, 256, 4096), ("o", 4096, 4096, 256, 0), ("gate_up", 22016, 4096, 256, 11008), ("down", 4096, 11008, 256, 0),
          ("lm_head", 32000, 4096, 0, 0), ("dgrad_gu", 4096, 22016, 0, 0), ("dgrad_qkv", 4096, 12288, 0, 0)]
if os.environ.get("AB_MODEL") == "13b":      # LLaVA-1.5-13B widths (hidden 5120, ffn 13824), LoRA r = 256
    shapes = [("qkv", 15360, 5120, 256, 5120), ("o", 5120, 5120, 256, 0), ("gate_up", 27648, 5120, 256, 13824), ("down", 5120, 13824, 256, 0),
              ("dgrad_gu", 5120, 27648, 0, 0), ("dgrad_qkv", 5120, 15360, 0, 0)]
if os.environ.get("AB_SHAPES"):
    shapes = [sh for sh in shapes if sh[0] in os.environ["AB_SHAPES"].split(",")]
rounds, iters = int(os.environ.get("AB_ROUNDS", 6)), int(os.environ.get("AB_ITERS", 20))
variants = [int(v) for v in os.environ.get("AB_VARIANTS", "10,31").split(",")]
if os.environ.get("AB_VENDOR", "1") == "1":
    variants.append(-1)      # -1: torch.matmul (hipBLASLt) on the K-concatenated operands - the yardstick, never used by the product
for M in [int(m) for m in os.environ.get("AB_M", "24576,32362").split(",")]:
    for name, N, K1, K2, grp in shapes:
        a1 = torch.randn(M, K1, device=dev).to(BF)
        b1 = (torch.randn(N, K1, device=dev) * 0.02).to(BF)
        out = torch.empty(M, N, dtype=BF, device=dev)
        kw = {}
        if K2:
            G = N // grp if grp else 1
            kw = dict(a2=torch.randn(M, G * K2, device=dev).to(BF), b2=(torch.randn(N, K2, device=dev) * 0.02).to(BF), a2_group_n=grp, a2_group_stride=K2 if grp else 0)
        va, vb = (torch.randn(M, K1 + K2, device=dev).to(BF), (torch.randn(N, K1 + K2, device=dev) * 0.02).to(BF)) if -1 in variants else (None, None)

        def launch(v):
            if v < 0:
                torch.matmul(va, vb.t())
            else:
                L.gemm_nt(a1, b1, out, **kw)
        tot = {v: 0.0 for v in variants}
        for v in variants:
            if v >= 0:
                L.set_flags(v, True)
            for _ in range(5):
                launch(v)
        for r in range(rounds):
            for v in (variants if r % 2 == 0 else variants[::-1]):
                if v >= 0:
                    L.set_flags(v, True)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                launch(v)
                e0.record()
                for _ in range(iters):
                    launch(v)
                e1.record()
                torch.cuda.synchronize()
                tot[v] += e0.elapsed_time(e1)
        ms = {v: tot[v] / (rounds * iters) for v in variants}
        fl = 2.0 * M * N * (K1 + K2)
        print(json.dumps(dict(name=name, M=M, N=N, K=K1 + K2, **{f"{'vendor' if v < 0 else 'v%d' % v}_ms": round(ms[v], 4) for v in variants}, **{f"{'vendor' if v < 0 else 'v%d' % v}_TF": round(fl / ms[v] / 1e9, 1) for v in variants},
                              first_over_second=round(ms[variants[0]] / ms[variants[1]], 4))), flush=True)
L.set_flags(True, True)
