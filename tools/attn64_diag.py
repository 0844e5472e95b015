#!/usr/bin/env python3
"""Per-phase cycle accounting of attn_fwd64_kernel (library built with -DA64_DIAG=1; OPADPO_LIB_PATH selects it): one block's four waves,
cycles per phase summed over its tile iterations.  0 barrier wait | 1/6 S phase of sub-tile 0/1 | 2/7 softmax_0 | 3/8 softmax_1 + O_0 | 4/9 O_1 | 5 K commit + V fetch | 10 V commit + mask."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import lib as L  # noqa: E402
BF = torch.bfloat16
dev = "cuda:0"
S, nh, hd, Ln = 22, 32, 128, int(os.environ.get("GB_L", 1087))
H = nh * hd
qkv = torch.randn(S * Ln, 3 * H, device=dev).to(BF)
o = torch.empty(S * Ln, H, dtype=BF, device=dev)
lse = torch.zeros(S * nh * Ln + 64, device=dev)
L.set_flags(True, 1)
for _ in range(3):
    L.call("opadpo_attn_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H,
           lse.data_ptr(), None, S, Ln, nh, hd, 1, hd ** -0.5, 0, 0, L.stream())
torch.cuda.synchronize()
d = lse[S * nh * Ln:].view(4, 16).cpu()
names = ["barrier", "S(0)", "sm0(0)", "sm1+PV0(0)", "PV1(0)", "stageK", "S(1)", "sm0(1)", "sm1+PV0(1)", "PV1(1)", "stageV", "-"]
for w in range(4):
    n = max(float(d[w, 12]), 1.0)
    print("wave %d  tiles %d  " % (w, n) + "  ".join("%s %.0f" % (names[i], float(d[w, i]) / n) for i in range(11)) + "   total/tile %.0f" % (float(d[w, :11].sum()) / n))
