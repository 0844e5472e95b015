import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import lib as L
BF = torch.bfloat16; dev = "cuda:0"
S, nh, hd, Ln = 1, 1, 128, int(os.environ.get("GB_L", 256))
causal = int(os.environ.get("GB_CAUSAL", 0))
H = nh * hd
torch.manual_seed(0)
qkv = torch.randn(S * Ln, 3 * H, device=dev).to(BF)
o = torch.zeros(S * Ln, H, dtype=BF, device=dev)
lse = torch.zeros(S * nh * Ln + 64, device=dev)
L.set_flags(True, 1)
L.call("opadpo_attn_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H,
       lse.data_ptr(), None, S, Ln, nh, hd, causal, hd ** -0.5, 0, 0, L.stream())
torch.cuda.synchronize()
q, k, v = [qkv[:, i * H:(i + 1) * H].float() for i in range(3)]
sc = (q @ k.T) * hd ** -0.5
if causal: sc = sc.masked_fill(~torch.tril(torch.ones(Ln, Ln, dtype=torch.bool, device=dev)), float("-inf"))
want = torch.softmax(sc, -1) @ v
err = (o.float() - want).abs()
print("max err", float(err.max()), "per-row max (first 128 rows):")
rm = err.max(1).values.cpu()
for r0 in range(0, min(Ln, 256), 32):
    print(r0, " ".join("%.2f" % x for x in rm[r0:r0 + 32].tolist()))
cm = err.max(0).values.cpu()
print("per-col max:", " ".join("%.2f" % x for x in cm.tolist()))
wl = torch.logsumexp(sc, -1)
print("lse err max", float((lse[:Ln] - wl).abs().max()))
le = (lse[:Ln] - wl).abs().cpu()
print("lse err rows:", " ".join("%.0e" % x for x in le[:64].tolist()))
print("o row0:", o[0, :16].float().cpu().tolist())
print("want row0:", want[0, :16].cpu().tolist())
print("ratio row0:", (o[0, :16].float() / want[0, :16]).cpu().tolist())
print("ratio row1:", (o[1, :8].float() / want[1, :8]).cpu().tolist())
print("ratio row33:", (o[33, :8].float() / want[33, :8]).cpu().tolist())
