import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import lib as L
L.load(); L.set_flags(True, True)
dev = torch.device("cuda:0"); BF = torch.bfloat16
S, Ln, nh, hd = 3, 480, 2, 128
H = nh * hd
g = torch.Generator(device="cpu").manual_seed(21)
qkv = (torch.randn(S * Ln, 3 * H, generator=g) * 0.7).to(BF).to(dev)
km = torch.ones(S, Ln, dtype=torch.uint8, device=dev)
km[0, :9] = 0; km[0, 300:340] = 0; km[1, 250:340] = 0; km[1, 400:] = 0; km[2, 128:] = 0
o = torch.full((S * Ln, H), 9.0, dtype=BF, device=dev)
lse = torch.zeros(S, nh, Ln, device=dev)
L.call("opadpo_attn_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H, lse.data_ptr(), km.data_ptr(), S, Ln, nh, hd, L.CAUSAL_SKIP_MASKED_Q, hd ** -0.5, 0, 0, L.stream())
torch.cuda.synchronize()
o3 = o.view(S, Ln, H).float()
for s_ in range(S):
    nz = (o3[s_].abs().amax(-1) != 0)
    masked = km[s_] == 0
    bad = (nz & masked).nonzero().flatten().tolist()
    # group into ranges
    rng = []
    for r in bad:
        if rng and rng[-1][1] == r - 1: rng[-1][1] = r
        else: rng.append([r, r])
    print("seq", s_, "masked rows with nonzero output:", rng, "| nines left:", int((o3[s_] == 9.0).all(-1).sum()))
    for h in range(nh):
        print("   head", h, "nonzero masked rows:", int(((o3[s_][:, h*hd:(h+1)*hd].abs().amax(-1) != 0) & masked).sum()))
