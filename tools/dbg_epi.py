import sys, os, torch
sys.path.insert(0, "opa-dpo_amd")
from opadpo_amd import lib as L
L.load(); dev = torch.device("cuda:0")
L.set_flags(23, True)
for (M, N, K) in ((256, 256, 64), (300, 256, 128), (700, 512, 64)):
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g).to(dev).to(torch.bfloat16); b = torch.randn(N, K, generator=g).to(dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    L.gemm_nt(a, b, out)
    torch.cuda.synchronize()
    want = a.float() @ b.float().t()
    bad = ((out.float() - want).abs() > 0.05 * want.abs() + 0.1)
    print(M, N, K, "bad", int(bad.sum()), "of", M * N)
    rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
    print(" bad rows", rows[:40], "... n=", len(rows)); print(" bad cols", cols[:40], "... n=", len(cols))
    idx = bad.nonzero()[:6]
    for r, c in idx.tolist():
        print("  ", r, c, float(out[r, c]), float(want[r, c]))
    # does each wrong value appear elsewhere in the same row?
    r = idx[0][0].item()
    print("  row", r, "out[:8]", out[r, :8].float().tolist(), "want[:8]", [round(x, 3) for x in want[r, :8].tolist()])
