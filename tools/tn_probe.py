"""Diagnostic: gemm_tn rate vs shape (loop-bound vs epilogue-bound)."""
import sys, os, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opa-dpo_amd"))
from opadpo_amd import lib as L
L.load(); dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / iters
for tr in (1, 9):
    L.set_flags(True, tr)
    for (M, N1, N2) in ((32362, 4096, 4096), (32362, 4096, 256), (8192, 4096, 256), (131072, 4096, 256), (131072, 1024, 256)):
        p = torch.randn(M, N1, device=dev).to(torch.bfloat16); q = torch.randn(M, N2, device=dev).to(torch.bfloat16)
        c = torch.zeros(N1, N2, device=dev)
        t = timeit(lambda: L.gemm_tn(p, q, c))
        print("tr", tr, M, N1, N2, "ms %.3f" % (t * 1e3), "TF/s %.0f" % (2.0 * M * N1 * N2 / t / 1e12), flush=True)
