"""residual-epilogue cost by K and tail width (diagnostic): fp32 out + fp32 residual vs fp32 out, M = 24576, N = 4096"""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opa-dpo_amd"))
from opadpo_amd import lib as L
L.load()
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=20, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
M, H = int(os.environ.get("GB_M", 24576)), 4096
res = torch.randn(M, H, device=dev); o32 = torch.empty(M, H, device=dev)
for K, r in ((4096, 0), (4160, 0), (4224, 0), (4352, 0), (4096, 64), (4096, 256), (8192, 0), (11008, 0), (11008, 256)):
    x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(H, K, device=dev) * 0.02).to(BF)
    kw = dict(a2=torch.randn(M, r, device=dev).to(BF), b2=(torch.randn(H, r, device=dev) * 0.02).to(BF)) if r else {}
    for rep in range(2):
        a = timeit(lambda: L.gemm_nt(x, w, o32, residual=res, **kw))
        c = timeit(lambda: L.gemm_nt(x, w, o32, **kw))
        print(f"K={K} tail={r}: residual {a:.3f} ms | plain {c:.3f} ms | +{a - c:.3f}", flush=True)
