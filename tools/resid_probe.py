import os, sys, torch
sys.path.insert(0, "/root/repo/opa-dpo_amd")
from opadpo_amd import lib as L
L.load()
dev = torch.device("cuda:0"); BF = torch.bfloat16
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters
M, H = 24576, 4096
for name, K in (("o", 4096), ("down", 11008)):
    x = torch.randn(M, K, device=dev).to(BF); w = (torch.randn(H, K, device=dev) * 0.02).to(BF)
    t = torch.randn(M, 256, device=dev).to(BF); b2 = (torch.randn(H, 256, device=dev) * 0.02).to(BF)
    res = torch.randn(M, H, device=dev); o32 = torch.empty(M, H, device=dev); o16 = torch.empty(M, H, dtype=BF, device=dev)
    for tail in (False, True):
        kw = dict(a2=t, b2=b2) if tail else {}
        a = timeit(lambda: L.gemm_nt(x, w, o32, residual=res, **kw))
        b = timeit(lambda: L.gemm_nt(x, w, o16, **kw))
        c = timeit(lambda: L.gemm_nt(x, w, o32, **kw))
        print(f"{name} K={K} tail={tail}: fp32 out + fp32 residual {a:.3f} ms | bf16 out {b:.3f} ms | fp32 out no residual {c:.3f} ms")
xr = torch.randn(M, H, device=dev); wn = torch.ones(H, dtype=BF, device=dev); y = torch.empty(M, H, dtype=BF, device=dev); rstd = torch.empty(M, device=dev)
part = torch.randn(M, H, device=dev); xo = torch.empty(M, H, device=dev)
a = timeit(lambda: L.call("opadpo_rmsnorm_fwd", L.ptr(xr), 1, L.ptr(wn), L.ptr(y), L.ptr(rstd), M, H, 1e-5, L.stream()))
b = timeit(lambda: L.call("opadpo_rmsnorm_sum_fwd", L.ptr(xr), 1, L.ptr(part), 1, M * H, L.ptr(wn), L.ptr(xo), L.ptr(y), L.ptr(rstd), M, H, 1e-5, L.stream()))
print(f"rmsnorm_fwd {a:.3f} ms | rmsnorm_sum_fwd (1 fp32 partial, 14 B/elem) {b:.3f} ms")
