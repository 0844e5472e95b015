"""Where a tile of the streaming 256x256 GEMM spends its cycles (diagnostics build: tools/build_diag.sh w4sdiag:"-DOPADPO_W4S_DIAG=1", then
OPADPO_LIB_PATH=opa-dpo_amd/lib/libopadpo_hip_w4sdiag.so python tools/w4s_diag.py).  Per shape: cycles per tile of wave 0 in the K-loop block,
the epilogue and the per-tile set-up, and the K-loop's cycles per K-tile."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opa-dpo_amd"))
from opadpo_amd import lib as L
lib = L.load()
lib.opadpo_debug_w4s_read.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
M = int(os.environ.get("GB_M", 32362))
out4 = (C.c_ulonglong * 4)()
for name, N, K, f32 in (("qkv", 12288, 4096, 0), ("o", 4096, 4096, 0), ("o_f32", 4096, 4096, 1), ("gate_up", 22016, 4096, 0), ("lm_head", 32000, 4096, 1), ("down8k", 4096, 8192, 0)):
    A = (torch.randn(M, K, device=dev) * 0.5).bfloat16(); B = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    Cc = torch.empty(M, N, device=dev, dtype=torch.float32 if f32 else torch.bfloat16)
    def run():
        L.call("opadpo_gemm_nt", A.data_ptr(), K, B.data_ptr(), K, K, None, 0, None, 0, 0, 0, 0, 0, 0, Cc.data_ptr(), N, f32, None, 0, 0, None, M, N, 1.0, 0, L.stream())
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    lib.opadpo_debug_w4s_read(out4, 1)
    n = 20
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        run()
    e1.record(); torch.cuda.synchronize()
    lib.opadpo_debug_w4s_read(out4, 1)
    a, e, g, t = [float(x) for x in out4]
    ms = e0.elapsed_time(e1) / n
    nt = K // 64
    print(f"{name:8s} M {M} N {N} K {K} f32 {f32}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.0f} TF/s | per tile: K-loop {a / t:.0f} cycles ({a / t / nt:.0f} per K-tile), epilogue {e / t:.0f}, set-up {g / t:.0f}  (tiles {t / n:.0f} per launch)", flush=True)
