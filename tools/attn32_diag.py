"""Where a workgroup of the 32-row attention forward spends its cycles (diagnostics build: DIAG_SRC=attention tools/build_diag.sh a32diag:"-DOPADPO_ATTN32_DIAG=1",
then OPADPO_LIB_PATH=opa-dpo_amd/lib/libopadpo_hip_a32diag.so python tools/attn32_diag.py).  Per shape: shader cycles of wave 0 before its first tile (geometry,
Q, first K / V tile), per tile of the loop, and from the loop's end to the last store - next to the launch time and the workgroup / tile counts."""
import ctypes as C
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "opa-dpo_amd"))
from opadpo_amd import lib as L
lib = L.load()
lib.opadpo_debug_attn32_read.argtypes = [C.c_void_p, C.c_int]
dev = torch.device("cuda:0")
BF = torch.bfloat16
out5 = (C.c_ulonglong * 5)()
nh, hd = 32, 128
H = nh * hd


def run_case(name, S, Ln, causal, seg, km):
    qkv = torch.randn(S * Ln, 3 * H, device=dev).to(BF)
    o = torch.empty(S * Ln, H, dtype=BF, device=dev)
    lse = torch.empty(S, nh, Ln, device=dev)
    f = lambda: L.call("opadpo_attn_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H, lse.data_ptr(),
                       km.data_ptr() if km is not None else None, S, Ln, nh, hd, causal, hd ** -0.5, seg[0], seg[1], L.stream())
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    lib.opadpo_debug_attn32_read(out5, 1)
    n = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    lib.opadpo_debug_attn32_read(out5, 1)
    pro, loop, epi, wgs, tiles = [float(x) for x in out5]
    ms = e0.elapsed_time(e1) / n
    print(f"{name:28s} S {S} L {Ln}: {ms * 1e3:8.1f} us | workgroups {wgs / n:6.0f}, tiles per workgroup {tiles / wgs:5.2f} | cycles per workgroup: before the first tile "
          f"{pro / wgs:7.0f}, loop {loop / wgs:8.0f} ({loop / max(tiles, 1):6.0f} per tile), after the loop {epi / wgs:7.0f} | fixed share {(pro + epi) / (pro + loop + epi):.3f}", flush=True)


S = int(os.environ.get("GB_S", 16))
for Ln in (1087, 2048, 4096):
    for causal in (1, 0):
        run_case("dense causal" if causal else "dense full", S, Ln, causal, (0, 0), None)
# the packed bench shape: 22 rows of [703 prefix | 384 | 384] with a realistic key mask (tools/gemm_bench.py GB_ONLY=attn2)
S2, pfx, T, K = 22, 703, 384, 2
Ln = pfx + K * T
km = torch.ones(S2, Ln, dtype=torch.uint8, device=dev)
g = torch.Generator().manual_seed(0)
for s_ in range(S2):
    km[s_, 576:576 + int(torch.randint(0, 64, (1,), generator=g))] = 0
    for k in range(K):
        n_ = int(torch.randint(64, 384, (1,), generator=g))
        km[s_, pfx + k * T + n_: pfx + (k + 1) * T] = 0
run_case("packed pairs (bench shape)", S2, Ln, 1, (pfx, T), km)
