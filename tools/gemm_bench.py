#!/usr/bin/env python3
"""Micro-benchmark of the bf16 MFMA GEMMs at the LLaVA-1.5-7B seq512 shapes (run on the GPU box)."""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "opa-dpo_amd"))
from opadpo_amd import lib as L  # noqa: E402

BF = torch.bfloat16


def timeit(fn, iters=10, warm=3):
    iters = int(os.environ.get("GB_ITERS", iters))      # GB_ITERS=300: long enough for the socket's sustained power state
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def main():
    L.load()
    dev = torch.device("cuda:0")
    res = []
    M = int(os.environ.get("GB_M", 8 * 1087))
    shapes = [("qkv", 12288, 4096, 256, 4096), ("o", 4096, 4096, 256, 0), ("gate_up", 22016, 4096, 256, 11008),
              ("down", 4096, 11008, 256, 0), ("lora_t", 768, 4096, 0, 0), ("lm_head", 32000, 4096, 0, 0)]
    only = os.environ.get("GB_ONLY", "")
    if only == "lora":     # the small-N projections of the LoRA path at the bench row count (22 packed pairs)
        M = int(os.environ.get("GB_M", 22 * 1471))
        shapes = [("a_qkv", 768, 4096, 0, 0), ("a_o", 256, 4096, 0, 0), ("a_gu", 512, 4096, 0, 0), ("a_d", 256, 11008, 0, 0),
                  ("dt_2r", 512, 11008, 0, 0)]
        only = "gemm" if os.environ.get("GB_VARIANTS") else "yard"
    if only == "attn2":    # training attention at the bench shape: packed pairs [703 prefix | 384 | 384], realistic key mask
        S, nh, hd, pfx, T, K = int(os.environ.get("GB_S", 22)), 32, 128, 703, 384, 2
        Ln, H = pfx + K * T, nh * hd
        qkv = torch.randn(S * Ln, 3 * H, device=dev).to(BF)
        o = torch.empty(S * Ln, H, dtype=BF, device=dev)
        lse = torch.empty(S, nh, Ln, device=dev)
        km = torch.ones(S, Ln, dtype=torch.uint8, device=dev)
        g = torch.Generator().manual_seed(0)
        for s_ in range(S):
            km[s_, 576:576 + int(torch.randint(0, 64, (1,), generator=g))] = 0          # left-padded query
            for k in range(K):
                n = int(torch.randint(64, 384, (1,), generator=g))
                km[s_, pfx + k * T + n: pfx + (k + 1) * T] = 0                            # right-padded response
        pairs = pfx * pfx / 2 + K * (T * pfx + T * T / 2)
        fl = S * nh * 4 * hd * pairs
        dqkv = torch.empty(S * Ln, 3 * H, dtype=BF, device=dev)
        delta = torch.empty(S, nh, Ln, device=dev)
        do = torch.randn(S * Ln, H, device=dev).to(BF)
        L.set_flags(True, int(os.environ.get("GB_TR", 1)))
        for seg in ((pfx, T),):
            f = lambda: L.call("opadpo_attn_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H,
                               lse.data_ptr(), km.data_ptr(), S, Ln, nh, hd, 1, hd ** -0.5, seg[0], seg[1], L.stream())
            t = timeit(f, iters=5, warm=2)
            res.append(dict(kernel="attn_fwd_packed", ms=t * 1e3, tflops=fl / t / 1e12))
            print(res[-1], flush=True)
            f = lambda: L.call("opadpo_attn_bwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(),
                               do.data_ptr(), H, lse.data_ptr(), km.data_ptr(), dqkv.data_ptr(), dqkv.data_ptr() + 2 * H, dqkv.data_ptr() + 4 * H,
                               None, delta.data_ptr(), S, Ln, nh, hd, 1, hd ** -0.5, seg[0], seg[1], L.stream())
            t = timeit(f, iters=5, warm=2)
            res.append(dict(kernel="attn_bwd_packed", ms=t * 1e3, tflops=2.5 * fl / t / 1e12))
            print(res[-1], flush=True)
        return
    if only == "attn3":    # dense causal attention (no key mask, no segments): the kernels' clean-tile ceiling at the 7B head geometry
        S, nh, hd, Ln = int(os.environ.get("GB_S", 22)), 32, 128, int(os.environ.get("GB_L", 1087))
        H = nh * hd
        qkv = torch.randn(S * Ln, 3 * H, device=dev).to(BF)
        o = torch.empty(S * Ln, H, dtype=BF, device=dev)
        lse = torch.empty(S, nh, Ln, device=dev)
        causal = int(os.environ.get("GB_CAUSAL", 1))        # 0: dense non-causal (the guide's attention ladder is quoted on it)
        fl = S * nh * 4 * hd * (Ln * Ln / 2 if causal else Ln * Ln)
        dqkv = torch.empty(S * Ln, 3 * H, dtype=BF, device=dev)
        delta = torch.empty(S, nh, Ln, device=dev)
        do = torch.randn(S * Ln, H, device=dev).to(BF)
        L.set_flags(True, int(os.environ.get("GB_TR", 1)))
        f = lambda: L.call("opadpo_attn_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H,
                           lse.data_ptr(), None, S, Ln, nh, hd, causal, hd ** -0.5, 0, 0, L.stream())
        t = timeit(f, iters=5, warm=2)
        print(dict(kernel="attn_fwd_dense_causal" if causal else "attn_fwd_dense_full", L=Ln, ms=t * 1e3, tflops=fl / t / 1e12), flush=True)
        f = lambda: L.call("opadpo_attn_bwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(),
                           do.data_ptr(), H, lse.data_ptr(), None, dqkv.data_ptr(), dqkv.data_ptr() + 2 * H, dqkv.data_ptr() + 4 * H,
                           None, delta.data_ptr(), S, Ln, nh, hd, causal, hd ** -0.5, 0, 0, L.stream())
        t = timeit(f, iters=5, warm=2)
        print(dict(kernel="attn_bwd_dense_causal" if causal else "attn_bwd_dense_full", L=Ln, ms=t * 1e3, tflops=2.5 * fl / t / 1e12), flush=True)
        return
    if only == "tail":     # partial last round of 256x256 tiles: row-tile count sweep at the ragged bench row counts
        for name, N, K1, K2, grp in shapes[:4]:
            for R in [int(v) for v in os.environ.get("GB_RT", "96,97,100,104,108,112").split(",")]:
                M_ = R * 256
                a1 = torch.randn(M_, K1, device=dev).to(BF)
                b1 = (torch.randn(N, K1, device=dev) * 0.02).to(BF)
                out = torch.empty(M_, N, dtype=BF, device=dev)
                G = N // grp if grp else 1
                kw = dict(a2=torch.randn(M_, G * K2, device=dev).to(BF), b2=(torch.randn(N, K2, device=dev) * 0.02).to(BF),
                          a2_group_n=grp, a2_group_stride=K2 if grp else 0)
                t = timeit(lambda: L.gemm_nt(a1, b1, out, **kw))
                tiles = R * (N // 256)
                res.append(dict(name=name, R=R, tiles=tiles, rounds=tiles / 256, ms=t * 1e3, tflops=2.0 * M_ * N * (K1 + K2) / t / 1e12))
                print(res[-1], flush=True)
        return
    if only == "cube":     # yardstick shapes of the micro-architecture guide's 256^2 template: 4096^3 and 8192^3, uniform [-1,1)
        for n in (4096, 8192):
            a1 = (torch.rand(n, n, device=dev) * 2 - 1).to(BF)
            b1 = (torch.rand(n, n, device=dev) * 2 - 1).to(BF)
            out = torch.empty(n, n, dtype=BF, device=dev)
            if os.environ.get("GB_CUBE_BCAST"):      # diagnostics: every row of both operands is the SAME row (stride 0) - all operand fetches hit the caches
                a1, b1 = a1[:1].expand(n, n), b1[:1].expand(n, n)
            for variant in [int(v) for v in os.environ.get("GB_VARIANTS", "31,17,4,-1").split(",")]:
                if variant >= 0:
                    L.set_flags(variant, True)
                    t = timeit(lambda: L.gemm_nt(a1, b1, out), iters=20, warm=5)
                else:
                    t = timeit(lambda: torch.matmul(a1, b1.t(), out=out), iters=20, warm=5)
                res.append(dict(kernel={17: "p8_256", 16: "w4_256", 8: "pp256", 4: "x128", -1: "hipBLASLt"}.get(variant, f"v{variant}"), n=n, ms=t * 1e3, tflops=2.0 * n ** 3 / t / 1e12))
                print(res[-1], flush=True)
        L.set_flags(10, True)
        return
    if only == "tn":       # LoRA wgrad shapes of one decoder layer at the bench M
        shapes_tn = [("dB_d", 4096, 256, 0, 0), ("dA_d", 256, 11008, 0, 0), ("dB_gu", 22016, 256, 11008, 256), ("dA_gu", 512, 4096, 0, 0),
                     ("dB_o", 4096, 256, 0, 0), ("dA_o", 256, 4096, 0, 0), ("dB_qkv", 12288, 256, 4096, 256), ("dA_qkv", 768, 4096, 0, 0)]
        tot = 0.0
        for name, N1, N2, gn1, gs in shapes_tn:
            G = N1 // gn1 if gn1 else 1
            pm = torch.randn(M, N1, device=dev).to(BF)
            qm = torch.randn(M, N2 * G, device=dev).to(BF)
            c = torch.zeros(N1, N2, device=dev)
            kw = dict(q_group_n1=gn1, q_group_stride=gs) if gn1 else {}
            t = timeit(lambda: L.gemm_tn(pm, qm, c, **kw))
            tot += t
            res.append(dict(kernel="gemm_tn", name=name, M=M, N1=N1, N2=N2, ms=t * 1e3, tflops=2.0 * M * N1 * N2 / t / 1e12,
                            GBps=(M * (N1 + N2 * G) * 2) / t / 1e9))
            print(res[-1], flush=True)
        print("total ms per layer", tot * 1e3)
        return
    if only == "skinny":   # decode-sized GEMMs: weight streaming rate; weights rotated over > 512 MB so MALL cannot hold them
        for M_ in [int(v) for v in os.environ.get("GB_MS", "8,16,32,64").split(",")]:
            for name, N, K1, K2, grp in shapes:
                if name == "lora_t":
                    continue
                copies = max(2, int(6.0e8 // (N * K1 * 2)) + 1)
                ws = [(torch.randn(N, K1, device=dev) * 0.02).to(BF) for _ in range(copies)]
                a1 = torch.randn(M_, K1, device=dev).to(BF)
                out = torch.empty(M_, N, dtype=BF, device=dev)
                for label, variant, tr in (("skinny", 10, 1), ("skinny16", 10, 17), ("tile128", 4, 1), ("hipBLASLt", -1, 1)):
                    if variant >= 0:
                        L.set_flags(variant, tr)
                    it = [0]

                    def fn():
                        w = ws[it[0] % copies]
                        it[0] += 1
                        if variant == 10:
                            with L.decode_schedule():
                                L.gemm_nt(a1, w, out)
                        elif variant >= 0:
                            L.gemm_nt(a1, w, out)
                        else:
                            torch.matmul(a1, w.t(), out=out)
                    t = timeit(fn, iters=3 * copies, warm=copies)
                    res.append(dict(kernel=label, name=name, M=M_, N=N, K=K1, us=t * 1e6, GBps=N * K1 * 2 / t / 1e9))
                    print(res[-1], flush=True)
                del ws
        L.set_flags(10, True)
        json.dump(res, open(os.path.join(REPO, "gpurun_out", "gemm_skinny.json"), "w"), indent=1)
        return
    if only == "dec":      # the decode GEMMs for 17..64 tokens (opadpo_gemm_nt_decode): ring kernel vs the register-streaming kernel
        lib = L.load()
        for M_ in [int(v) for v in os.environ.get("GB_MS", "64,32").split(",")]:
            for name, N, K1, mode, splits in (("qkv", 12288, 4096, 0, 1), ("o", 4096, 4096, 1, 0), ("gate_up", 22016, 4096, 2, 1),
                                              ("down", 4096, 11008, 1, 0), ("lm_head", 32000, 4096, 1, 1)):
                copies = max(2, int(6.0e8 // (N * K1 * 2)) + 1)
                ws = [(torch.randn(N, K1, device=dev) * 0.02).to(BF) for _ in range(copies)]
                pad = int(os.environ.get("GB_LDA_PAD", 0))      # row stride of the activations = K + pad elements (L2 channel spread)
                a1 = torch.randn(M_, K1 + pad, device=dev).to(BF)[:, :K1]
                ref = None
                for label, v in (("ring", 1), ("x48", 3 | (1 << 2)), ("x64", 3 | (2 << 2)), ("x128", 3 | (3 << 2)), ("auto", 0)):
                    L.set_flags(10, 1 | (v << 5))
                    S = lib.opadpo_gemm_nt_decode_splits(N, K1, splits) if mode == 1 else 1      # the K-split target depends on the kernel in use
                    out = torch.empty(S * M_ * (N // 2 if mode == 2 else N), dtype=torch.float32 if mode == 1 else BF, device=dev)
                    it = [0]

                    def fn():
                        w = ws[it[0] % copies]
                        it[0] += 1
                        L.call("opadpo_gemm_nt_decode", L.ptr(a1), a1.stride(0), L.ptr(w), K1, K1, L.ptr(out), N // 2 if mode == 2 else N, mode, M_, N, splits, L.stream())
                    t = timeit(fn, iters=3 * copies, warm=copies)
                    it[0] = 0
                    fn()
                    torch.cuda.synchronize()
                    got = out.float().view(S, M_, -1).sum(0) if mode == 1 else out.float().view(M_, -1)
                    if ref is None:
                        ref = got.clone()
                    err = float((got - ref).abs().max() / ref.abs().max())
                    res.append(dict(kernel=label, name=name, M=M_, N=N, K=K1, splits=S, us=round(t * 1e6, 2), GBps=round(N * K1 * 2 / t / 1e9), rel_vs_ring=err))
                    print(res[-1], flush=True)
                del ws
        L.set_flags(10, True)
        return
    if only == "pmc":      # few launches of the two big shapes, default variant only (PMC passes serialize kernels)
        L.set_flags(int(os.environ.get("GB_VARIANT", 10)), True)
        pmc_shapes = [sh for sh in shapes if sh[0] in os.environ["GB_SHAPES"].split(",")] if os.environ.get("GB_SHAPES") else shapes[:2]
        for name, N, K1, K2, grp in pmc_shapes:
            a1 = torch.randn(M, K1, device=dev).to(BF)
            b1 = (torch.randn(N, K1, device=dev) * 0.02).to(BF)
            out = torch.empty(M, N, dtype=BF, device=dev)
            G = N // grp if grp else 1
            kw = dict(a2=torch.randn(M, G * K2, device=dev).to(BF), b2=(torch.randn(N, K2, device=dev) * 0.02).to(BF),
                      a2_group_n=grp, a2_group_stride=K2 if grp else 0)
            for _ in range(3):
                L.gemm_nt(a1, b1, out, **kw)
        torch.cuda.synchronize()
        return
    vlist = [int(v) for v in os.environ["GB_VARIANTS"].split(",")] if (only == "gemm" and os.environ.get("GB_VARIANTS")) else None
    # variant 10 = the DEFAULT dispatch (what ships: streaming w4s where the launcher chooses it) is the FIRST row of every campaign, so the per-shape
    # vs-vendor table measures the shipped kernel choice; 31 / 17 / 4 force one kernel (31 also disables streaming)
    if not os.environ.get("GB_NO_WARM"):      # the first timed row must not also be the process's first kernels (r06n: qkv read 1370 TF/s there, 1489-1510 in tools/ab_stream.py)
        wa, wb = torch.randn(M, 4096, device=dev).to(BF), (torch.randn(4096, 4096, device=dev) * 0.02).to(BF)
        wo = torch.empty(M, 4096, dtype=BF, device=dev)
        for _ in range(400):
            L.gemm_nt(wa, wb, wo)
        torch.cuda.synchronize()
        del wa, wb, wo
    for glds in (vlist if vlist else ((10, 31, 17, 4) if not only else ((10, 31, 17, 4) if only == "gemm" else ((17, 31) if only == "pp" else ())))):
        L.set_flags(glds, True)
        for name, N, K1, K2, grp in shapes:
            a1 = torch.randn(M, K1, device=dev).to(BF)
            b1 = (torch.randn(N, K1, device=dev) * 0.02).to(BF)
            out = torch.empty(M, N, dtype=BF, device=dev)
            kw = {}
            if K2:
                G = N // grp if grp else 1
                kw = dict(a2=torch.randn(M, G * K2, device=dev).to(BF), b2=(torch.randn(N, K2, device=dev) * 0.02).to(BF),
                          a2_group_n=grp, a2_group_stride=K2 if grp else 0)
            t = timeit(lambda: L.gemm_nt(a1, b1, out, **kw))
            tf = 2.0 * M * N * (K1 + K2) / t / 1e12
            res.append(dict(kernel="gemm_nt", glds=glds, name=name, M=M, N=N, K=K1 + K2, ms=t * 1e3, tflops=tf))
            print(res[-1], flush=True)
    if only in ("", "yard"):      # diagnostic yardstick only (never used by the product): vendor GEMM at the same shapes
        for name, N, K1, K2, grp in shapes:
            a1 = torch.randn(M, K1 + K2, device=dev).to(BF)
            b1 = (torch.randn(N, K1 + K2, device=dev) * 0.02).to(BF)
            t = timeit(lambda: torch.matmul(a1, b1.t()))
            res.append(dict(kernel="torch.matmul(hipBLASLt)", name=name, M=M, N=N, K=K1 + K2, ms=t * 1e3, tflops=2.0 * M * N * (K1 + K2) / t / 1e12))
            print(res[-1], flush=True)
    L.set_flags(True, True)
    for tr in ((1, 0) if not only else ((1,) if only == "gemm" else ())):
        L.set_flags(True, bool(tr))
        for name, N1, N2 in (("dB_qkv", 12288, 256), ("dA_qkv", 768, 4096), ("dB_d", 4096, 256), ("dA_d", 256, 11008)):
            p = torch.randn(M, N1, device=dev).to(BF)
            q = torch.randn(M, N2, device=dev).to(BF)
            c = torch.zeros(N1, N2, device=dev)
            t = timeit(lambda: L.gemm_tn(p, q, c))
            res.append(dict(kernel="gemm_tn", tr=tr, name=name, M=M, N1=N1, N2=N2, ms=t * 1e3, tflops=2.0 * M * N1 * N2 / t / 1e12))
            print(res[-1], flush=True)
    # attention
    S, Ln, nh, hd = 8, 1087, 32, 128
    H = nh * hd
    qkv = torch.randn(S * Ln, 3 * H, device=dev).to(BF)
    o = torch.empty(S * Ln, H, dtype=BF, device=dev)
    lse = torch.empty(S, nh, Ln, device=dev)
    for tr in (1, 0):
        L.set_flags(True, bool(tr))
        f = lambda: L.call("opadpo_attn_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H,
                           lse.data_ptr(), None, S, Ln, nh, hd, 1, hd ** -0.5, 0, 0, L.stream())
        t = timeit(f)
        fl = S * nh * 2 * Ln * Ln * hd * 2 / 2
        res.append(dict(kernel="attn_fwd", tr=tr, ms=t * 1e3, tflops=fl / t / 1e12))
        print(res[-1], flush=True)
        dq = torch.zeros(S * Ln, H, device=dev)
        dqkv = torch.empty(S * Ln, 3 * H, dtype=BF, device=dev)
        delta = torch.empty(S, nh, Ln, device=dev)
        do = torch.randn(S * Ln, H, device=dev).to(BF)
        f = lambda: L.call("opadpo_attn_bwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(),
                           do.data_ptr(), H, lse.data_ptr(), None, dqkv.data_ptr(), dqkv.data_ptr() + 2 * H, dqkv.data_ptr() + 4 * H,
                           None, delta.data_ptr(), S, Ln, nh, hd, 1, hd ** -0.5, 0, 0, L.stream())
        t = timeit(f)
        res.append(dict(kernel="attn_bwd", tr=tr, ms=t * 1e3, tflops=2.5 * fl / t / 1e12))
        print(res[-1], flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(REPO, "gpurun_out", "gemm_bench.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
