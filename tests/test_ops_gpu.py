"""GPU: every kernel of libopadpo_hip.so, called through the C ABI, against a torch fp32 reference of
the same op (floating-point kernels -> tolerance stated per test).  Variant switches (global_load_lds
staging, ds_read_b64_tr_b16 transposed reads) are exercised both ways."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def L():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from opadpo_amd import lib
    lib.load()
    yield lib
    lib.set_flags(True, True)


def dev():
    return torch.device("cuda:0")


def rnd(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).to(dev())


def relerr(got, want):
    got, want = got.float(), want.float()
    return float((got - want).norm() / (want.norm() + 1e-12))


def maxabs(got, want):
    return float((got.float() - want.float()).abs().max())


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("glds", [31, 17, 10, 4])
@pytest.mark.parametrize("M,N,K1,K2,groups", [(300, 256, 128, 0, 0), (1, 128, 64, 0, 0), (129, 384, 192, 128, 3),
                                              (1000, 1024, 512, 128, 2), (257, 128, 64, 64, 1), (515, 768, 64, 64, 3),
                                              (2, 256, 4096, 0, 0), (131, 512, 64 * 3, 64, 2), (700, 512, 64, 0, 0), (513, 256, 128, 128, 1)])
def test_gemm_nt(L, glds, M, N, K1, K2, groups):
    L.set_flags(glds, True)
    a1 = rnd(M, K1, seed=1)
    b1 = rnd(N, K1, seed=2)
    out = torch.empty(M, N, dtype=BF, device=dev())
    want = a1.float() @ b1.float().t()
    kw = {}
    if K2:
        G = max(groups, 1)
        a2 = rnd(M, G * K2, seed=3)
        b2 = rnd(N, K2, seed=4)
        kw = dict(a2=a2, b2=b2, a2_group_n=(N // G if groups > 1 else 0), a2_group_stride=(K2 if groups > 1 else 0))
        if groups > 1:
            ng = N // G
            for g in range(G):
                want[:, g * ng:(g + 1) * ng] += a2[:, g * K2:(g + 1) * K2].float() @ b2[g * ng:(g + 1) * ng].float().t()
        else:
            want += a2[:, :K2].float() @ b2.float().t()
    L.gemm_nt(a1, b1, out, **kw)
    torch.cuda.synchronize()
    e = relerr(out, want)
    assert e < 6e-3, f"gemm_nt rel err {e}"   # bf16 output rounding ~ 2^-9


@pytest.mark.parametrize("M,tr", [(1, 1), (3, 1), (8, 1), (9, 1), (16, 1), (17, 1), (24, 1), (25, 1), (32, 1), (40, 1), (64, 1), (1, 17), (8, 17),
                                  (16, 17), (24, 17)])
@pytest.mark.parametrize("N,K1,K2,groups", [(256, 4096, 0, 0), (384, 192, 128, 3), (16384, 128, 64, 2), (128, 2752, 256, 1),
                                            (22016, 64, 128, 2)])
def test_gemm_nt_skinny(L, M, tr, N, K1, K2, groups):
    """decode-sized GEMMs (M <= 64) take the weight-streaming kernels: in-workgroup split-K, LoRA tail as a second K
    segment, grouped tail columns, fused epilogue (alpha / bias / residual, bf16 and fp32 outputs).  M <= 16: the whole-cache-line
    form (two 8x8x32 products per MFMA); tr = 17 (bit 4) switches it off so the 16-row form is covered at those sizes too."""
    L.set_flags(15, tr)            # 15 = force the streaming kernel (product: OPADPO_GEMM_STREAM hint per call)
    a1, b1 = rnd(M, K1, scale=0.5, seed=1), rnd(N, K1, scale=0.5, seed=2)
    want = a1.float() @ b1.float().t()
    kw = {}
    if K2:
        G = max(groups, 1)
        a2, b2 = rnd(M, G * K2, scale=0.5, seed=3), rnd(N, K2, scale=0.5, seed=4)
        kw = dict(a2=a2, b2=b2, a2_group_n=(N // G if groups > 1 else 0), a2_group_stride=(K2 if groups > 1 else 0))
        ng = N // G
        for g in range(G):
            want[:, g * ng:(g + 1) * ng] += a2[:, g * K2:(g + 1) * K2].float() @ b2[g * ng:(g + 1) * ng].float().t()
    out = torch.full((M + 1, N), 7.0, dtype=BF, device=dev())
    L.gemm_nt(a1, b1, out[:M], **kw)
    assert relerr(out[:M], want) < 6e-3
    assert float((out[M:].float() - 7.0).abs().max()) == 0.0          # rows >= M untouched
    bias, res32 = rnd(N, seed=5), torch.randn(M, N, device=dev())
    o32 = torch.empty(M, N, device=dev())
    L.gemm_nt(a1, b1, o32, bias=bias, residual=res32, alpha=0.25, **kw)
    assert relerr(o32, 0.25 * want + bias.float() + res32) < 1e-5
    resb = rnd(M, N, seed=6)
    ob = torch.empty(M, N, dtype=BF, device=dev())
    L.gemm_nt(a1, b1, ob, residual=resb, act=2, **kw)
    assert relerr(ob, torch.nn.functional.gelu(want) + resb.float()) < 6e-3
    # the general tile kernel on the same problem agrees
    L.set_flags(4, True)
    o4 = torch.empty(M, N, device=dev())
    L.gemm_nt(a1, b1, o4, bias=bias, residual=res32, alpha=0.25, **kw)
    L.set_flags(10, tr)
    assert relerr(o32, o4) < 1e-5
    with L.decode_schedule():      # the per-call hint selects the same kernel: identical bits
        o5 = torch.empty(M, N, device=dev())
        L.gemm_nt(a1, b1, o5, bias=bias, residual=res32, alpha=0.25, **kw)
    L.set_flags(10, True)
    assert torch.equal(o5, o32)


@pytest.mark.parametrize("K", [4096, 11008, 1024])
def test_gemm_nt_r_wide_products_k_folded(L, K):
    """Round 6: the N = lora_r = 256 products of the LoRA path (t = s x A^T, dT = s dY B; peft lora_A / lora_B forward and autograd, rl_models.py:120) run
    K-FOLDED on the 256x256 kernel: two K halves as two column groups of one launch, then out = bf16(alpha (P0 + P1)).  Against torch fp32; against the
    one-pass 128x128 kernel (variant 4: another fp32 association, same bf16 result up to an ulp); the rule depends on N and K only, so a row's bits do not
    depend on the rows around it (a 700-row batch == its first 300 rows run alone == its last row run alone), strided operands, alpha."""
    N, alpha = 256, 2.0
    A = rnd(700, K + 64, scale=0.5, seed=1); W = rnd(N, K + 32, scale=0.05, seed=2)
    a, w = A[:, 32:32 + K], W[:, 16:16 + K]
    want = alpha * (a.float() @ w.float().t())
    outs = {}
    for variant in (10, 31, 4):
        L.set_flags(variant, True)
        O = torch.full((701, N + 16), 7.0, dtype=BF, device=dev())
        L.gemm_nt(a, w, O[:700, 8:8 + N], alpha=alpha)
        torch.cuda.synchronize()
        outs[variant] = O[:700, 8:8 + N].clone()
        assert relerr(outs[variant], want) < 6e-3
        O[:700, 8:8 + N] = 7.0
        assert float((O.float() - 7.0).abs().max()) == 0.0               # nothing outside the window written
    assert torch.equal(outs[10], outs[31])
    d4 = (outs[10].float() - outs[4].float()).abs()
    assert float(d4.max()) <= 2.0 ** -7 * float(want.abs().max()) and float((d4 > 0).float().mean()) < 0.2      # the fold re-associates fp32 sums: rare one-ulp flips
    L.set_flags(10, True)
    o300 = torch.empty(300, N, dtype=BF, device=dev()); o1 = torch.empty(1, N, dtype=BF, device=dev()); again = torch.empty(700, N, dtype=BF, device=dev())
    L.gemm_nt(a[:300], w, o300, alpha=alpha); L.gemm_nt(a[699:], w, o1, alpha=alpha); L.gemm_nt(a, w, again, alpha=alpha)
    assert torch.equal(o300, outs[10][:300]) and torch.equal(o1, outs[10][699:]) and torch.equal(again, outs[10])


@pytest.mark.parametrize("N,K1,K2", [(4096, 11008, 256), (4096, 8192, 0), (512, 22016, 0), (8192, 8448, 64)])
def test_gemm_nt_deep_k_text_is_bit_identical(L, N, K1, K2):
    """Round 6: products of >= 128 K-tiles (down projection K = 11008 + r, the dgrads K = 12288 / 22016; LlamaMLP / LlamaAttention backward under peft,
    rl_models.py:120) run the DEEP text of the generated K-loop (csrc/w4_kloop_gen.py: the same loads, barriers and MFMAs placed differently).  The MFMA order
    is the same, so the result must be BIT-equal to the default text (flag bit 11) - on both piece orders (N <= 4096: B first, beyond: A first), with and
    without the K-concatenated LoRA tail, ragged M, fp32 output with a residual - and equal to torch fp32 within bf16 rounding."""
    M = 4300 + 37                                    # 17 row tiles: more than one full round of 256 tiles on every N but 512 (where the deals coincide)
    a1, b1 = rnd(M, K1, scale=0.5, seed=1), rnd(N, K1, scale=0.03, seed=2)
    kw = dict(a2=rnd(M, K2, scale=0.5, seed=3), b2=rnd(N, K2, scale=0.03, seed=4)) if K2 else {}
    want = a1.float() @ b1.float().t() + (kw["a2"].float() @ kw["b2"].float().t() if K2 else 0.0)
    res = torch.randn(M, N, device=dev())
    got = {}
    # variant 31: one tile per workgroup; variant 10 with bit 10: the STREAMING kernel on 8 workgroups (its DEEP text when bit 11 is clear)
    # bit 12: the tile order dealt to the XCDs in contiguous chunks (rounds 1-5) instead of block-cyclically - the same tiles in another order
    for variant, bits in ((31, 0), (31, 2048), (10, 1024), (10, 1024 | 2048), (31, 4096), (10, 1024 | 4096)):
        L.set_flags(variant, 1 | bits)
        ob = torch.empty(M, N, dtype=BF, device=dev()); of = torch.empty(M, N, device=dev())
        L.gemm_nt(a1, b1, ob, **kw)
        L.gemm_nt(a1, b1, of, residual=res, **kw)
        torch.cuda.synchronize()
        got[(variant, bits)] = (ob, of)
        assert relerr(ob, want) < 6e-3 and relerr(of, want + res) < 2e-5
    L.set_flags(True, True)
    for key, (ob, of) in got.items():
        assert torch.equal(ob, got[(31, 2048)][0]) and torch.equal(of, got[(31, 2048)][1]), key


@pytest.mark.parametrize("M", [2, 8, 13, 24, 31, 50])
def test_gemm_nt_decode_strided_operands(L, M):
    """Decode-schedule kernels with every operand a column slice of a wider buffer (lda / ldb / ldc / ldr != logical width),
    LoRA tail included: the streaming kernels address rows through the leading dimensions only."""
    L.set_flags(10, True)
    N, K1, K2 = 640, 320, 64
    A = rnd(M, K1 + 96, scale=0.5, seed=1); W = rnd(N, K1 + 32, scale=0.2, seed=2)
    A2 = rnd(M, K2 + 64, scale=0.5, seed=3); W2 = rnd(N, K2 + 128, scale=0.2, seed=4)
    a1, b1, a2, b2 = A[:, 32:32 + K1], W[:, 8:8 + K1], A2[:, 64:], W2[:, 64:64 + K2]
    Rf = torch.randn(M, N + 16, device=dev()); res = Rf[:, 8:8 + N]
    O = torch.full((M + 1, N + 24), 5.0, device=dev()); out = O[:M, 16:16 + N]
    want = a1.float() @ b1.float().t() + a2.float() @ b2.float().t() + res
    with L.decode_schedule():
        L.gemm_nt(a1, b1, out, a2=a2, b2=b2, residual=res)
    torch.cuda.synchronize()
    assert relerr(out, want) < 2e-5
    O2 = O.clone(); O2[:M, 16:16 + N] = 5.0
    assert float((O2 - 5.0).abs().max()) == 0.0          # nothing outside the [M, N] window written
    F = 256
    Wg = rnd(2 * F, K1 + 64, scale=0.2, seed=7); wsw = Wg[:, 16:16 + K1]
    Ob = torch.full((M + 1, F + 8), 3.0, dtype=BF, device=dev()); ob = Ob[:M, 8:]
    with L.decode_schedule():
        L.gemm_nt(a1, wsw, ob, act=L.ACT_SWIGLU_PAIR)
    torch.cuda.synchronize()
    z = a1.float() @ wsw.float().t()
    z = z.view(M, F // 64, 2, 64)
    assert relerr(ob, (torch.nn.functional.silu(z[:, :, 0]) * z[:, :, 1]).reshape(M, F)) < 2e-2
    assert float((Ob[M].float() - 3.0).abs().max()) == 0.0 and float((Ob[:M, :8].float() - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize("variant", [10, 31, 17])
def test_gemm_nt_epilogue_large(L, variant):
    """Epilogues at a size the auto dispatch sends to the 256x256 kernels (>= 320 blocks, ragged M edge): the 4-wave kernel's
    row-contiguous staged paths (bf16 one pass; fp32 two passes with fp32 / bf16 residual, alpha != 1) and the routing of
    bias / activation problems to the 8-wave kernel."""
    L.set_flags(variant, True)
    M, N, K = 5000, 4096, 192
    a, b = rnd(M, K, scale=0.5, seed=1), rnd(N, K, scale=0.5, seed=2)
    want = a.float() @ b.float().t()
    bias, res32, resb = rnd(N, seed=3), torch.randn(M, N, device=dev()), rnd(M, N, seed=4)
    ob = torch.full((M + 3, N), 7.0, dtype=BF, device=dev())
    L.gemm_nt(a, b, ob[:M])
    assert relerr(ob[:M], want) < 6e-3 and float((ob[M:].float() - 7.0).abs().max()) == 0.0
    L.gemm_nt(a, b, ob[:M], residual=resb, alpha=0.5)
    assert relerr(ob[:M], 0.5 * want + resb.float()) < 6e-3
    o32 = torch.empty(M, N, device=dev())
    L.gemm_nt(a, b, o32, residual=res32)
    assert relerr(o32, want + res32) < 1e-5
    L.gemm_nt(a, b, o32, residual=resb, alpha=0.25)
    assert relerr(o32, 0.25 * want + resb.float()) < 1e-5
    L.gemm_nt(a, b, o32, bias=bias, residual=res32, alpha=0.25, act=1)
    v = 0.25 * want + bias.float()
    assert relerr(o32, v * torch.sigmoid(1.702 * v) + res32) < 1e-5
    L.gemm_nt(a, b, ob[:M], bias=bias, act=2)
    assert relerr(ob[:M], torch.nn.functional.gelu(want + bias.float())) < 6e-3
    L.set_flags(10, True)


@pytest.mark.parametrize("M,N,K", [(5000, 4096, 192), (12694, 1024, 1024), (10000, 1024, 256)])
def test_gemm_nt_bias_activation_on_the_4wave_kernel(L, M, N, K):
    """Round 5: bias / quick-GELU / GELU problems (+ an fp32 residual into an fp32 result) - the vision tower's and the projector's GEMMs - run on the BA
    instantiations of the 4-wave 256x256 kernel (direct epilogue: alpha, the bias of the lane's 8 columns, activation, residual - epilogue4's order).
    Default dispatch (variant 10) and the 256x256 kernel on every tile (31) == the 8-wave kernel (17) == the 128x128 kernel (4), bit for bit; rows >= M untouched."""
    a, b = rnd(M, K, scale=0.5, seed=1), rnd(N, K, scale=0.1, seed=2)
    bias, res32 = rnd(N, seed=3), torch.randn(M, N, device=dev())
    outs = {}
    try:
        for v in (10, 31, 17, 4):
            L.set_flags(v, True)
            o1 = torch.full((M + 2, N), 7.0, dtype=BF, device=dev())
            L.gemm_nt(a, b, o1[:M], bias=bias, act=1)
            o2 = torch.full((M + 2, N), 7.0, dtype=torch.float32, device=dev())
            L.gemm_nt(a, b, o2[:M], bias=bias, residual=res32, alpha=0.5)
            o3 = torch.full((M + 2, N), 7.0, dtype=BF, device=dev())
            L.gemm_nt(a, b, o3[:M], bias=bias, act=2)
            torch.cuda.synchronize()
            outs[v] = (o1, o2, o3)
    finally:
        L.set_flags(10, True)
    for v in (31, 17, 4):
        for k in range(3):
            assert torch.equal(outs[10][k], outs[v][k]), f"variant 10 != variant {v} (output {k})"
    want = a.float() @ b.float().t()
    x1 = want + bias.float()
    assert relerr(outs[10][0][:M], x1 * torch.sigmoid(1.702 * x1)) < 6e-3
    assert relerr(outs[10][1][:M], 0.5 * want + bias.float() + res32) < 1e-5
    assert relerr(outs[10][2][:M], torch.nn.functional.gelu(x1)) < 6e-3
    for k in range(3):
        assert float((outs[10][k][M:].float() - 7.0).abs().max()) == 0.0


def test_gemm_nt_256_kernels_race_screen(L):
    """The two 256x256 kernels (4-wave long-lead w4 = 31 / default for large GEMMs, 8-wave 4-phase p8 = 17) accumulate every
    output in the same k order, so their results must be BIT-identical; repeated on a shape with many K-tiles, a LoRA tail, a ragged M
    edge and more blocks than CUs, any LDS-DMA / barrier race in a schedule shows up as a differing tile."""
    M, N, K1, K2 = 3000, 5120, 4096, 256
    a1, b1 = rnd(M, K1, seed=11), rnd(N, K1, scale=0.05, seed=12)
    a2, b2 = rnd(M, K2, seed=13), rnd(N, K2, scale=0.05, seed=14)
    outs = {}
    for rep in range(6):
        for v in (17, 31, 10):
            L.set_flags(v, True)
            o = torch.empty(M, N, dtype=BF, device=dev())
            L.gemm_nt(a1, b1, o, a2=a2, b2=b2)
            key = v
            if key in outs:
                assert torch.equal(outs[key], o), f"variant {v} not reproducible (run {rep})"
            outs[key] = o
    L.set_flags(10, True)
    torch.cuda.synchronize()
    assert torch.equal(outs[31], outs[17]) and torch.equal(outs[17], outs[10])
    want = a1.float() @ b1.float().t() + a2.float() @ b2.float().t()
    assert relerr(outs[17], want) < 6e-3


@pytest.mark.parametrize("M,N,K1,K2,f32,alpha", [(1100, 1024, 192, 0, False, 1.0),      # 5 x 4 tiles on 8 workgroups: walks of 2-3 tiles, K1 = 3 K-tiles (FIRST + the two hand-over tiles only)
                                                 (1100, 1024, 256, 64, True, 1.0),       # + one K-tile of the LoRA tail, fp32 out (64 stores per wave behind the next tile's pieces)
                                                 (2049, 768, 512, 256, False, 0.5),      # 9 x 3 = 27 tiles (3.4 per workgroup), ragged last row tile, alpha epilogue, 4 tail K-tiles
                                                 (4096, 2048, 1024, 0, False, 1.0),      # 128 tiles = 16 per workgroup, N <= 16 column tiles -> B pieces first
                                                 (700, 5120, 320, 128, True, 1.0)])      # 3 x 20 tiles, N > 16 column tiles -> A pieces first, K1 = 5
def test_gemm_nt_streaming_kernel_equals_one_tile_per_workgroup(L, M, N, K1, K2, f32, alpha):
    """gemm_nt_w4s_kernel (round 5: one workgroup walks many output tiles with the K-tile pipeline kept full across them) against the
    one-tile-per-workgroup kernel (variant 31) and the 128x128 kernel (variant 4): same k order -> BIT-identical, for every way a walk
    can go - first tile / middle tiles / last tile of a workgroup, with and without the second operand pair, the shortest K that streams
    (3 K-tiles), fp32 and bf16 stores, workgroups with different tile counts, both piece orders.  opadpo_set_flags use_tr bit 10 runs the walk on
    8 workgroups so that these small problems stream at all (>= 2 tiles per workgroup); repeated to catch a schedule race."""
    lib = L.load()
    a1, b1 = rnd(M, K1, seed=31), rnd(N, K1, scale=0.05, seed=32)
    a2, b2 = (rnd(M, K2, seed=33), rnd(N, K2, scale=0.05, seed=34)) if K2 else (None, None)
    dt = torch.float32 if f32 else BF
    outs = {}
    try:
        for rep in range(3):
            for key, variant, flags in (("stream8", 10, 1 | 1024), ("one_tile", 31, 1), ("k128", 4, 1)):
                lib.opadpo_set_flags(variant, flags)
                o = torch.full((M, N), 7.0, dtype=dt, device=dev())
                L.gemm_nt(a1, b1, o, a2=a2, b2=b2, alpha=alpha)
                torch.cuda.synchronize()
                if key in outs:
                    assert torch.equal(outs[key], o), f"{key} not reproducible (run {rep})"
                outs[key] = o
    finally:
        lib.opadpo_set_flags(10, 1)
    assert torch.equal(outs["stream8"], outs["one_tile"]) and torch.equal(outs["one_tile"], outs["k128"])
    want = alpha * (a1.float() @ b1.float().t() + (a2.float() @ b2.float().t() if K2 else 0.0))
    assert relerr(outs["stream8"].float(), want) < (2e-5 if f32 else 6e-3)


@pytest.mark.parametrize("M,N,K1,K2", [(1100, 1024, 256, 64), (2500, 512, 192, 0), (4096, 2048, 1024, 256)])
def test_gemm_nt_direct_residual_and_swiglu_bwd_epilogues_stream(L, M, N, K1, K2):
    """Round 5: the fp32 residual operand and the SwiGLU backward leave the 256x256 kernels through their DIRECT epilogue (operands requested one row
    block ahead of the accumulator read-out, no LDS), which lets those products stream too.  fp32 residual: streaming kernel (8-workgroup test
    walk) == one tile per workgroup == 128x128 kernel, bit for bit, and == the plain fp32 product + the residual added by torch (one fp32 add of the
    same operands: what the RMSNorm pass did before).  SwiGLU backward: streaming == one tile per workgroup == plain product + opadpo_silu_mul_bwd."""
    lib = L.load()
    a1, b1 = rnd(M, K1, seed=41), rnd(N, K1, scale=0.05, seed=42)
    a2, b2 = (rnd(M, K2, seed=43), rnd(N, K2, scale=0.05, seed=44)) if K2 else (None, None)
    res = rnd(M, N, seed=45).float() * 1.001
    gu = rnd(M, 2 * N, seed=46)
    outs = {}
    try:
        for rep in range(2):
            for key, variant, flags in (("stream8", 10, 1 | 1024), ("one_tile", 31, 1), ("k128", 4, 1)):
                lib.opadpo_set_flags(variant, flags)
                o = torch.full((M + 2, N), 7.0, dtype=torch.float32, device=dev())
                L.gemm_nt(a1, b1, o[:M], a2=a2, b2=b2, residual=res)
                g = None
                if key != "k128":          # the SwiGLU backward epilogue exists in the 256x256 kernels only
                    g = torch.full((M + 2, 2 * N), 7.0, dtype=BF, device=dev())
                    L.gemm_nt(a1, b1, g[:M], a2=a2, b2=b2, residual=gu, act=L.ACT_SWIGLU_BWD)
                torch.cuda.synchronize()
                if key in outs:
                    assert torch.equal(outs[key][0], o) and (g is None or torch.equal(outs[key][1], g)), f"{key} not reproducible (run {rep})"
                outs[key] = (o, g)
        lib.opadpo_set_flags(10, 1)
        plain = torch.empty(M, N, dtype=torch.float32, device=dev())
        L.gemm_nt(a1, b1, plain, a2=a2, b2=b2)
        dact = torch.empty(M, N, dtype=BF, device=dev())
        L.gemm_nt(a1, b1, dact, a2=a2, b2=b2)
        want_g = torch.empty(M, 2 * N, dtype=BF, device=dev())
        L.call("opadpo_silu_mul_bwd", L.ptr(dact), L.ptr(gu), L.ptr(want_g), M, N, L.stream())
        torch.cuda.synchronize()
    finally:
        lib.opadpo_set_flags(10, 1)
    assert torch.equal(outs["stream8"][0], outs["one_tile"][0]) and torch.equal(outs["one_tile"][0], outs["k128"][0])
    assert torch.equal(outs["stream8"][0][:M], plain + res), "residual in the epilogue != product + residual"
    assert float((outs["stream8"][0][M:] - 7.0).abs().max()) == 0.0
    assert torch.equal(outs["stream8"][1], outs["one_tile"][1])
    assert torch.equal(outs["stream8"][1][:M], want_g), "SwiGLU backward in the epilogue != projection + opadpo_silu_mul_bwd"
    assert float((outs["stream8"][1][M:].float() - 7.0).abs().max()) == 0.0


@pytest.mark.parametrize("variant", [4, 17, 31, 10])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_nt_epilogue(L, act, variant):
    L.set_flags(variant, True)
    M, N, K = 200, 256, 128
    a, b = rnd(M, K, scale=0.5, seed=1), rnd(N, K, scale=0.5, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    out = torch.empty(M, N, dtype=BF, device=dev())
    L.gemm_nt(a, b, out, bias=bias, residual=res, alpha=0.5, act=act)
    v = 0.5 * (a.float() @ b.float().t()) + bias.float()
    if act == 1:
        v = v * torch.sigmoid(1.702 * v)
    elif act == 2:
        v = torch.nn.functional.gelu(v)
    want = v + res.float()
    assert relerr(out, want) < 6e-3
    out32 = torch.empty(M, N, dtype=torch.float32, device=dev())
    L.gemm_nt(a, b, out32)
    assert relerr(out32, a.float() @ b.float().t()) < 1e-5
    res32 = torch.randn(M, N, device=dev())
    o32 = torch.empty(M, N, device=dev())
    L.gemm_nt(a, b, o32, residual=res32)
    assert relerr(o32, a.float() @ b.float().t() + res32) < 1e-5
    # strided views (leading dimension != width)
    big = rnd(M, 3 * K, seed=5)
    outw = torch.zeros(M, 2 * N, dtype=BF, device=dev())
    L.gemm_nt(big[:, K:2 * K], b, outw[:, N:])
    assert relerr(outw[:, N:], big[:, K:2 * K].float() @ b.float().t()) < 6e-3
    assert float(outw[:, :N].abs().max()) == 0.0


@pytest.mark.parametrize("variant", [4, 17, 31])
def test_gemm_nt_grouped_a1(L, variant):
    """block-diagonal dT_g = dY_g . B_g for fused projections in one launch (a1 groups)."""
    L.set_flags(variant, True)
    M, G, r, K = 600, 3, 256, 128
    a = rnd(M, G * K, seed=1)
    b = rnd(G * r, K, seed=2)
    out = torch.empty(M, G * r, dtype=BF, device=dev())
    L.gemm_nt(a, b, out, alpha=2.0, k1=K, a1_group_n=r, a1_group_stride=K)
    want = torch.cat([2.0 * a[:, g * K:(g + 1) * K].float() @ b[g * r:(g + 1) * r].float().t() for g in range(G)], 1)
    L.set_flags(True, True)
    assert relerr(out, want) < 6e-3


def test_gemm_nt_grouped_a1_partial_round_tail(L):
    """The benchmark's dT_qkv shape on ragged rows: N = 3r = 768 with one A1 column group per 256 outputs, 96 row tiles -> 288 tiles of
    256x256 = one full round + 32 tiles run as 8 K-slices each (the group offset and the K-slice offset of A1 must compose); also the
    deep-K one-round case (N = 256, K = 11008: 96 tiles as two slices each).  Against fp32 torch and the 128x128 kernel."""
    M, G, r, K = 96 * 256 - 77, 3, 256, 4096
    a = rnd(M, G * K, seed=1)
    b = rnd(G * r, K, scale=0.05, seed=2)
    out = torch.empty(M, G * r, dtype=BF, device=dev())
    small = torch.empty_like(out)
    L.set_flags(10, True)
    L.gemm_nt(a, b, out, alpha=2.0, k1=K, a1_group_n=r, a1_group_stride=K)
    L.set_flags(4, True)
    L.gemm_nt(a, b, small, alpha=2.0, k1=K, a1_group_n=r, a1_group_stride=K)
    L.set_flags(10, True)
    want = torch.cat([2.0 * a[:, g * K:(g + 1) * K].float() @ b[g * r:(g + 1) * r].float().t() for g in range(G)], 1)
    assert relerr(out, want) < 3e-3 and relerr(small, want) < 3e-3
    assert torch.equal(out, small), "quarter-tile tail with grouped A1 != 128x128 kernel"
    x, w = rnd(M, 11008, seed=3), rnd(256, 11008, scale=0.05, seed=4)
    t = torch.empty(M, 256, dtype=BF, device=dev())
    L.gemm_nt(x, w, t, alpha=0.5)                         # deep-K one-round problem: all quarter tiles
    assert relerr(t, 0.5 * (x.float() @ w.float().t())) < 3e-3
    L.set_flags(31, True)
    t2 = torch.empty_like(t)
    L.gemm_nt(x, w, t2, alpha=0.5)
    L.set_flags(10, True)
    assert torch.equal(t, t2)


@pytest.mark.parametrize("tr", [1, 9, 0, 13])      # 1 = default (256x256 stream-K kernel where both dims allow), 9 = 128x128 kernel, 13 = wide tiles
@pytest.mark.parametrize("M,N1,N2,groups", [(777, 256, 128, 0), (64, 128, 128, 0), (1500, 384, 128, 3), (130, 128, 256, 0),
                                            (1100, 512, 256, 0), (1100, 256, 1024, 0), (1300, 768, 256, 3), (2100, 1024, 256, 2),
                                            (999, 512, 384, 0), (70, 256, 256, 0), (20011, 2048, 256, 0), (9001, 512, 1024, 2)])
def test_gemm_tn(L, tr, M, N1, N2, groups):
    L.set_flags(True, tr)
    p = rnd(M, N1, seed=1)
    G = max(groups, 1)
    q = rnd(M, G * N2, seed=2)
    c0 = torch.randn(N1, N2, device=dev())
    c = c0.clone()
    if groups:
        L.gemm_tn(p, q, c, n2=N2, q_group_n1=N1 // G, q_group_stride=N2, alpha=2.0)
        want = c0.clone()
        ng = N1 // G
        for g in range(G):
            want[g * ng:(g + 1) * ng] += 2.0 * (p[:, g * ng:(g + 1) * ng].float().t() @ q[:, g * N2:(g + 1) * N2].float())
    else:
        L.gemm_tn(p, q, c, alpha=2.0)
        want = c0 + 2.0 * (p.float().t() @ q.float())
    e = relerr(c, want)
    assert e < 1e-4, f"gemm_tn rel err {e}"


def ref_attention(q, k, v, key_mask, causal, scale, seg=(0, 0)):
    """q,k,v [S,L,nh,hd] fp32 -> o [S,L,nh,hd], fully-masked rows -> 0.  seg = (prefix, seg_len): packed responses, a
    query attends the prefix and its own response segment only."""
    S, Ln, nh, hd = q.shape
    sc = torch.einsum("sqhd,skhd->shqk", q, k) * scale
    allow = torch.ones(S, 1, Ln, Ln, dtype=torch.bool, device=q.device)
    if causal:
        allow = allow & torch.tril(torch.ones(Ln, Ln, dtype=torch.bool, device=q.device))[None, None]
    if seg[1] > 0:
        pos = torch.arange(Ln, device=q.device)
        sid = torch.where(pos < seg[0], torch.full_like(pos, -1), (pos - seg[0]) // seg[1])
        same = (sid[None, :] == -1) | (sid[:, None] == sid[None, :])          # key in prefix, or same segment
        allow = allow & same[None, None]
    if key_mask is not None:
        allow = allow & key_mask.bool()[:, None, None, :]
    sc = sc.masked_fill(~allow, float("-inf"))
    p = torch.nan_to_num(torch.softmax(sc, -1), nan=0.0)
    return torch.einsum("shqk,skhd->sqhd", p, v)


@pytest.mark.parametrize("tr", [1, 0, 3])      # 3 = transposed reads + direct-to-LDS double-buffered forward
@pytest.mark.parametrize("S,Ln,nh,hd,causal,masked", [(2, 200, 2, 128, 1, True), (1, 64, 1, 128, 1, False),
                                                       (2, 77, 2, 64, 0, False), (1, 300, 1, 64, 1, True),
                                                       # several 128-row blocks / many K-V tiles of the 32-rows-per-wave kernel, ragged ends
                                                       (2, 517, 2, 128, 1, True), (1, 1087, 4, 128, 1, False), (2, 333, 2, 128, 0, True),
                                                       (1, 129, 1, 128, 1, True), (3, 31, 1, 128, 1, False)])
def test_attn_fwd(L, tr, S, Ln, nh, hd, causal, masked):
    L.set_flags(True, tr)
    H = nh * hd
    qkv = rnd(S * Ln, 3 * H, scale=1.0, seed=7)
    km = None
    if masked:
        km = torch.ones(S, Ln, dtype=torch.uint8, device=dev())
        km[0, :5] = 0
        km[-1, Ln - 9:] = 0
        km[0, 40:44] = 0
    o = torch.zeros(S * Ln, H, dtype=BF, device=dev())
    lse = torch.zeros(S, nh, Ln, device=dev())
    st = L.stream()
    L.call("opadpo_attn_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H,
           lse.data_ptr(), L.ptr(km), S, Ln, nh, hd, causal, hd ** -0.5, 0, 0, st)
    torch.cuda.synchronize()
    q4 = qkv[:, :H].float().view(S, Ln, nh, hd)
    k4 = qkv[:, H:2 * H].float().view(S, Ln, nh, hd)
    v4 = qkv[:, 2 * H:].float().view(S, Ln, nh, hd)
    want = ref_attention(q4, k4, v4, km, causal, hd ** -0.5).reshape(S * Ln, H)
    e = maxabs(o, want)
    assert e < 3e-2, f"attn_fwd max abs err {e} (|o| ~ {float(want.abs().max()):.2f})"
    assert relerr(o, want) < 1e-2
    # log-sum-exp (the backward's softmax statistics) against torch, on rows with at least one visible key
    sc = torch.einsum("sqhd,skhd->shqk", q4, k4) * hd ** -0.5
    allow = torch.ones(S, 1, Ln, Ln, dtype=torch.bool, device=dev())
    if causal:
        allow = allow & torch.tril(torch.ones(Ln, Ln, dtype=torch.bool, device=dev()))[None, None]
    if km is not None:
        allow = allow & km.bool()[:, None, None, :]
    want_lse = torch.logsumexp(sc.masked_fill(~allow, float("-inf")), -1)
    okr = allow.any(-1).expand(S, nh, Ln)
    assert float((lse - want_lse)[okr].abs().max()) < 2e-2


@pytest.mark.parametrize("tr", [1, 0])
@pytest.mark.parametrize("S,Ln,nh,hd,masked", [(2, 150, 2, 128, True), (1, 64, 1, 128, False), (1, 200, 2, 64, True), (2, 413, 2, 128, True)])
def test_attn_bwd(L, tr, S, Ln, nh, hd, masked):
    L.set_flags(True, bool(tr))
    H = nh * hd
    qkv = rnd(S * Ln, 3 * H, scale=0.7, seed=11)
    dout = rnd(S * Ln, H, scale=1.0, seed=12)
    km = None
    if masked:
        km = torch.ones(S, Ln, dtype=torch.uint8, device=dev())
        km[0, :7] = 0
        km[-1, Ln - 11:] = 0
    o = torch.zeros(S * Ln, H, dtype=BF, device=dev())
    lse = torch.zeros(S, nh, Ln, device=dev())
    st = L.stream()
    scale = hd ** -0.5
    L.call("opadpo_attn_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H,
           lse.data_ptr(), L.ptr(km), S, Ln, nh, hd, 1, scale, 0, 0, st)
    dq_acc = torch.zeros(S * Ln, H, device=dev())
    dqkv = torch.zeros(S * Ln, 3 * H, dtype=BF, device=dev())
    delta = torch.zeros(S, nh, Ln, device=dev())
    L.call("opadpo_attn_bwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(),
           dout.data_ptr(), H, lse.data_ptr(), L.ptr(km), dqkv.data_ptr(), dqkv.data_ptr() + 2 * H,
           dqkv.data_ptr() + 4 * H, dq_acc.data_ptr(), delta.data_ptr(), S, Ln, nh, hd, 1, scale, 0, 0, st)
    torch.cuda.synchronize()
    q4 = qkv[:, :H].float().view(S, Ln, nh, hd).requires_grad_(True)
    k4 = qkv[:, H:2 * H].float().view(S, Ln, nh, hd).requires_grad_(True)
    v4 = qkv[:, 2 * H:].float().view(S, Ln, nh, hd).requires_grad_(True)
    want = ref_attention(q4, k4, v4, km, 1, scale)
    want.backward(dout.float().view(S, Ln, nh, hd))
    assert torch.equal(dqkv[:, :H], dq_acc.to(BF))
    for name, got, ref in (("dq", dq_acc, q4.grad.reshape(S * Ln, H)), ("dk", dqkv[:, H:2 * H], k4.grad.reshape(S * Ln, H)),
                           ("dv", dqkv[:, 2 * H:], v4.grad.reshape(S * Ln, H))):
        e = relerr(got, ref)
        assert e < 2e-2, f"attn_bwd {name} rel err {e}"


@pytest.mark.parametrize("tr", [1, 0, 3])
@pytest.mark.parametrize("S,nh,hd,pfx,T,K", [(2, 2, 128, 70, 37, 2), (1, 1, 128, 130, 100, 3), (2, 2, 64, 64, 64, 2), (1, 2, 128, 5, 150, 2),
                                             (1, 1, 128, 200, 128, 3)])
def test_attn_packed_responses(L, tr, S, nh, hd, pfx, T, K):
    """seg_len > 0: rows are [prefix | response_0 | ... | response_{K-1}]; forward and backward equal (a) torch attention
    with the explicit segment mask and (b) K separate [prefix | response_k] sequences — what the reference runs."""
    L.set_flags(True, tr)
    H = nh * hd
    Ln = pfx + K * T
    scale = hd ** -0.5
    qkv = rnd(S * Ln, 3 * H, scale=0.8, seed=21)
    dout = rnd(S * Ln, H, seed=22)
    km = torch.ones(S, Ln, dtype=torch.uint8, device=dev())
    km[0, :3] = 0
    km[-1, pfx + T - 5: pfx + T] = 0                   # right padding of response 0
    st = L.stream()

    def run(x, do, mask, S_, L_, seg):
        o = torch.zeros(S_ * L_, H, dtype=BF, device=dev())
        lse = torch.zeros(S_, nh, L_, device=dev())
        L.call("opadpo_attn_fwd", x.data_ptr(), x.data_ptr() + 2 * H, x.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H,
               lse.data_ptr(), mask.data_ptr(), S_, L_, nh, hd, 1, scale, seg[0], seg[1], st)
        dx = torch.zeros(S_ * L_, 3 * H, dtype=BF, device=dev())
        dq32 = torch.zeros(S_ * L_, H, device=dev())
        delta = torch.zeros(S_, nh, L_, device=dev())
        L.call("opadpo_attn_bwd", x.data_ptr(), x.data_ptr() + 2 * H, x.data_ptr() + 4 * H, 3 * H, o.data_ptr(),
               do.data_ptr(), H, lse.data_ptr(), mask.data_ptr(), dx.data_ptr(), dx.data_ptr() + 2 * H, dx.data_ptr() + 4 * H,
               dq32.data_ptr(), delta.data_ptr(), S_, L_, nh, hd, 1, scale, seg[0], seg[1], st)
        torch.cuda.synchronize()
        return o, dx

    o, dx = run(qkv, dout, km, S, Ln, (pfx, T))
    q4 = qkv[:, :H].float().view(S, Ln, nh, hd).requires_grad_(True)
    k4 = qkv[:, H:2 * H].float().view(S, Ln, nh, hd).requires_grad_(True)
    v4 = qkv[:, 2 * H:].float().view(S, Ln, nh, hd).requires_grad_(True)
    want = ref_attention(q4, k4, v4, km, 1, scale, seg=(pfx, T))
    want.backward(dout.float().view(S, Ln, nh, hd))
    assert relerr(o, want.reshape(S * Ln, H)) < 1e-2
    for name, got, ref in (("dq", dx[:, :H], q4.grad), ("dk", dx[:, H:2 * H], k4.grad), ("dv", dx[:, 2 * H:], v4.grad)):
        e = relerr(got, ref.reshape(S * Ln, H))
        assert e < 2e-2, f"packed attn_bwd {name} rel err {e}"
    # (b) the K separate sequences [prefix | response_k] (kernel without segments): same response outputs; the prefix
    # gradients of the packed row are the sum over the K sequences
    x3 = qkv.view(S, Ln, 3 * H)
    d3 = dout.view(S, Ln, H)
    dk_pfx = torch.zeros(S, pfx, 2 * H, device=dev())
    for k in range(K):
        sl = slice(pfx + k * T, pfx + (k + 1) * T)
        xs = torch.cat([x3[:, :pfx], x3[:, sl]], 1).reshape(S * (pfx + T), 3 * H).contiguous()
        ds = torch.cat([torch.zeros_like(d3[:, :pfx]), d3[:, sl]], 1).reshape(S * (pfx + T), H).contiguous()
        ms = torch.cat([km[:, :pfx], km[:, sl]], 1).contiguous()
        os_, dxs = run(xs, ds, ms, S, pfx + T, (0, 0))
        assert relerr(o.view(S, Ln, H)[:, sl], os_.view(S, pfx + T, H)[:, pfx:]) < 4e-3
        assert relerr(dx.view(S, Ln, 3 * H)[:, sl], dxs.view(S, pfx + T, 3 * H)[:, pfx:]) < 1e-2
        dk_pfx += dxs.view(S, pfx + T, 3 * H)[:, :pfx, H:].float()
    # prefix queries get gradient only through dout on prefix rows (zeroed in the split runs), so compare dK/dV of the
    # prefix against the packed run with the prefix dout removed
    dout0 = dout.clone().view(S, Ln, H)
    dout0[:, :pfx] = 0
    _, dx0 = run(qkv, dout0.reshape(S * Ln, H).contiguous(), km, S, Ln, (pfx, T))
    assert relerr(dx0.view(S, Ln, 3 * H)[:, :pfx, H:], dk_pfx) < 1.5e-2


def test_rope_packed_positions(L):
    S, nh, hd, pfx, T, K = 2, 2, 128, 11, 7, 3
    Ln, H = pfx + K * T, nh * hd
    qkv = rnd(S * Ln, 3 * H, seed=1)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    f = torch.outer(torch.arange(Ln).float(), inv)
    cos, sin = f.cos().to(dev()).contiguous(), f.sin().to(dev()).contiguous()
    x = qkv.clone()
    L.call("opadpo_rope", x.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), S * Ln, Ln, 2 * nh, hd, 0, None, pfx, T, L.stream())
    # reference: the plain kernel on each [prefix | response_k] sequence
    x3, q3 = x.view(S, Ln, 3 * H), qkv.view(S, Ln, 3 * H)
    for k in range(K):
        sl = slice(pfx + k * T, pfx + (k + 1) * T)
        xs = torch.cat([q3[:, :pfx], q3[:, sl]], 1).reshape(S * (pfx + T), 3 * H).contiguous()
        L.call("opadpo_rope", xs.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), S * (pfx + T), pfx + T, 2 * nh, hd, 0, None, 0, 0, L.stream())
        xs = xs.view(S, pfx + T, 3 * H)
        assert torch.equal(x3[:, sl], xs[:, pfx:]) and torch.equal(x3[:, :pfx], xs[:, :pfx])
    L.call("opadpo_rope", x.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), S * Ln, Ln, 2 * nh, hd, 1, None, pfx, T, L.stream())
    assert relerr(x, qkv) < 8e-3


def test_scatter_add_rows_f32(L):
    n, H, R = 9, 256, 20
    src = torch.randn(n, H, device=dev())
    idx = torch.tensor([3, 5, 3, 0, 19, 5, 3, 7, 8], dtype=torch.int32, device=dev())     # duplicates accumulate
    dst = torch.randn(R, H, device=dev())
    want = dst.clone().index_add_(0, idx.long(), src)
    L.call("opadpo_scatter_add_rows_f32", src.data_ptr(), idx.data_ptr(), dst.data_ptr(), H, n, H, L.stream())
    torch.cuda.synchronize()
    assert float((dst - want).abs().max()) < 1e-5


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("xf32", [0, 1])
def test_rmsnorm(L, xf32):
    rows, H = 37, 512
    xb, w = rnd(rows, H, seed=1), (1 + 0.1 * torch.randn(H)).to(BF).to(dev())
    x = xb.float().contiguous() if xf32 else xb
    y = torch.empty(rows, H, dtype=BF, device=dev())
    rstd = torch.empty(rows, device=dev())
    L.call("opadpo_rmsnorm_fwd", x.data_ptr(), xf32, w.data_ptr(), y.data_ptr(), rstd.data_ptr(), rows, H, 1e-5, L.stream())
    xf = xb.float().requires_grad_(True)
    r = torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-5)
    want = xf * r * w.float()
    assert relerr(y, want) < 4e-3
    assert relerr(rstd, r.squeeze(-1)) < 1e-5
    dy, dresb = rnd(rows, H, seed=2), rnd(rows, H, seed=3)
    dres = dresb.float().contiguous() if xf32 else dresb
    dx32 = torch.empty(rows, H, device=dev())
    dx16 = torch.empty(rows, H, dtype=BF, device=dev())
    L.call("opadpo_rmsnorm_bwd", dy.data_ptr(), x.data_ptr(), xf32, w.data_ptr(), rstd.data_ptr(), dres.data_ptr(), xf32,
           dx32.data_ptr(), dx16.data_ptr(), rows, H, L.stream())
    want.backward(dy.float())
    assert relerr(dx32, xf.grad + dresb.float()) < 1e-5
    assert torch.equal(dx16, dx32.to(BF))


def test_layernorm_bwd_and_act(L):
    """CLIP training-path pieces (OPA LoRA-SFT stage): LayerNorm backward w.r.t. x (+ residual-path gradient), activation fwd/bwd
    on a stored pre-activation."""
    rows, H = 41, 1024
    x, w, b = rnd(rows, H, seed=1), (1 + 0.1 * torch.randn(H)).to(BF).to(dev()), rnd(H, scale=0.1, seed=3)
    dy, dres = rnd(rows, H, seed=4), rnd(rows, H, seed=5)
    dx = torch.empty_like(x)
    L.call("opadpo_layernorm_bwd", dy.data_ptr(), x.data_ptr(), w.data_ptr(), dres.data_ptr(), dx.data_ptr(), rows, H, 1e-5, L.stream())
    xf = x.float().requires_grad_(True)
    y = torch.nn.functional.layer_norm(xf, (H,), w.float(), b.float(), 1e-5)
    y.backward(dy.float())
    assert relerr(dx, xf.grad + dres.float()) < 6e-3
    dx0 = torch.empty_like(x)
    L.call("opadpo_layernorm_bwd", dy.data_ptr(), x.data_ptr(), w.data_ptr(), None, dx0.data_ptr(), rows, H, 1e-5, L.stream())
    assert relerr(dx0, xf.grad) < 6e-3
    for act, fn in ((1, lambda z: z * torch.sigmoid(1.702 * z)), (2, torch.nn.functional.gelu)):
        z = rnd(37, 512, scale=1.5, seed=6 + act)
        out, dz, do = torch.empty_like(z), torch.empty_like(z), rnd(37, 512, seed=9)
        L.call("opadpo_act_fwd", z.data_ptr(), out.data_ptr(), z.numel(), act, L.stream())
        L.call("opadpo_act_bwd", do.data_ptr(), z.data_ptr(), dz.data_ptr(), z.numel(), act, L.stream())
        zf = z.float().requires_grad_(True)
        ref = fn(zf)
        ref.backward(do.float())
        assert relerr(out, ref) < 4e-3 and relerr(dz, zf.grad) < 6e-3


def test_layernorm(L):
    rows, H = 19, 256
    x, w, b = rnd(rows, H, seed=1), rnd(H, seed=2), rnd(H, seed=3)
    y = torch.empty_like(x)
    L.call("opadpo_layernorm_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), rows, H, 1e-5, L.stream())
    want = torch.nn.functional.layer_norm(x.float(), (H,), w.float(), b.float(), 1e-5)
    assert relerr(y, want) < 4e-3


def test_rope(L):
    S, Ln, nh, hd = 2, 33, 2, 128
    H = nh * hd
    qkv = rnd(S * Ln, 3 * H, seed=1)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    f = torch.outer(torch.arange(Ln).float(), inv)
    cos, sin = f.cos().to(dev()).contiguous(), f.sin().to(dev()).contiguous()
    x = qkv.clone()
    L.call("opadpo_rope", x.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), S * Ln, Ln, 2 * nh, hd, 0, None, 0, 0, L.stream())
    ref = qkv.float().view(S, Ln, 3 * nh, hd)
    c = torch.cat([f, f], -1).cos().to(dev())[None, :, None, :]
    s = torch.cat([f, f], -1).sin().to(dev())[None, :, None, :]
    rot = torch.cat([-ref[..., hd // 2:], ref[..., : hd // 2]], -1)
    want = ref.clone()
    want[:, :, : 2 * nh] = (ref * c + rot * s)[:, :, : 2 * nh]
    assert relerr(x, want.reshape(S * Ln, 3 * H)) < 4e-3
    assert torch.equal(x[:, 2 * H:], qkv[:, 2 * H:])          # v untouched
    L.call("opadpo_rope", x.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), S * Ln, Ln, 2 * nh, hd, 1, None, 0, 0, L.stream())
    assert relerr(x, qkv) < 8e-3                               # inverse rotation restores the input
    # device-resident position offset (decode step replayed from a graph): one row per sequence at position 7
    one = qkv[:S].clone()
    pos = torch.tensor([7], dtype=torch.int32, device=dev())
    L.call("opadpo_rope", one.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), S, 1, 2 * nh, hd, 0, pos.data_ptr(), 0, 0, L.stream())
    r1 = qkv[:S].float().view(S, 3 * nh, hd)
    c7, s7 = torch.cat([f[7], f[7]]).cos().to(dev()), torch.cat([f[7], f[7]]).sin().to(dev())
    w1 = r1.clone()
    w1[:, : 2 * nh] = (r1 * c7 + torch.cat([-r1[..., hd // 2:], r1[..., : hd // 2]], -1) * s7)[:, : 2 * nh]
    assert relerr(one, w1.reshape(S, 3 * H)) < 4e-3


def test_silu_mul(L):
    rows, F = 21, 384
    gu = rnd(rows, 2 * F, seed=1)
    act = torch.empty(rows, F, dtype=BF, device=dev())
    L.call("opadpo_silu_mul_fwd", gu.data_ptr(), act.data_ptr(), rows, F, L.stream())
    g = gu[:, :F].float().requires_grad_(True)
    u = gu[:, F:].float().requires_grad_(True)
    want = torch.nn.functional.silu(g) * u
    assert relerr(act, want) < 4e-3
    dact = rnd(rows, F, seed=2)
    dgu = torch.empty_like(gu)
    L.call("opadpo_silu_mul_bwd", dact.data_ptr(), gu.data_ptr(), dgu.data_ptr(), rows, F, L.stream())
    want.backward(dact.float())
    assert relerr(dgu[:, :F], g.grad) < 4e-3 and relerr(dgu[:, F:], u.grad) < 4e-3


def test_embed_splice(L):
    S, n_txt, P, H, V = 3, 10, 4, 128, 50
    embed, feats = rnd(V, H, seed=1), rnd(2, P, H, seed=2)
    ids = torch.randint(1, V, (S, n_txt))
    ids[0, 2] = -200
    ids[1, 0] = -200
    ids[2, 6] = -200
    tm = torch.ones(S, n_txt, dtype=torch.uint8)
    tm[0, :2] = 0
    tm[2, 8:] = 0
    feat_row = torch.tensor([0, 1, 0], dtype=torch.int32)
    im = torch.ones(S, P, dtype=torch.uint8)
    im[1, 1] = 0
    x = torch.zeros(S, n_txt + P - 1, H, dtype=BF, device=dev())
    x32 = torch.zeros(S, n_txt + P - 1, H, device=dev())
    km = torch.zeros(S, n_txt + P - 1, dtype=torch.uint8, device=dev())
    ids_d, tm_d, fr_d, im_d = ids.to(torch.int32).to(dev()), tm.to(dev()), feat_row.to(dev()), im.to(dev())   # keep alive
    L.call("opadpo_embed_splice", ids_d.data_ptr(), tm_d.data_ptr(), embed.data_ptr(),
           feats.data_ptr(), fr_d.data_ptr(), im_d.data_ptr(), x.data_ptr(), 0, km.data_ptr(),
           S, n_txt, P, H, -200, L.stream())
    L.call("opadpo_embed_splice", ids_d.data_ptr(), tm_d.data_ptr(), embed.data_ptr(),
           feats.data_ptr(), fr_d.data_ptr(), im_d.data_ptr(), x32.data_ptr(), 1, km.data_ptr(),
           S, n_txt, P, H, -200, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(x32, x.float())
    for s in range(S):
        p = int((ids[s] == -200).nonzero()[0, 0])
        e = embed[ids[s].clamp_min(0).to(dev())]
        want = torch.cat([e[:p], feats[feat_row[s]], e[p + 1:]], 0)
        wm = torch.cat([tm[s, :p], im[s], tm[s, p + 1:]], 0)
        assert torch.equal(x[s], want), f"row {s}"
        assert torch.equal(km[s].cpu(), wm), f"mask row {s}"


def test_vision_embed_pieces(L):
    B, IS, patch, h = 2, 28, 14, 128
    G = IS // patch
    P = G * G
    kpad = 640
    px = rnd(B, 3, IS, IS, seed=1)
    cols = torch.empty(B * P, kpad, dtype=BF, device=dev())
    L.call("opadpo_im2col", px.data_ptr(), cols.data_ptr(), B, IS, patch, kpad, L.stream())
    want = torch.nn.functional.unfold(px.float(), patch, stride=patch).transpose(1, 2).reshape(B * P, -1)
    assert torch.equal(cols[:, :588].float(), want) and float(cols[:, 588:].abs().max()) == 0.0
    patches, cls, pos = rnd(B * P, h, seed=2), rnd(h, seed=3), rnd(P + 1, h, seed=4)
    x = torch.empty(B, P + 1, h, dtype=BF, device=dev())
    L.call("opadpo_vision_embed", patches.data_ptr(), cls.data_ptr(), pos.data_ptr(), x.data_ptr(), B, P, h, L.stream())
    w = torch.cat([cls.float().expand(B, 1, h), patches.float().view(B, P, h)], 1) + pos.float()[None]
    assert relerr(x, w) < 4e-3


def test_data_movement(L):
    src = rnd(40, 256, seed=1)
    idx = torch.tensor([5, 0, 39, 7], dtype=torch.int32, device=dev())
    dst = torch.empty(4, 128, dtype=BF, device=dev())
    L.call("opadpo_gather_rows", src.data_ptr(), 256, idx.data_ptr(), dst.data_ptr(), 4, 128, L.stream())
    assert torch.equal(dst, src[idx.long(), :128])
    back = torch.zeros(40, 256, dtype=BF, device=dev())
    L.call("opadpo_scatter_rows", dst.data_ptr(), idx.data_ptr(), back.data_ptr(), 256, 4, 128, L.stream())
    assert torch.equal(back[idx.long(), :128], dst) and float(back.float().abs().sum()) == float(dst.float().abs().sum())
    a = rnd(100, 70, seed=2)
    t = torch.empty(70, 100, dtype=BF, device=dev())
    L.call("opadpo_transpose", a.data_ptr(), t.data_ptr(), 100, 70, L.stream())
    assert torch.equal(t, a.t().contiguous())
    f = torch.randn(1003, device=dev())
    o = torch.empty(1003, dtype=BF, device=dev())
    L.call("opadpo_f32_to_bf16", f.data_ptr(), o.data_ptr(), 1003, L.stream())
    assert torch.equal(o, f.to(BF))
    f2 = torch.randn(9, 64, device=dev())
    o2 = torch.zeros(9, 192, dtype=BF, device=dev())
    L.call("opadpo_f32_to_bf16_strided", f2.data_ptr(), o2.data_ptr(), 9, 64, 192, L.stream())
    assert torch.equal(o2[:, :64], f2.to(BF)) and float(o2[:, 64:].abs().max()) == 0.0


def test_head(L):
    rows, V = 23, 512
    logits = torch.randn(rows, V, device=dev()) * 3
    labels = torch.randint(1, V, (rows,), dtype=torch.int32, device=dev())
    labels[3] = 0
    labels[10] = 0
    logp, ent, lse = (torch.empty(rows, device=dev()) for _ in range(3))
    temp = 0.7
    L.call("opadpo_head_fwd", logits.data_ptr(), V, labels.data_ptr(), 1 / temp, logp.data_ptr(), ent.data_ptr(),
           lse.data_ptr(), rows, V, L.stream())
    z = (logits / temp).requires_grad_(True)
    lsm = torch.log_softmax(z, -1)
    want_lp = lsm.gather(-1, labels.long().unsqueeze(-1)).squeeze(-1) * (labels != 0)
    want_ent = -(lsm.exp() * lsm).sum(-1) * (labels != 0)
    assert maxabs(logp, want_lp) < 1e-4 and maxabs(ent, want_ent) < 1e-4
    assert float(logp[3]) == 0.0 and math.copysign(1.0, float(logp[3])) == -1.0     # -0.0 on pad (Quirk Q4)
    dlogp = torch.randn(rows, device=dev())
    dz = torch.empty(rows, V, dtype=BF, device=dev())
    L.call("opadpo_head_bwd", logits.data_ptr(), V, labels.data_ptr(), lse.data_ptr(), dlogp.data_ptr(), None, None, 1 / temp,
           dz.data_ptr(), V, rows, V, L.stream())
    (want_lp * dlogp).sum().backward(retain_graph=True)
    want_dz = z.grad / temp      # d/d logits
    assert relerr(dz, want_dz) < 5e-3
    assert float(dz[3].float().abs().max()) == 0.0
    # entropy gradient (OPA-SFT regulariser): loss = sum(dlogp * logp) + sum(dent * H)
    dent = torch.randn(rows, device=dev())
    L.call("opadpo_head_bwd", logits.data_ptr(), V, labels.data_ptr(), lse.data_ptr(), dlogp.data_ptr(), ent.data_ptr(), dent.data_ptr(),
           1 / temp, dz.data_ptr(), V, rows, V, L.stream())
    z.grad = None
    ((want_lp * dlogp).sum() + (want_ent * dent).sum()).backward()
    assert relerr(dz, z.grad / temp) < 5e-3 and float(dz[3].float().abs().max()) == 0.0


def test_adamw_and_sumsq(L):
    n = 100_003
    torch.manual_seed(0)
    p = torch.randn(n, device=dev())
    ref = torch.nn.Parameter(p.clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pb = torch.empty(n, dtype=BF, device=dev())
    ss = torch.zeros(1, device=dev())
    for step in range(1, 5):
        g = torch.randn(n, device=dev()) * (5.0 if step % 2 else 0.001)
        ss.zero_()
        L.call("opadpo_sumsq", g.data_ptr(), n, ss.data_ptr(), L.stream())
        assert abs(float(ss) - float((g.double() ** 2).sum())) / float((g.double() ** 2).sum()) < 1e-4
        L.call("opadpo_adamw", p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), pb.data_ptr(), n, 1e-3, 0.9, 0.999,
               1e-8, 0.01, step, ss.data_ptr(), 1.0, 0.5, L.stream())
        ref.grad = g.clone() * 0.5
        torch.nn.utils.clip_grad_norm_([ref], 1.0)
        opt.step()
        assert maxabs(p, ref.detach()) < 2e-6, f"step {step}"
        assert torch.equal(pb, p.to(BF))


@pytest.mark.parametrize("B,nh,hd,ctx,max_ctx", [(3, 2, 128, 70, 96), (1, 4, 128, 900, 1024), (2, 2, 64, 333, 512), (40, 32, 128, 130, 160),
                                                 (2, 3, 128, 1, 64), (8, 32, 128, 750, 768), (1, 1, 64, 2000, 2048),
                                                 (36, 32, 128, 530, 544), (5, 32, 128, 600, 640)])
def test_attn_decode(L, B, nh, hd, ctx, max_ctx):
    """single-token attention over the head-major KV cache; split-KV path (few sequences -> workspace + merge launch)
    and single-block path agree with torch fp32; the device-resident ctx pointer gives the same bits."""
    H = nh * hd
    q = rnd(B, H, seed=1)
    kc, vc = rnd(B, nh, max_ctx, hd, seed=2), rnd(B, nh, max_ctx, hd, seed=3)
    kc[:, :, ctx:] = float("nan")          # slots beyond ctx are uninitialised memory in the product: must never be read
    vc[:, :, ctx:] = float("nan")
    km = torch.ones(B, max_ctx, dtype=torch.uint8, device=dev())
    km[0, : min(4, ctx - 1)] = 0
    if ctx > 40:
        km[B - 1, 17:33] = 0
    ws_bytes = int(L.load().opadpo_attn_decode_workspace_bytes(B, nh, hd, max_ctx))
    ws = torch.zeros(max(ws_bytes, 4), dtype=torch.uint8, device=dev())
    outs = []
    for wsp, wsb in ((ws.data_ptr(), ws_bytes), (None, 0)):
        o = torch.empty(B, H, dtype=BF, device=dev())
        for _ in range(2):
            L.call("opadpo_attn_decode", q.data_ptr(), H, kc.data_ptr(), vc.data_ptr(), o.data_ptr(), km.data_ptr(), B, nh, hd, ctx,
                   None, max_ctx, hd ** -0.5, wsp, wsb, L.stream())
        outs.append(o)
    o2 = torch.empty_like(outs[0])
    posd = torch.tensor([ctx - 1], dtype=torch.int32, device=dev())        # device-resident newest-key position
    L.call("opadpo_attn_decode", q.data_ptr(), H, kc.data_ptr(), vc.data_ptr(), o2.data_ptr(), km.data_ptr(), B, nh, hd, 0,
           posd.data_ptr(), max_ctx, hd ** -0.5, ws.data_ptr(), ws_bytes, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(outs[0], o2)
    qf = q.float().view(B, nh, hd)
    kf, vf = kc.float()[:, :, :ctx], vc.float()[:, :, :ctx]
    sc = torch.einsum("bhd,bhkd->bhk", qf, kf) * hd ** -0.5
    sc = sc.masked_fill(~km[:, None, :ctx].bool(), float("-inf"))
    want = torch.einsum("bhk,bhkd->bhd", torch.softmax(sc, -1), vf).reshape(B, H)
    for o in outs:
        assert relerr(o, want) < 6e-3


def test_rope_kv_append(L):
    """decode-step RoPE + cache append in one launch == rope kernel on q|k + copies into [b, h, pos, :]."""
    B, nh, hd, max_ctx, pos = 5, 4, 128, 48, 29
    H = nh * hd
    qkv = rnd(B, 3 * H, seed=1)
    half = hd // 2
    inv = 1.0 / (10000.0 ** (torch.arange(0, half, device=dev()).float() / half))
    ang = torch.arange(max_ctx, device=dev()).float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    posd = torch.tensor([pos], dtype=torch.int32, device=dev())
    ref = qkv.clone()
    L.call("opadpo_rope", ref.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), B, 1, 2 * nh, hd, 0, posd.data_ptr(), 0, 0, L.stream())
    kc = torch.zeros(B, nh, max_ctx, hd, dtype=BF, device=dev())
    vc = torch.zeros_like(kc)
    got = qkv.clone()
    L.call("opadpo_rope_kv_append", got.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), kc.data_ptr(), vc.data_ptr(), B, nh, hd,
           posd.data_ptr(), max_ctx, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(got[:, :H], ref[:, :H])                               # q rotated in place, same bits
    assert torch.equal(kc[:, :, pos].reshape(B, H), ref[:, H:2 * H])         # rotated k appended
    assert torch.equal(vc[:, :, pos].reshape(B, H), qkv[:, 2 * H:])          # v appended
    kc[:, :, pos] = 0
    vc[:, :, pos] = 0
    assert float(kc.abs().max()) == 0.0 and float(vc.abs().max()) == 0.0     # nothing else touched


@pytest.mark.parametrize("B,nh,hd,pos,max_ctx", [(2, 4, 128, 29, 48), (3, 2, 64, 0, 16), (1, 8, 128, 700, 1024), (5, 32, 128, 301, 512),
                                                 (64, 32, 128, 130, 256), (4, 32, 128, 751, 800), (40, 32, 128, 520, 576), (6, 32, 128, 300, 512)])
def test_attn_decode_fused(L, B, nh, hd, pos, max_ctx):
    """opadpo_attn_decode_fused == opadpo_rope_kv_append followed by opadpo_attn_decode: identical cache contents (bits), attention
    output equal up to the order in which the newest key enters the online softmax; qkv is not modified; slots > pos never read."""
    H = nh * hd
    qkv = rnd(B, 3 * H, seed=1)
    half = hd // 2
    inv = 1.0 / (10000.0 ** (torch.arange(0, half, device=dev()).float() / half))
    ang = torch.arange(max_ctx, device=dev()).float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    posd = torch.tensor([pos], dtype=torch.int32, device=dev())
    kc0, vc0 = rnd(B, nh, max_ctx, hd, seed=2), rnd(B, nh, max_ctx, hd, seed=3)
    kc0[:, :, pos:] = float("nan")                     # the slot being appended and everything after it: uninitialised
    vc0[:, :, pos:] = float("nan")
    km = torch.ones(B, max_ctx, dtype=torch.uint8, device=dev())
    if pos > 6:
        km[0, :4] = 0
    ws_bytes = int(L.load().opadpo_attn_decode_workspace_bytes(B, nh, hd, max_ctx))
    ws = torch.zeros(max(ws_bytes, 4), dtype=torch.uint8, device=dev())
    # two launches
    q1, kc1, vc1 = qkv.clone(), kc0.clone(), vc0.clone()
    o1 = torch.empty(B, H, dtype=BF, device=dev())
    L.call("opadpo_rope_kv_append", q1.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), kc1.data_ptr(), vc1.data_ptr(), B, nh, hd,
           posd.data_ptr(), max_ctx, L.stream())
    L.call("opadpo_attn_decode", q1.data_ptr(), 3 * H, kc1.data_ptr(), vc1.data_ptr(), o1.data_ptr(), km.data_ptr(), B, nh, hd, 0,
           posd.data_ptr(), max_ctx, hd ** -0.5, ws.data_ptr(), ws_bytes, L.stream())
    # one launch, with and without the split-KV workspace
    for wsp, wsb in ((ws.data_ptr(), ws_bytes), (None, 0)):
        q2, kc2, vc2 = qkv.clone(), kc0.clone(), vc0.clone()
        o2 = torch.empty(B, H, dtype=BF, device=dev())
        L.call("opadpo_attn_decode_fused", q2.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), kc2.data_ptr(), vc2.data_ptr(), o2.data_ptr(),
               km.data_ptr(), B, nh, hd, posd.data_ptr(), max_ctx, hd ** -0.5, wsp, wsb, L.stream())
        torch.cuda.synchronize()
        assert torch.equal(q2, qkv)
        assert torch.equal(kc2[:, :, :pos + 1], kc1[:, :, :pos + 1]) and torch.equal(vc2[:, :, :pos + 1], vc1[:, :, :pos + 1])
        assert bool(torch.isnan(kc2[:, :, pos + 1:].float()).all()) and bool(torch.isnan(vc2[:, :, pos + 1:].float()).all())
        assert not bool(torch.isnan(o2.float()).any())
        assert relerr(o2, o1.float()) < 4e-3


def test_attn_decode_fused_last_slot_and_wide_rows(L):
    """Fused decode attention at the LAST cache slot (pos = max_ctx - 1) with the q|k|v rows embedded in a wider buffer (ld > 3H)."""
    B, nh, hd, max_ctx = 3, 4, 128, 40
    pos, H = max_ctx - 1, nh * hd
    wide = rnd(B, 3 * H + 64, seed=1)
    qkv = wide[:, 32:32 + 3 * H]
    half = hd // 2
    inv = 1.0 / (10000.0 ** (torch.arange(0, half, device=dev()).float() / half))
    ang = torch.arange(max_ctx, device=dev()).float()[:, None] * inv[None]
    cos, sin = ang.cos().contiguous(), ang.sin().contiguous()
    posd = torch.tensor([pos], dtype=torch.int32, device=dev())
    kc, vc = rnd(B, nh, max_ctx, hd, seed=2), rnd(B, nh, max_ctx, hd, seed=3)
    km = torch.ones(B, max_ctx, dtype=torch.uint8, device=dev())
    km[1, 5:9] = 0
    kc1, vc1, q1 = kc.clone(), vc.clone(), qkv.clone().contiguous()
    o1 = torch.empty(B, H, dtype=BF, device=dev())
    L.call("opadpo_rope_kv_append", q1.data_ptr(), 3 * H, cos.data_ptr(), sin.data_ptr(), kc1.data_ptr(), vc1.data_ptr(), B, nh, hd, posd.data_ptr(), max_ctx, L.stream())
    L.call("opadpo_attn_decode", q1.data_ptr(), 3 * H, kc1.data_ptr(), vc1.data_ptr(), o1.data_ptr(), km.data_ptr(), B, nh, hd, 0, posd.data_ptr(), max_ctx,
           hd ** -0.5, None, 0, L.stream())
    o2 = torch.empty(B, H, dtype=BF, device=dev())
    before = wide.clone()
    L.call("opadpo_attn_decode_fused", qkv.data_ptr(), wide.stride(0), cos.data_ptr(), sin.data_ptr(), kc.data_ptr(), vc.data_ptr(), o2.data_ptr(), km.data_ptr(),
           B, nh, hd, posd.data_ptr(), max_ctx, hd ** -0.5, None, 0, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(wide, before) and torch.equal(kc, kc1) and torch.equal(vc, vc1)
    assert relerr(o2, o1.float()) < 4e-3


def test_sampler_distribution(L):
    V, rows = 512, 4000
    torch.manual_seed(3)
    base = torch.randn(V, device=dev()) * 2
    logits = base[None].repeat(rows, 1).contiguous()
    out = torch.empty(rows, dtype=torch.int32, device=dev())
    fin = torch.zeros(rows, dtype=torch.uint8, device=dev())
    fin[7] = 1
    L.call("opadpo_sample", logits.data_ptr(), V, rows, V, 0.8, 30, 0.95, 1234, 5, None, fin.data_ptr(), 0, -1, out.data_ptr(), None, L.stream())
    torch.cuda.synchronize()
    assert int(out[7]) == 0
    z = base / 0.8
    kth = torch.topk(z, 30).values[-1]
    z = z.masked_fill(z < kth, float("-inf"))
    s, idx = torch.sort(z)
    rem = torch.softmax(s, -1).cumsum(-1) <= 0.05
    rem[-1] = False
    z[idx[rem]] = float("-inf")
    p = torch.softmax(z, -1)
    picks = out.long()[fin == 0]
    assert bool((p[picks] > 0).all()), "sampled a filtered token"
    emp = torch.bincount(picks, minlength=V).float() / picks.numel()
    assert float((emp - p).abs().max()) < 0.03
    # determinism for (seed, step, row) and pure multinomial path
    out2 = torch.empty_like(out)
    stepd = torch.tensor([5], dtype=torch.int32, device=dev())
    hist = torch.full((8, rows), -7, dtype=torch.int32, device=dev())
    fin2 = fin.clone()
    eos = int(out[0])
    L.call("opadpo_sample", logits.data_ptr(), V, rows, V, 0.8, 30, 0.95, 1234, 0, stepd.data_ptr(), fin2.data_ptr(), 0, eos,
           out2.data_ptr(), hist.data_ptr(), L.stream())
    assert torch.equal(out, out2) and torch.equal(hist[5], out) and int((hist[4] != -7).sum()) == 0
    assert torch.equal(fin2.bool(), fin.bool() | (out == eos))              # rows that drew EOS are marked finished
    L.call("opadpo_sample", logits.data_ptr(), V, rows, V, 1.0, 0, 1.0, 99, 0, None, None, 0, -1, out2.data_ptr(), None, L.stream())
    emp = torch.bincount(out2.long(), minlength=V).float() / rows
    assert float((emp - torch.softmax(base, -1)).abs().max()) < 0.03


@pytest.mark.parametrize("V", [32000, 32768, 40000, 1001])
def test_sampler_filters_large_vocab(L, V):
    """top-k / top-p thresholds at LLaVA's vocabulary size (register-resident radix select; V > 32768 or V % 4 != 0 take the
    generic kernel): every draw lies inside the HF-filtered set, every member of that set is reachable."""
    rows = 256
    g = torch.Generator(device="cpu").manual_seed(V)
    base = (torch.randn(V, generator=g) * 3).to(dev())
    logits = base[None].repeat(rows, 1).contiguous()
    for top_k, top_p, temp in ((30, 0.95, 1.0), (5, 1.0, 0.7), (0, 0.6, 1.0), (1, 1.0, 1.0)):
        out = torch.empty(rows, dtype=torch.int32, device=dev())
        picks = []
        for step in range(8):
            L.call("opadpo_sample", logits.data_ptr(), V, rows, V, temp, top_k, top_p, 77, step, None, None, 0, -1, out.data_ptr(), None, L.stream())
            picks.append(out.clone())
        picks = torch.cat(picks).long()
        z = base.double() / temp
        if top_k:
            kth = torch.topk(z, top_k).values[-1]
            z = z.masked_fill(z < kth, float("-inf"))
        edge = torch.zeros(V, dtype=torch.bool, device=dev())
        if top_p < 1.0:
            sv, idx = torch.sort(z)
            cum = torch.softmax(sv, -1).cumsum(-1)
            rem = cum <= (1 - top_p)
            rem[-1] = False
            # a token whose removal hinges on the last bits of the cumulative mass may go either way (fp32 exp there, fp64 here)
            edge[idx[(cum - (1 - top_p)).abs() < 1e-4]] = True
            z[idx[rem]] = float("-inf")
        p = torch.softmax(z, -1)
        ok = (p > 0) | edge
        assert bool(ok[picks].all()), f"V={V} k={top_k} p={top_p}: sampled a filtered token"
        if top_k == 1:
            assert bool((picks == int(base.argmax())).all())
        emp = torch.bincount(picks, minlength=V).double() / picks.numel()
        assert float((emp - p).abs().max()) < 0.06


def test_sampler_compact_tail_is_exact(L):
    """The sampler's compact tail (top-p threshold, mass sums and the draw on the <= 64 tokens the top-k threshold keeps, by one wave) draws the SAME
    token as the full vocabulary sweeps for every (seed, step, row): distinct rows, several temperatures / k / p, coarse logits (ties at the
    threshold: more than 64 kept tokens fall back to the sweeps), finished rows and the EOS flag."""
    import os
    V, rows = 32000, 512
    g = torch.Generator(device="cpu").manual_seed(11)
    smooth = (torch.randn(rows, V, generator=g) * 3).to(dev())
    coarse = (torch.randint(-6, 7, (rows, V), generator=g).float() * 0.5).to(dev())          # 13 distinct values: thousands of ties at any threshold
    mixed = smooth.clone()
    mixed[:, :40] = 9.0                                                                     # 40 equal maxima: top-k 30 keeps all 40
    fin = torch.zeros(rows, dtype=torch.uint8, device=dev())
    fin[3] = 1
    lib = L.load()
    try:
        for logits in (smooth, coarse, mixed):
            for top_k, top_p, temp in ((30, 0.95, 1.0), (30, 0.95, 0.1), (64, 0.5, 1.3), (5, 1.0, 0.7), (1, 0.9, 1.0), (30, 0.01, 1.0), (200, 0.9, 1.0)):
                res = []
                for full_sweeps in (512, 0):                    # opadpo_set_flags use_tr bit 9
                    lib.opadpo_set_flags(10, 1 | full_sweeps)
                    outs = []
                    for step in range(4):
                        out = torch.full((rows,), -5, dtype=torch.int32, device=dev())
                        f2 = fin.clone()
                        L.call("opadpo_sample", logits.data_ptr(), V, rows, V, temp, top_k, top_p, 4242, step, None, f2.data_ptr(), 0, 17,
                               out.data_ptr(), None, L.stream())
                        outs += [out, f2.int()]
                    res.append(torch.stack(outs))
                torch.cuda.synchronize()
                assert torch.equal(res[0], res[1]), (top_k, top_p, temp)
    finally:
        lib.opadpo_set_flags(10, 1)


def test_errors_are_loud(L):
    a, b = rnd(10, 64), rnd(100, 64)   # N not a multiple of 128
    out = torch.empty(10, 100, dtype=BF, device=dev())
    with pytest.raises(L.OpadpoError):
        L.gemm_nt(a, b, out)


@pytest.mark.parametrize("M,F,K", [(700, 256, 128), (3000, 1408, 192), (40, 128, 64), (2600, 1024, 256)])
def test_gemm_nt_swiglu_pair(L, M, F, K):
    """ACT_SWIGLU_PAIR: weight rows arranged per 128 as [64 gate | 64 up]; the projection's epilogue writes silu(gate) * up
    ([M, F]) - bit-identical to the projection followed by opadpo_silu_mul_fwd (same bf16 rounding of the pre-activations)."""
    L.set_flags(10, True)
    x = rnd(M, K, seed=1)
    wgu = rnd(2 * F, K, scale=0.3, seed=2)                     # rows [0, F) gate, [F, 2F) up
    gu = torch.empty(M, 2 * F, dtype=BF, device=dev())
    L.gemm_nt(x, wgu, gu)
    want = torch.empty(M, F, dtype=BF, device=dev())
    L.call("opadpo_silu_mul_fwd", L.ptr(gu), L.ptr(want), M, F, L.stream())
    w_sw = torch.stack([wgu[:F].view(F // 64, 64, K), wgu[F:].view(F // 64, 64, K)], dim=1).reshape(2 * F, K).contiguous()
    got = torch.full((M + 2, F), 7.0, dtype=BF, device=dev())
    L.gemm_nt(x, w_sw, got[:M], act=L.ACT_SWIGLU_PAIR)
    torch.cuda.synchronize()
    assert torch.equal(got[:M], want)
    assert float((got[M:].float() - 7.0).abs().max()) == 0.0
    ref = torch.nn.functional.silu(x.float() @ wgu[:F].float().t()) * (x.float() @ wgu[F:].float().t())
    assert relerr(got[:M], ref) < 2e-2
    # round 5: the pair epilogue is DIRECT (gate and up lanes exchange through DPP, no LDS) and streams - the 8-workgroup test walk and the 256x256 kernel on
    # every tile give the same bits
    lib = L.load()
    try:
        for variant, flags in ((10, 1 | 1024), (31, 1)):
            lib.opadpo_set_flags(variant, flags)
            g2 = torch.full((M + 2, F), 7.0, dtype=BF, device=dev())
            L.gemm_nt(x, w_sw, g2[:M], act=L.ACT_SWIGLU_PAIR)
            torch.cuda.synchronize()
            assert torch.equal(g2[:M], want), (variant, flags)
            assert float((g2[M:].float() - 7.0).abs().max()) == 0.0
    finally:
        lib.opadpo_set_flags(10, 1)


@pytest.mark.parametrize("R,K,r,mode", [(17, 4096, 0, "bf16"), (17, 4096, 256, "f32"), (18, 4096, 256, "bf16_res"), (19, 2048, 0, "f32_res"),
                                         (22, 1024, 256, "bf16"), (22, 4096, 0, "alpha"), (33, 4096, 256, "bf16")])
def test_gemm_nt_partial_round_tail_tiles(L, R, K, r, mode):
    """A partly filled last round of 256x256 tiles (R row tiles x 16 column tiles: 272 / 288 / 304 / 352 / 528 tiles) runs as quarter
    tiles on the 128x128 kernel (round 2 ran a split-K tail + reduce launch here; removed): every epilogue of the plain kernel
    (bf16 / fp32 out, bf16 / fp32 residual, alpha, K-concatenated LoRA tail, ragged last row tile) against fp32 torch, and BIT-EQUAL to the
    256x256 kernel on every tile and to the 128x128 kernel; rows >= M untouched."""
    L.set_flags(10, True)
    N, M = 4096, R * 256 - 100
    x, w = rnd(M, K, seed=1), rnd(N, K, scale=0.05, seed=2)
    kw = dict(a2=rnd(M, r, seed=3), b2=rnd(N, r, scale=0.05, seed=4)) if r else {}
    want = x.float() @ w.float().t()
    if r:
        want = want + kw["a2"].float() @ kw["b2"].float().t()
    out_dtype = torch.float32 if mode.startswith("f32") else BF
    if mode == "alpha":
        kw["alpha"] = 0.37
        want = want * 0.37
    if mode.endswith("_res"):
        res = rnd(M, N, seed=5) if mode == "bf16_res" else rnd(M, N, seed=5).float() * 1.001
        kw["residual"] = res
        want = want + res.float()
    got = torch.full((M + 3, N), 7.0, dtype=out_dtype, device=dev())
    L.gemm_nt(x, w, got[:M], **kw)
    again = torch.empty(M, N, dtype=out_dtype, device=dev())
    L.gemm_nt(x, w, again, **kw)
    L.set_flags(4, True)
    small = torch.empty(M, N, dtype=out_dtype, device=dev())
    L.gemm_nt(x, w, small, **kw)
    L.set_flags(31, True)                  # the 256x256 kernel on EVERY tile (no tail handling)
    whole = torch.empty(M, N, dtype=out_dtype, device=dev())
    L.gemm_nt(x, w, whole, **kw)
    L.set_flags(10, True)
    torch.cuda.synchronize()
    assert relerr(got[:M], want) < (3e-3 if out_dtype == BF else 2e-5 * (K ** 0.5))
    assert float((got[M:].float() - 7.0).abs().max()) == 0.0
    assert torch.equal(got[:M], again)
    # round 3: the tail of a partly filled last round runs as QUARTER tiles over the full K range (same k order per element), so the
    # result is bit-identical to the 256x256 kernel on every tile AND to the 128x128 kernel - which tiles are tail tiles (a function of
    # the batch's row count) cannot change a single bit of the output
    assert torch.equal(got[:M], whole), "quarter-tile tail != 256x256 kernel on every tile"
    assert torch.equal(got[:M], small), "256x256 dispatch != 128x128 kernel"


@pytest.mark.parametrize("M,F,K,r", [(300, 256, 128, 64), (1000, 768, 256, 0), (257, 1536, 512, 256), (3, 256, 64, 64)])
def test_gemm_nt_swiglu_bwd(L, M, F, K, r):
    """ACT_SWIGLU_BWD: dgrad of the down projection (+ K-concatenated LoRA tail) with opadpo_silu_mul_bwd in its epilogue ==
    the projection followed by opadpo_silu_mul_bwd, bit for bit; rows >= M of the output untouched; an unsupported width is refused."""
    L.set_flags(10, True)
    dy = rnd(M, K, seed=1)
    wd_t = rnd(F, K, scale=0.3, seed=2)
    gu = rnd(M, 2 * F, seed=3)
    kw = dict(a2=rnd(M, r, seed=4), b2=rnd(F, r, scale=0.3, seed=5)) if r else {}
    d_act = torch.empty(M, F, dtype=BF, device=dev())
    L.gemm_nt(dy, wd_t, d_act, **kw)
    want = torch.empty(M, 2 * F, dtype=BF, device=dev())
    L.call("opadpo_silu_mul_bwd", L.ptr(d_act), L.ptr(gu), L.ptr(want), M, F, L.stream())
    got = torch.full((M + 2, 2 * F), 7.0, dtype=BF, device=dev())
    L.gemm_nt(dy, wd_t, got[:M], residual=gu, act=L.ACT_SWIGLU_BWD, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got[:M], want)
    assert float((got[M:].float() - 7.0).abs().max()) == 0.0
    with pytest.raises(L.OpadpoError):         # N = 128 is not a whole 256-column tile: no silent fallback
        L.gemm_nt(dy, wd_t[:128], got[:M, :256], residual=gu[:, :256], act=L.ACT_SWIGLU_BWD)


@pytest.mark.parametrize("M,tr", [(1, 1), (4, 1), (8, 1), (9, 1), (16, 1), (17, 1), (24, 1), (32, 1), (40, 1), (64, 1), (4, 17), (16, 17), (24, 17)])
@pytest.mark.parametrize("F,K", [(128, 256), (1408, 4096), (384, 11008), (11008, 256)])
def test_gemm_nt_swiglu_pair_decode(L, M, tr, F, K):
    """The same fused SwiGLU epilogue in the weight-streaming (decode, M <= 64) kernel: bit-identical to the streaming projection
    followed by opadpo_silu_mul_fwd; rows >= M untouched."""
    L.set_flags(10, tr)
    x = rnd(M, K, seed=3)
    wgu = rnd(2 * F, K, scale=0.3, seed=4)
    w_sw = torch.stack([wgu[:F].view(F // 64, 64, K), wgu[F:].view(F // 64, 64, K)], dim=1).reshape(2 * F, K).contiguous()
    gu = torch.empty(M, 2 * F, dtype=BF, device=dev())
    want = torch.empty(M, F, dtype=BF, device=dev())
    got = torch.full((M + 2, F), 7.0, dtype=BF, device=dev())
    with L.decode_schedule():
        L.gemm_nt(x, wgu, gu)
        L.call("opadpo_silu_mul_fwd", L.ptr(gu), L.ptr(want), M, F, L.stream())
        L.gemm_nt(x, w_sw, got[:M], act=L.ACT_SWIGLU_PAIR)
    torch.cuda.synchronize()
    L.set_flags(10, True)
    assert torch.equal(got[:M], want)
    assert float((got[M:].float() - 7.0).abs().max()) == 0.0
    ref = torch.nn.functional.silu(x.float() @ wgu[:F].float().t()) * (x.float() @ wgu[F:].float().t())
    assert relerr(got[:M], ref) < 2e-2


@pytest.mark.parametrize("S,Lp,seg", [(3, 150, (0, 0)), (2, 200, (120, 40)), (1, 700, (300, 100)), (5, 37, (17, 10))])
@pytest.mark.parametrize("lora", [False, True])
def test_gemm_nt_rope(L, S, Lp, seg, lora):
    """opadpo_gemm_nt_rope: q|k|v projection with the rotary embedding in its epilogue == projection followed by opadpo_rope on the
    stored tensor (both rotate the bf16-rounded projection; fp32 contraction order may differ by one bf16 ulp on a few elements)."""
    L.set_flags(10, True)
    nh, hd, K, r = 2, 128, 192, 64
    H = nh * hd
    M = S * Lp
    x, w = rnd(M, K, seed=1), rnd(3 * H, K, scale=0.3, seed=2)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    f = torch.outer(torch.arange(Lp, dtype=torch.float32), inv)
    cos, sin = f.cos().to(dev()).contiguous(), f.sin().to(dev()).contiguous()
    kw = {}
    if lora:
        kw = dict(a2=rnd(M, 3 * r, seed=3), b2=rnd(3 * H, r, scale=0.3, seed=4), a2_group_n=H, a2_group_stride=r)
    want = torch.empty(M, 3 * H, dtype=BF, device=dev())
    L.gemm_nt(x, w, want, **kw)
    v_before = want[:, 2 * H:].clone()
    L.call("opadpo_rope", L.ptr(want), 3 * H, L.ptr(cos), L.ptr(sin), M, Lp, 2 * nh, hd, 0, None, seg[0], seg[1], L.stream())
    got = torch.full((M + 1, 3 * H), 7.0, dtype=BF, device=dev())
    L.gemm_nt_rope(x, w, got[:M], cos, sin, Lp, 2 * H, seg, **kw)
    torch.cuda.synchronize()
    assert torch.equal(got[:M, 2 * H:], v_before)                      # v columns: untouched by the rotation
    assert float((got[M:].float() - 7.0).abs().max()) == 0.0
    d = (got[:M].float() - want.float()).abs()
    tol = 2.0 ** -7 * want.float().abs() + 1e-6                         # one bf16 ulp
    assert bool((d <= tol).all()), f"max excess {(d - tol).max().item()}"
    assert float((d > 0).float().mean()) < 0.02                        # and almost everywhere identical


@pytest.mark.parametrize("lora", [False, True])
def test_gemm_nt_rope_pos(L, lora):
    """opadpo_gemm_nt_rope_pos: the q|k|v projection with the TABLE-FREE rotary epilogue (per-row positions, hardware sin / cos of the
    fractional revolution per (row, frequency)) == projection followed by the rotation of the
    bf16-rounded result with exact fp32 angles: within one bf16 ulp everywhere (the angles differ by <= 2e-4 rad), v columns and rows
    >= M untouched.  Row positions as a ragged batch has them: sequences of different lengths, packed responses restarting at their
    prefix end, large positions (precision of the fractional revolution), runs shorter than the 4-row stride of a lane."""
    L.set_flags(10, True)
    nh, hd, K, r = 2, 128, 192, 64
    H = nh * hd
    pos = []
    for n_pfx, resp in ((300, (60, 45)), (7, (3, 2)), (1500, (130, 1)), (64, (64, 64)), (2, ())):
        pos += list(range(n_pfx))
        for n in resp:
            pos += list(range(n_pfx, n_pfx + n))
    pos += [1999, 0, 5, 4, 3, 1000, 1001]          # no run at all
    M = len(pos)
    assert M % 256 != 0
    row_pos = torch.tensor(pos, dtype=torch.int32, device=dev())
    x, w = rnd(M, K, seed=1), rnd(3 * H, K, scale=0.3, seed=2)
    kw = {}
    if lora:
        kw = dict(a2=rnd(M, 3 * r, seed=3), b2=rnd(3 * H, r, scale=0.3, seed=4), a2_group_n=H, a2_group_stride=r)
    plain = torch.empty(M, 3 * H, dtype=BF, device=dev())
    L.gemm_nt(x, w, plain, **kw)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    ang = torch.outer(torch.tensor(pos, dtype=torch.float32), inv).to(dev())          # HF: fp32 angle = pos * inv_freq
    cos, sin = ang.cos()[:, None, :], ang.sin()[:, None, :]
    qk = plain[:, :2 * H].float().view(M, 2 * nh, hd)
    x1, x2 = qk[..., :hd // 2], qk[..., hd // 2:]
    want = torch.cat([x1 * cos - x2 * sin, x2 * cos + x1 * sin], -1).view(M, 2 * H)
    got = torch.full((M + 1, 3 * H), 7.0, dtype=BF, device=dev())
    a2, b2 = kw.get("a2"), kw.get("b2")
    L.call("opadpo_gemm_nt_rope_pos", L.ptr(x), x.stride(0), L.ptr(w), w.stride(0), K, L.ptr(a2), a2.stride(0) if lora else 0,
           L.ptr(b2), b2.stride(0) if lora else 0, r if lora else 0, kw.get("a2_group_n", 0), kw.get("a2_group_stride", 0),
           L.ptr(got), got.stride(0), M, 3 * H, L.ptr(row_pos), 10000.0, 2 * H, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(got[:M, 2 * H:], plain[:, 2 * H:])                  # v columns: not rotated
    assert float((got[M:].float() - 7.0).abs().max()) == 0.0
    d = (got[:M, :2 * H].float() - want).abs()
    tol = 2.0 ** -7 * want.abs() + 2e-4 * (x1.abs().amax() + 1.0)          # one bf16 ulp of the result + the angle error on an O(1) operand
    assert bool((d <= tol).all()), f"max excess {(d - tol).max().item()}"
    assert float((got[:M, :2 * H].float() - want.to(BF).float()).abs().gt(0).float().mean()) < 0.06      # almost everywhere the same bf16 value
    # a row's result does not depend on where the row sits in the batch: the same rows behind 37 other rows -> the same bits
    sh = 37
    x2 = torch.cat([rnd(sh, K, seed=9), x], 0).contiguous()
    pos2 = torch.cat([torch.arange(sh, dtype=torch.int32, device=dev()), row_pos]).contiguous()
    kw2 = dict(kw)
    if lora:
        kw2["a2"] = torch.cat([rnd(sh, 3 * r, seed=10), a2], 0).contiguous()
    got2 = torch.empty(M + sh, 3 * H, dtype=BF, device=dev())
    a22 = kw2.get("a2")
    L.call("opadpo_gemm_nt_rope_pos", L.ptr(x2), x2.stride(0), L.ptr(w), w.stride(0), K, L.ptr(a22), a22.stride(0) if lora else 0,
           L.ptr(b2), b2.stride(0) if lora else 0, r if lora else 0, kw.get("a2_group_n", 0), kw.get("a2_group_stride", 0),
           L.ptr(got2), got2.stride(0), M + sh, 3 * H, L.ptr(pos2), 10000.0, 2 * H, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(got2[sh:], got[:M])
    # round 5: on the streaming kernel the rotation runs in the DIRECT epilogue (partner columns d / d + 64 meet through DPP, one sin / cos pair per (row,
    # frequency)); the 8-workgroup test walk sends this small problem there - same bits as the staged epilogue above
    lib = L.load()
    try:
        lib.opadpo_set_flags(10, 1 | 1024)
        got3 = torch.full((M + 1, 3 * H), 7.0, dtype=BF, device=dev())
        L.call("opadpo_gemm_nt_rope_pos", L.ptr(x), x.stride(0), L.ptr(w), w.stride(0), K, L.ptr(a2), a2.stride(0) if lora else 0,
               L.ptr(b2), b2.stride(0) if lora else 0, r if lora else 0, kw.get("a2_group_n", 0), kw.get("a2_group_stride", 0),
               L.ptr(got3), got3.stride(0), M, 3 * H, L.ptr(row_pos), 10000.0, 2 * H, L.stream())
        torch.cuda.synchronize()
    finally:
        lib.opadpo_set_flags(10, 1)
    assert float((got3[M:].float() - 7.0).abs().max()) == 0.0
    d3 = (got3[:M, :2 * H].float() - want).abs()
    assert bool((d3 <= tol).all()), f"direct epilogue: max excess {(d3 - tol).max().item()}"
    assert torch.equal(got3[:M], got[:M]), f"direct != staged rotary epilogue on {int((got3[:M] != got[:M]).sum())} elements"


@pytest.mark.parametrize("M", [64, 50, 33, 16, 5])
@pytest.mark.parametrize("shape", [(4096, 4096), (4096, 11008), (12288, 4096), (1024, 512)])
@pytest.mark.parametrize("kernel", [0, 1, 3 | (1 << 2), 3 | (2 << 2), 3 | (3 << 2)])
def test_gemm_nt_decode_modes(L, M, shape, kernel):
    """opadpo_gemm_nt_decode: bf16 output, fp32 K-split partial tiles (their sum in slice order) and the SwiGLU-pair epilogue, against
    torch fp32; rows >= M untouched.  kernel 0 = the library's choice = gemm_nt_dec64x_kernel (weights global -> registers in whole 128-byte
    lines, activations shared through LDS in 256-deep chunks by two loader waves) with the rows per workgroup chosen by shape; 3 + (1 / 2 / 3)
    << 2 forces 48 / 64 / 128 weight rows per workgroup where the shape allows; 1 = the LDS-ring kernel of rounds 2-4 (64 weight rows x <= 64
    tokens x one K-slice per workgroup, a barrier per k-tile)."""
    import ctypes as C
    N, K = shape
    lib = L.load()
    L.set_flags(10, 1 | (kernel << 5))
    a, w = rnd(M, K, scale=0.5, seed=1), rnd(N, K, scale=0.05, seed=2)
    want = a.float() @ w.float().t()
    ob = torch.full((M + 2, N), 7.0, dtype=BF, device=dev())
    L.call("opadpo_gemm_nt_decode", L.ptr(a), K, L.ptr(w), K, K, L.ptr(ob), N, 0, M, N, 1, L.stream())
    assert relerr(ob[:M], want) < 6e-3 and float((ob[M:].float() - 7.0).abs().max()) == 0.0
    for splits in (0, 1, 3):
        S = lib.opadpo_gemm_nt_decode_splits(N, K, splits)
        assert 1 <= S <= K // 64
        part = torch.full((S, M, N), 3.0, device=dev())
        L.call("opadpo_gemm_nt_decode", L.ptr(a), K, L.ptr(w), K, K, L.ptr(part), N, 1, M, N, splits, L.stream())
        assert relerr(part.sum(0), want) < 1e-5, (splits, S)
        if S > 1:
            assert float(part[0].abs().max()) > 0 and relerr(part[0], want) > 1e-2      # really split
    # SwiGLU pair: rows per 128 = [64 gate | 64 up]
    F = N // 2
    z = want.view(M, F // 64, 2, 64)
    zb = z.to(BF).float()
    oa = torch.full((M + 1, F), 5.0, dtype=BF, device=dev())
    L.call("opadpo_gemm_nt_decode", L.ptr(a), K, L.ptr(w), K, K, L.ptr(oa), F, 2, M, N, 1, L.stream())
    assert relerr(oa[:M], (torch.nn.functional.silu(zb[:, :, 0]) * zb[:, :, 1]).reshape(M, F)) < 2e-2
    assert float((oa[M:].float() - 5.0).abs().max()) == 0.0
    # strided operands (rows inside wider buffers)
    abig, wbig = rnd(M, K + 64, scale=0.5, seed=3), rnd(N, K + 128, scale=0.05, seed=4)
    o2 = torch.empty(M, N, dtype=BF, device=dev())
    L.call("opadpo_gemm_nt_decode", L.ptr(abig), K + 64, L.ptr(wbig), K + 128, K, L.ptr(o2), N, 0, M, N, 1, L.stream())
    assert relerr(o2, abig[:, :K].float() @ wbig[:, :K].float().t()) < 6e-3
    # repeated runs are bit-identical (DMA ring / barrier race screen)
    o3 = torch.empty_like(o2)
    for _ in range(5):
        L.call("opadpo_gemm_nt_decode", L.ptr(abig), K + 64, L.ptr(wbig), K + 128, K, L.ptr(o3), N, 0, M, N, 1, L.stream())
        assert torch.equal(o2, o3)
    L.set_flags(10, True)


@pytest.mark.parametrize("M", [8, 13, 64])
def test_gemm_nt_stream_routes_to_decode_kernel(L, M):
    """opadpo_gemm_nt with the stream hint and nothing fused (8-64 token rows, plain bf16 / fp32 or SwiGLU-pair store) hands the problem to the
    whole-line kernel of opadpo_gemm_nt_decode: the two entry points agree bit for bit, column-window operands included; a residual keeps the
    call on the 8-row kernels (still right)."""
    N, K = 1024, 512
    L.set_flags(10, True)
    A = rnd(M, K + 64, scale=0.5, seed=21); W = rnd(N, K + 32, scale=0.05, seed=22)
    a, w = A[:, 32:32 + K], W[:, 8:8 + K]
    want = a.float() @ w.float().t()
    O = torch.full((M + 1, N + 16), 5.0, dtype=BF, device=dev()); out = O[:M, 8:8 + N]
    with L.decode_schedule():
        L.gemm_nt(a, w, out)
    ref = torch.empty(M, N, dtype=BF, device=dev())
    L.call("opadpo_gemm_nt_decode", L.ptr(a), A.stride(0), L.ptr(w), W.stride(0), K, L.ptr(ref), N, 0, M, N, 1, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(out, ref) and relerr(out, want) < 6e-3
    assert float((O[M].float() - 5.0).abs().max()) == 0.0 and float((O[:M, :8].float() - 5.0).abs().max()) == 0.0 and float((O[:M, 8 + N:].float() - 5.0).abs().max()) == 0.0
    of = torch.empty(M, N, device=dev()); rf = torch.empty(1, M, N, device=dev())
    with L.decode_schedule():
        L.gemm_nt(a, w, of)
    L.call("opadpo_gemm_nt_decode", L.ptr(a), A.stride(0), L.ptr(w), W.stride(0), K, L.ptr(rf), N, 1, M, N, 1, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(of, rf[0]) and relerr(of, want) < 1e-5
    F = N // 2
    os_ = torch.empty(M, F, dtype=BF, device=dev()); rs = torch.empty(M, F, dtype=BF, device=dev())
    with L.decode_schedule():
        L.gemm_nt(a, w, os_, act=L.ACT_SWIGLU_PAIR)
    L.call("opadpo_gemm_nt_decode", L.ptr(a), A.stride(0), L.ptr(w), W.stride(0), K, L.ptr(rs), F, 2, M, N, 1, L.stream())
    torch.cuda.synchronize()
    assert torch.equal(os_, rs)
    res = torch.randn(M, N, device=dev()); o2 = torch.empty(M, N, device=dev())
    with L.decode_schedule():
        L.gemm_nt(a, w, o2, residual=res)
    torch.cuda.synchronize()
    assert relerr(o2, want + res) < 2e-5


@pytest.mark.parametrize("shape,mode", [((15360, 5120), 0), ((5120, 13824), 1), ((27648, 5120), 2), ((5120, 5120), 1)])
def test_gemm_nt_decode_13b_shapes(L, shape, mode):
    """The decode projections of LLaVA-1.5-13B (hidden 5120, ffn 13824) at 64 and 40 tokens through the library's choice of kernel and rows per
    workgroup: K ranges that are not whole 256-deep activation chunks per K-slice, 120 / 320-workgroup grids."""
    N, K = shape
    lib = L.load()
    L.set_flags(10, 1)
    for M in (64, 40):
        a, w = rnd(M, K, scale=0.5, seed=11), rnd(N, K, scale=0.05, seed=12)
        want = a.float() @ w.float().t()
        if mode == 0:
            ob = torch.empty(M, N, dtype=BF, device=dev())
            L.call("opadpo_gemm_nt_decode", L.ptr(a), K, L.ptr(w), K, K, L.ptr(ob), N, 0, M, N, 1, L.stream())
            assert relerr(ob, want) < 6e-3
        elif mode == 1:
            S = lib.opadpo_gemm_nt_decode_splits(N, K, 0)
            part = torch.full((S, M, N), 3.0, device=dev())
            L.call("opadpo_gemm_nt_decode", L.ptr(a), K, L.ptr(w), K, K, L.ptr(part), N, 1, M, N, 0, L.stream())
            assert relerr(part.sum(0), want) < 1e-5, S
        else:
            F = N // 2
            zb = want.view(M, F // 64, 2, 64).to(BF).float()
            oa = torch.empty(M, F, dtype=BF, device=dev())
            L.call("opadpo_gemm_nt_decode", L.ptr(a), K, L.ptr(w), K, K, L.ptr(oa), F, 2, M, N, 1, L.stream())
            assert relerr(oa, (torch.nn.functional.silu(zb[:, :, 0]) * zb[:, :, 1]).reshape(M, F)) < 2e-2


@pytest.mark.parametrize("resid_f32", [True, False])
def test_rmsnorm_sum_fwd(L, resid_f32):
    rows, H, S = 37, 4096, 4
    g = torch.Generator().manual_seed(5)
    resid = torch.randn(rows, H, generator=g).to(dev())
    if not resid_f32:
        resid = resid.to(BF)
    part = torch.randn(S, rows, H, generator=g).to(dev())
    w = (1.0 + 0.1 * torch.randn(H, generator=g)).to(dev()).to(BF)
    x_out = torch.empty(rows, H, device=dev())
    y = torch.empty(rows, H, dtype=BF, device=dev())
    rstd = torch.empty(rows, device=dev())
    for n in (S, 0):
        L.call("opadpo_rmsnorm_sum_fwd", L.ptr(resid), int(resid_f32), L.ptr(part) if n else None, n, rows * H, L.ptr(w), L.ptr(x_out), L.ptr(y),
               L.ptr(rstd), rows, H, 1e-5, L.stream())
        want_x = resid.float()
        for s_ in range(n):
            want_x = want_x + part[s_]
        assert torch.equal(x_out, want_x), "the slices are added in order on top of the residual: exact in fp32"
        r = torch.rsqrt(want_x.pow(2).mean(-1, keepdim=True) + 1e-5)
        assert relerr(y, want_x * r * w.float()) < 5e-3 and relerr(rstd, r.squeeze(-1)) < 1e-5


def test_gemm_tn_group_equals_single_launches(L):
    """opadpo_gemm_tn_group: the 8 LoRA wgrads of a decoder layer (7B shapes, M = 3000) as ONE launch of the 256x256 kernel ==
    8 single launches (fp32 atomics: same sums up to accumulation order) == torch fp32; a group with a problem the big kernel
    cannot take (N2 = 128) falls back to single launches."""
    import ctypes as C
    M, H, F, r = 3000, 1024, 2816, 256
    g = torch.Generator().manual_seed(3)
    mk = lambda n: (torch.randn(M, n, generator=g) * 0.3).to(BF).to(dev())
    dY, t_d, dt_r, act, d_gu, t_gu, dt_2r, n2 = mk(H), mk(r), mk(r), mk(F), mk(2 * F), mk(2 * r), mk(2 * r), mk(H)
    # (P, Q, N1, N2, q_group_n1, q_group_stride): b_d, a_d, b_gu (grouped Q columns), a_gu, and three more of the same kinds
    probs = [(dY, t_d, H, r, 0, 0), (dt_r, act, r, F, 0, 0), (d_gu, t_gu, 2 * F, r, F, r), (dt_2r, n2, 2 * r, H, 0, 0),
             (dY, t_d, H, r, 0, 0), (dt_r, n2, r, H, 0, 0), (dt_2r, act, 2 * r, F, 0, 0)]

    def run(group):
        outs = [torch.zeros(n1, n2_, device=dev()) for _, _, n1, n2_, _, _ in probs]
        if group:
            n = len(probs)
            arr = lambda vals, T: (T * n)(*vals)
            L.call("opadpo_gemm_tn_group", n, arr([p_[0].data_ptr() for p_ in probs], C.c_void_p), arr([p_[0].stride(0) for p_ in probs], C.c_int),
                   arr([p_[1].data_ptr() for p_ in probs], C.c_void_p), arr([p_[1].stride(0) for p_ in probs], C.c_int),
                   arr([o.data_ptr() for o in outs], C.c_void_p), arr([o.stride(0) for o in outs], C.c_int), M,
                   arr([p_[2] for p_ in probs], C.c_int), arr([p_[3] for p_ in probs], C.c_int), arr([p_[4] for p_ in probs], C.c_int),
                   arr([p_[5] for p_ in probs], C.c_int), 1.0, L.stream())
        else:
            for (P, Q, n1, n2_, qg, qs), o in zip(probs, outs):
                L.gemm_tn(P, Q, o, q_group_n1=qg, q_group_stride=qs)
        torch.cuda.synchronize()
        return outs

    single, grouped = run(False), run(True)
    for (P, Q, n1, n2_, qg, qs), a, b in zip(probs, single, grouped):
        if qg:
            want = torch.cat([P[:, i * qg:(i + 1) * qg].float().t() @ Q[:, i * qs:i * qs + n2_].float() for i in range(n1 // qg)], 0)
        else:
            want = P.float().t() @ Q[:, :n2_].float()
        assert relerr(b, want) < 1e-4 and relerr(a, b) < 1e-5
    # not groupable (N2 = 128): falls back, same numbers
    P, Q = mk(256), mk(128)
    o1, o2 = torch.zeros(256, 128, device=dev()), torch.zeros(256, 128, device=dev())
    L.gemm_tn(P, Q, o1)
    one = lambda v, T: (T * 1)(v)
    L.call("opadpo_gemm_tn_group", 1, one(P.data_ptr(), C.c_void_p), one(256, C.c_int), one(Q.data_ptr(), C.c_void_p), one(128, C.c_int),
           one(o2.data_ptr(), C.c_void_p), one(128, C.c_int), M, one(256, C.c_int), one(128, C.c_int), None, None, 1.0, L.stream())
    torch.cuda.synchronize()
    assert relerr(o2, o1) < 1e-5 and relerr(o1, P.float().t() @ Q.float()) < 1e-4


@pytest.mark.parametrize("M", [3000, 24500, 130])
def test_gemm_tn_group_deterministic_form(L, M):
    """opadpo_gemm_tn_group_det (round 4; what the context's backward runs): partial tiles to a workspace + ordered reduce instead of fp32
    atomics.  == torch fp32, == the atomic form up to summation order, ACCUMULATES into C (gradient accumulation), and repeated launches are
    BIT-identical (the atomic form is not).  M = 24500: the bench's ragged row count (383 K-steps of 64 rows, one run per CU crossing tiles);
    M = 130: fewer K-steps than K-chunks.  The same call on one problem at a time (the compact top layer's fallback) agrees bit for bit with
    itself over repeats too; a too-small workspace and a problem the 256x256 kernel cannot take are refused."""
    import ctypes as C
    L.set_flags(True, True)          # the process switches of an earlier test (use_tr bit 3 = the 128x128 wgrad kernel) must not decide what this one measures
    H, F, r = 1024, 2816, 256
    g = torch.Generator().manual_seed(5)
    mk = lambda n: (torch.randn(M, n, generator=g) * 0.3).to(BF).to(dev())
    dY, t_d, dt_r, act, d_gu, t_gu, dt_2r, n2 = mk(H), mk(r), mk(r), mk(F), mk(2 * F), mk(2 * r), mk(2 * r), mk(H)
    probs = [(dY, t_d, H, r, 0, 0), (dt_r, act, r, F, 0, 0), (d_gu, t_gu, 2 * F, r, F, r), (dt_2r, n2, 2 * r, H, 0, 0),
             (dY, t_d, H, r, 0, 0), (dt_r, n2, r, H, 0, 0), (dt_2r, act, 2 * r, F, 0, 0)]
    lib = L.load()

    def arrs(ps, outs):
        n = len(ps)
        arr = lambda vals, T: (T * n)(*vals)
        return (n, arr([p_[0].data_ptr() for p_ in ps], C.c_void_p), arr([p_[0].stride(0) for p_ in ps], C.c_int),
                arr([p_[1].data_ptr() for p_ in ps], C.c_void_p), arr([p_[1].stride(0) for p_ in ps], C.c_int),
                arr([o.data_ptr() for o in outs], C.c_void_p), arr([o.stride(0) for o in outs], C.c_int), M,
                arr([p_[2] for p_ in ps], C.c_int), arr([p_[3] for p_ in ps], C.c_int), arr([p_[4] for p_ in ps], C.c_int), arr([p_[5] for p_ in ps], C.c_int), 1.0)

    def need_of(ps):
        n = len(ps)
        arr = lambda vals: (C.c_int * n)(*vals)
        return lib.opadpo_gemm_tn_group_workspace_bytes(n, M, arr([p_[2] for p_ in ps]), arr([p_[3] for p_ in ps]), arr([p_[4] for p_ in ps]))

    need = need_of(probs)
    assert 0 < need <= 256 * 3 * 65536 * 4
    ws = torch.empty(need, dtype=torch.uint8, device=dev())

    def run(det, base=0.0, ps=probs):
        outs = [torch.full((n1, n2_), base, device=dev()) for _, _, n1, n2_, _, _ in ps]
        a = arrs(ps, outs)
        if det:
            L.call("opadpo_gemm_tn_group_det", *a, ws.data_ptr(), need, L.stream())
        else:
            L.call("opadpo_gemm_tn_group", *a, L.stream())
        torch.cuda.synchronize()
        return outs

    d1, d2, at = run(True), run(True), run(False)
    for (P, Q, n1, n2_, qg, qs), a, b, c in zip(probs, d1, d2, at):
        if qg:
            want = torch.cat([P[:, i * qg:(i + 1) * qg].float().t() @ Q[:, i * qs:i * qs + n2_].float() for i in range(n1 // qg)], 0)
        else:
            want = P.float().t() @ Q[:, :n2_].float()
        assert relerr(a, want) < 1e-4 and relerr(a, c) < 1e-5
        assert torch.equal(a, b), "deterministic wgrad flush differs between two launches"
    acc = run(True, base=1.5)                                       # C is accumulated into, not overwritten
    for a, b in zip(d1, acc):
        assert float((b - 1.5 - a).abs().max()) <= 2e-6 * float(a.abs().max()) + 1e-6
    for k in (0, 2):                                                # one problem per call (how a list of mixed row counts is run)
        s1, s2 = run(True, ps=probs[k:k + 1])[0], run(True, ps=probs[k:k + 1])[0]
        assert torch.equal(s1, s2) and relerr(s1, d1[k]) < 1e-5
    with pytest.raises(L.OpadpoError):
        L.call("opadpo_gemm_tn_group_det", *arrs(probs, d1), ws.data_ptr(), need - 1, L.stream())
    bad = [(mk(256), mk(128), 256, 128, 0, 0)]
    assert need_of(bad) == 0
    with pytest.raises(L.OpadpoError):
        L.call("opadpo_gemm_tn_group_det", *arrs(bad, [torch.zeros(256, 128, device=dev())]), ws.data_ptr(), need, L.stream())
    # a MIXED list (one problem the 256x256 kernel cannot take among good ones) is refused too: its odd member would flush with fp32 atomics
    # behind the deterministic entry point (round-4 review)
    mixed = probs[:2] + bad
    assert need_of(mixed) == 0
    with pytest.raises(L.OpadpoError):
        L.call("opadpo_gemm_tn_group_det", *arrs(mixed, [d1[0], d1[1], torch.zeros(256, 128, device=dev())]), ws.data_ptr(), need, L.stream())


@pytest.mark.parametrize("seg", [(0, 0), (200, 140)])
def test_attn_skip_masked_q_tiles(L, seg):
    """causal | OPADPO_ATTN_SKIP_MASKED_Q: q tiles of 64 positions that are all masked as keys (trailing padding) write zeros and are
    skipped in the backward; every other row, and every gradient (with zero dO on the padding rows, what the LLM backward produces),
    is BIT-identical to the plain causal kernels."""
    L.set_flags(True, True)
    S, Ln, nh, hd = 3, 480, 2, 128
    H = nh * hd
    qkv = rnd(S * Ln, 3 * H, scale=0.7, seed=21)
    dout = rnd(S * Ln, H, scale=1.0, seed=22)
    km = torch.ones(S, Ln, dtype=torch.uint8, device=dev())
    km[0, :9] = 0                      # left padding (partial tile)
    km[0, 300:340] = 0                 # a masked stretch that does not cover a whole tile
    km[1, 250:340] = 0                 # covers tile [256, 320) entirely (inside segment 0 when packed: 200 + 140 = 340)
    km[1, 400:] = 0                    # trailing padding: tiles [448, 480) dead, [384, 448) partially
    km[2, 128:] = 0                    # almost everything is padding
    dout.view(S, Ln, H)[km == 0] = 0   # padding rows carry no output gradient
    st = L.stream()
    scale = hd ** -0.5
    res = {}
    for causal in (1, L.CAUSAL_SKIP_MASKED_Q):
        o = torch.full((S * Ln, H), 9.0, dtype=BF, device=dev())
        lse = torch.zeros(S, nh, Ln, device=dev())
        L.call("opadpo_attn_fwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(), H,
               lse.data_ptr(), L.ptr(km), S, Ln, nh, hd, causal, scale, seg[0], seg[1], st)
        dqkv = torch.full((S * Ln, 3 * H), 5.0, dtype=BF, device=dev())
        delta = torch.zeros(S, nh, Ln, device=dev())
        L.call("opadpo_attn_bwd", qkv.data_ptr(), qkv.data_ptr() + 2 * H, qkv.data_ptr() + 4 * H, 3 * H, o.data_ptr(),
               dout.data_ptr(), H, lse.data_ptr(), L.ptr(km), dqkv.data_ptr(), dqkv.data_ptr() + 2 * H,
               dqkv.data_ptr() + 4 * H, None, delta.data_ptr(), S, Ln, nh, hd, causal, scale, seg[0], seg[1], st)
        torch.cuda.synchronize()
        res[causal] = (o.view(S, Ln, H), dqkv.view(S, Ln, 3 * H))
    (o1, g1), (o3, g3) = res[1], res[L.CAUSAL_SKIP_MASKED_Q]
    valid = km.bool()
    assert torch.equal(o1[valid], o3[valid]), "valid rows must not change"
    dead_tiles = 0
    for s_ in range(S):
        for t0 in range(0, Ln, 64):
            if not bool(valid[s_, t0:t0 + 64].any()):
                dead_tiles += 1
                assert float(o3[s_, t0:t0 + 64].float().abs().max()) == 0.0
                assert float(g3[s_, t0:t0 + 64, :H].float().abs().max()) == 0.0          # dQ of a dead tile
    assert dead_tiles >= 7
    assert torch.equal(g1[valid], g3[valid]), "gradients of valid rows must be bit-identical"
    assert torch.equal(g1[:, :, H:], g3[:, :, H:]), "dK / dV must be bit-identical (the skipped q tiles contribute exact zeros)"
