"""ctypes binding of libopadpo_hip.so (C ABI: include/opadpo_hip.h).

The product path has NO fallback: if the library is missing or a call fails this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# OPADPO_LIB_PATH: another build of the same library (same-box A/B runs of two kernel versions); never a different implementation
LIB_PATH = os.environ.get("OPADPO_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libopadpo_hip.so")

ACT_NONE, ACT_QUICK_GELU, ACT_GELU, ACT_SWIGLU_PAIR, ACT_SWIGLU_BWD = 0, 1, 2, 3, 4
CAUSAL_SKIP_MASKED_Q = 3      # `causal` of opadpo_attn_fwd / _bwd: 1 = causal, | 2 = OPADPO_ATTN_SKIP_MASKED_Q (all-padding q tiles write zeros)

_p = C.c_void_p
_i = C.c_int
_f = C.c_float
_d = C.c_double
_sz = C.c_size_t
_u64 = C.c_uint64

# name -> argtypes (restype is int unless noted).  Must list EVERY symbol of include/opadpo_hip.h.
SIGNATURES = {
    "opadpo_gemm_nt": [_p, _i, _p, _i, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _p, _i, _i, _p, _i, _i, _p, _i, _i, _f, _i, _p],
    "opadpo_gemm_nt_rope": [_p, _i, _p, _i, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _p, _p, _i, _i, _i, _i, _p],
    "opadpo_gemm_nt_rope_pos": [_p, _i, _p, _i, _i, _p, _i, _p, _i, _i, _i, _i, _p, _i, _i, _i, _p, _f, _i, _p],
    "opadpo_gemm_nt_decode": [_p, _i, _p, _i, _i, _p, _i, _i, _i, _i, _i, _p],
    "opadpo_gemm_nt_decode_splits": [_i, _i, _i],
    "opadpo_rmsnorm_sum_fwd": [_p, _i, _p, _i, _sz, _p, _p, _p, _p, _i, _i, _f, _p],
    "opadpo_gemm_tn": [_p, _i, _p, _i, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p],
    "opadpo_gemm_tn_group": [_i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _f, _p],
    "opadpo_gemm_tn_group_det": [_i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _f, _p, _sz, _p],
    "opadpo_attn_fwd": [_p, _p, _p, _i, _p, _i, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p],
    "opadpo_attn_bwd": [_p, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _f, _i, _i, _p],
    "opadpo_rmsnorm_fwd": [_p, _i, _p, _p, _p, _i, _i, _f, _p],
    "opadpo_rmsnorm_bwd": [_p, _p, _i, _p, _p, _p, _i, _p, _p, _i, _i, _p],
    "opadpo_layernorm_fwd": [_p, _p, _p, _p, _i, _i, _f, _p],
    "opadpo_layernorm_fwd_f32": [_p, _p, _p, _p, _i, _i, _i, _f, _p],
    "opadpo_layernorm_bwd": [_p, _p, _p, _p, _p, _i, _i, _f, _p],
    "opadpo_act_fwd": [_p, _p, _sz, _i, _p],
    "opadpo_act_bwd": [_p, _p, _p, _sz, _i, _p],
    "opadpo_rope": [_p, _i, _p, _p, _i, _i, _i, _i, _i, _p, _i, _i, _p],
    "opadpo_silu_mul_fwd": [_p, _p, _i, _i, _p],
    "opadpo_silu_mul_bwd": [_p, _p, _p, _i, _i, _p],
    "opadpo_embed_splice": [_p, _p, _p, _p, _p, _p, _p, _i, _p, _i, _i, _i, _i, _i, _p],
    "opadpo_im2col": [_p, _p, _i, _i, _i, _i, _p],
    "opadpo_vision_embed": [_p, _p, _p, _p, _i, _i, _i, _p],
    "opadpo_vision_embed_f32": [_p, _p, _p, _p, _i, _i, _i, _p],
    "opadpo_gather_rows": [_p, _i, _p, _p, _i, _i, _p],
    "opadpo_scatter_rows": [_p, _p, _p, _i, _i, _i, _p],
    "opadpo_scatter_add_rows_f32": [_p, _p, _p, _i, _i, _i, _p],
    "opadpo_transpose": [_p, _p, _i, _i, _p],
    "opadpo_transpose_batched": [_p, _p, _p, _i, _i, _p],
    "opadpo_f32_to_bf16": [_p, _p, _sz, _p],
    "opadpo_bf16_to_f32": [_p, _p, _sz, _p],
    "opadpo_f32_to_bf16_strided": [_p, _p, _sz, _i, _i, _p],
    "opadpo_head_fwd": [_p, _i, _p, _f, _p, _p, _p, _i, _i, _p],
    "opadpo_head_bwd": [_p, _i, _p, _p, _p, _p, _p, _f, _p, _i, _i, _i, _p],
    "opadpo_sumsq": [_p, _sz, _p, _p],
    "opadpo_adamw": [_p, _p, _p, _p, _p, _sz, _d, _d, _d, _d, _d, _i, _p, _d, _d, _p],
    "opadpo_attn_decode": [_p, _i, _p, _p, _p, _p, _i, _i, _i, _i, _p, _i, _f, _p, _sz, _p],
    "opadpo_rope_kv_append": [_p, _i, _p, _p, _p, _p, _i, _i, _i, _p, _i, _p],
    "opadpo_attn_decode_fused": [_p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _p, _i, _f, _p, _sz, _p],
    "opadpo_sample": [_p, _i, _i, _i, _f, _i, _f, _u64, _u64, _p, _p, _i, _i, _p, _p, _p],
}
# context API (include/opadpo_hip.h, "Context API"): sequence-level entry points on an opaque opadpo_ctx*
SIGNATURES.update({
    "opadpo_ctx_create": [_p, _i, _p],
    "opadpo_ctx_set_allocator": [_p, _p, _p, _p],
    "opadpo_ctx_set_flags": [_p, _i, _i],
    "opadpo_ctx_trim": [_p],
    "opadpo_ctx_profile": [_p, _i],
    "opadpo_ctx_profile_read": [_p, _p, _p, _p],
    "opadpo_ctx_set_llm_weights": [_p, _p, _p, _p, _p, _p, _i],
    "opadpo_ctx_set_vision_weights": [_p, _p, _p, _i],
    "opadpo_ctx_set_rope_tables": [_p, _p, _p, _i],
    "opadpo_ctx_set_adapter": [_p, _i, _p, _p, _p],
    "opadpo_ctx_set_merged_adapter": [_p, _i, _p, _i, _i],
    "opadpo_vision_encode": [_p, _p, _i, _p, _p],
    "opadpo_seq_logprobs_fwd": [_p, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _p, _p, _p, _p, _p],
    "opadpo_seq_logprobs_bwd": [_p, _p, _p, _p, _p, _i, _i, _p],
    "opadpo_saved_release": [_p, _p],
    "opadpo_saved_residual": [_p, _p, _i, _p, _p, _p],
    "opadpo_decode_begin": [_p, _i, _p, _p, _p, _i, _i, _i, _f, _i, _f, _u64, _i, _i, _i, _p, _p],
    "opadpo_decode_step": [_p, _p],
    "opadpo_decode_run": [_p, _i, _i, _p],
    "opadpo_decode_all_finished": [_p, _p, _p],
    "opadpo_decode_end": [_p],
    "opadpo_allreduce_grads": [_p, _p, _sz, _i, _p],
    "opadpo_reduce_scatter_grads": [_p, _p, _p, _sz, _i, _p],
    "opadpo_all_gather_params": [_p, _p, _p, _sz, _i, _p],
})
OTHER_SYMBOLS = ["opadpo_abi_version", "opadpo_last_error", "opadpo_set_flags", "opadpo_attn_decode_workspace_bytes", "opadpo_gemm_tn_group_workspace_bytes",
                 "opadpo_ctx_destroy", "opadpo_ctx_last_error", "opadpo_ctx_bytes_peak", "opadpo_ctx_wgrad_deterministic"]

_lib: Optional[C.CDLL] = None


class OpadpoError(RuntimeError):
    pass


def load() -> C.CDLL:
    """dlopen the HIP library (build it first with opa-dpo_amd/build.py / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OpadpoError(
            f"{LIB_PATH} not found: build it with `python opa-dpo_amd/build.py` (there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is None:
            # OPADPO_LIB_PATH may point at an OLDER build of the library (same-box A/B of two kernel versions): entry points added since
            # are simply absent there - calling one raises at the call site; the shipped library must export everything
            if os.environ.get("OPADPO_LIB_PATH"):
                continue
            raise OpadpoError(f"{LIB_PATH} does not export {name}: stale build? run `python opa-dpo_amd/build.py`")
        fn.argtypes = argtypes
        fn.restype = _i
    lib.opadpo_abi_version.restype = _i
    lib.opadpo_last_error.restype = C.c_char_p
    lib.opadpo_set_flags.argtypes = [_i, _i]
    lib.opadpo_set_flags.restype = None
    lib.opadpo_attn_decode_workspace_bytes.argtypes = [_i, _i, _i, _i]
    lib.opadpo_attn_decode_workspace_bytes.restype = _sz
    if hasattr(lib, "opadpo_gemm_tn_group_workspace_bytes"):
        lib.opadpo_gemm_tn_group_workspace_bytes.argtypes = [_i, _i, _p, _p, _p]
        lib.opadpo_gemm_tn_group_workspace_bytes.restype = _sz
    lib.opadpo_ctx_destroy.argtypes = [_p]
    lib.opadpo_ctx_destroy.restype = None
    lib.opadpo_ctx_last_error.argtypes = [_p]
    lib.opadpo_ctx_last_error.restype = C.c_char_p
    lib.opadpo_ctx_bytes_peak.argtypes = [_p]
    lib.opadpo_ctx_bytes_peak.restype = _sz
    if hasattr(lib, "opadpo_ctx_wgrad_deterministic"):
        lib.opadpo_ctx_wgrad_deterministic.argtypes = [_p]
        lib.opadpo_ctx_wgrad_deterministic.restype = _i
    # the shipped library must be THIS ABI; an older build named by OPADPO_LIB_PATH (same-box A/B of two kernel versions) may be one behind
    if lib.opadpo_abi_version() != 2 and not (os.environ.get("OPADPO_LIB_PATH") and lib.opadpo_abi_version() == 1):
        raise OpadpoError(f"ABI version mismatch: library {lib.opadpo_abi_version()}, binding 2 (rebuild: python opa-dpo_amd/build.py)")
    _lib = lib
    g = os.environ.get("OPADPO_USE_GLDS")
    t = os.environ.get("OPADPO_USE_TR")
    if g is not None or t is not None:
        lib.opadpo_set_flags(int(g or 10), int(t or 1))
    return lib


def set_flags(use_glds=10, use_tr: bool = True) -> None:
    """Process-default kernel variants for the op-level entry points (a context carries its own: opadpo_ctx_set_flags).
    gemm_nt: 10 (default) auto = 4-wave 256x256 kernel (w4; streaming form w4s for plain products with >= 2 tiles per CU) from 320 blocks or when one round fills >= 88 % of the CUs,
    128x128 kernel otherwise (bias / activation problems: 8-wave kernel from 320 blocks); 4 the 128x128 kernel everywhere,
    17 the 8-wave 4-phase 256x256 kernel (p8), 31 the 4-wave kernel forced, 15 force the M <= 64 streaming kernel.  True -> default.
    use_tr: bit 0 = ds_read_b64_tr_b16 transposed LDS reads, bit 1 = attention forward through a direct-to-LDS double-buffered K/V
    ring (default: register-staged single buffer, 3 blocks per CU), bit 3 = 128x128 gemm_tn kernel instead of the default 256x256
    gemm_tn_w4_kernel, bit 4 = 16-row streaming kernel also for M <= 16 (default: gemm_nt_skinny8_kernel), bits 5-6 = kernel of
    opadpo_gemm_nt_decode: 0 / 3 the whole-line streaming kernel (gemm_nt_dec64x, the library's choice), 1 the LDS-ring kernel of rounds 2-4;
    bits 7-8 = weight rows per workgroup of the streaming kernel (0 by shape, 1 / 2 / 3 = 48 / 64 / 128), bit 9 = opadpo_sample runs its full
    vocabulary sweeps (the exactness yardstick; clear: OPADPO_SAMPLE_COMPACT decides, default compact), bit 10 = the streaming 256x256 GEMM walks
    its tile list on 8 workgroups (tests), bit 11 = the products of >= 128 K-tiles keep the default K-loop text (default since round 6: the DEEP text,
    bit-identical results), bit 12 = contiguous-chunk deal of the tile order to the XCDs (rounds 1-5; default since round 6: block-cyclic, bit-identical).
    NOTE: a context's use_tr (CtxEngine / opadpo_ctx_set_flags) is a different bit space above bit 4 - see include/opadpo_hip.h."""
    v = 10 if use_glds is True else int(use_glds)
    load().opadpo_set_flags(v, int(use_tr))


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def call(name: str, *args) -> None:
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise OpadpoError(f"{name} failed: {lib.opadpo_last_error().decode()}")


# ---------------------------------------------------------------------------------------------
# thin typed wrappers over torch tensors (device memory + stream plumbing only)
# ---------------------------------------------------------------------------------------------
def _chk(t: torch.Tensor, dtype, name: str):
    if t.dtype != dtype or not t.is_cuda:
        raise OpadpoError(f"{name}: expected cuda {dtype}, got {t.device} {t.dtype}")
    if t.dim() >= 2 and t.stride(-1) != 1:
        raise OpadpoError(f"{name}: innermost dimension must be contiguous")


# Optional live profiling of the dominant kernel (bench.py): when PROFILE is a list, every gemm_nt launch
# is bracketed by HIP events recorded on the stream the kernel is launched on.
PROFILE = None


GEMM_STREAM = 0x100
_stream_weights = False


class decode_schedule:
    """with decode_schedule(): every gemm_nt with M <= 64 takes the weight-streaming kernel (OPADPO_GEMM_STREAM)."""

    def __enter__(self):
        global _stream_weights
        self._prev, _stream_weights = _stream_weights, True

    def __exit__(self, *exc):
        global _stream_weights
        _stream_weights = self._prev


def gemm_nt(a1: torch.Tensor, b1: torch.Tensor, out: torch.Tensor, *, a2: Optional[torch.Tensor] = None,
            b2: Optional[torch.Tensor] = None, a2_group_n: int = 0, a2_group_stride: int = 0,
            a1_group_n: int = 0, a1_group_stride: int = 0, k1: Optional[int] = None, residual: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, alpha: float = 1.0,
            act: int = ACT_NONE) -> torch.Tensor:
    """out[M,N] = act(alpha*(a1 @ b1^T + a2[:, group] @ b2^T) + bias) + residual.  2-D row-major views
    with arbitrary row stride (leading dimension).  act = ACT_SWIGLU_PAIR: b1 rows per 128 are [64 gate | 64 up], out is [M, N/2] =
    silu(gate) * up; act = ACT_SWIGLU_BWD: the product is d_act [M,N], `residual` holds the stored [gate | up] ([M,2N]) and out is
    [d_gate | d_up] ([M,2N]) (include/opadpo_hip.h)."""
    _chk(a1, torch.bfloat16, "a1"); _chk(b1, torch.bfloat16, "b1")
    M = a1.shape[0]
    K1 = a1.shape[1] if k1 is None else k1       # k1: per-group K when a1 is the wide [M, G*K1] grouped operand
    N = b1.shape[0]
    assert b1.shape[1] == K1 and out.shape[0] == M and out.shape[1] == (N // 2 if act == ACT_SWIGLU_PAIR else 2 * N if act == ACT_SWIGLU_BWD else N)
    K2 = 0
    if a2 is not None:
        _chk(a2, torch.bfloat16, "a2"); _chk(b2, torch.bfloat16, "b2")
        K2 = b2.shape[1]
        assert b2.shape[0] == N and a2.shape[0] == M
    out_f32 = out.dtype == torch.float32
    assert out_f32 or out.dtype == torch.bfloat16
    res_f32 = residual is not None and residual.dtype == torch.float32
    assert residual is None or res_f32 or residual.dtype == torch.bfloat16
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("opadpo_gemm_nt", ptr(a1), a1.stride(0), ptr(b1), b1.stride(0), K1,
         ptr(a2), a2.stride(0) if a2 is not None else 0, ptr(b2), b2.stride(0) if b2 is not None else 0, K2,
         a2_group_n, a2_group_stride, a1_group_n, a1_group_stride, ptr(out), out.stride(0), int(out_f32),
         ptr(residual), residual.stride(0) if residual is not None else 0, int(res_f32), ptr(bias), M, N, float(alpha),
         act | (GEMM_STREAM if _stream_weights else 0), stream())
    if PROFILE is not None:
        e1.record()
        PROFILE.append((2.0 * M * N * (K1 + K2), e0, e1))
    return out


def gemm_nt_rope(a1: torch.Tensor, b1: torch.Tensor, out: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, L: int, rope_cols: int,
                 seg=(0, 0), *, a2: Optional[torch.Tensor] = None, b2: Optional[torch.Tensor] = None, a2_group_n: int = 0,
                 a2_group_stride: int = 0) -> torch.Tensor:
    """out[M,N] (bf16) = a1 @ b1^T (+ a2[:, group] @ b2^T), columns [0, rope_cols) rotated per head of 128 with the position of
    their row (row % L; seg = (prefix, seg_len): packed responses restart at the prefix) - opadpo_gemm_nt_rope."""
    _chk(a1, torch.bfloat16, "a1"); _chk(b1, torch.bfloat16, "b1"); _chk(out, torch.bfloat16, "out")
    _chk(cos, torch.float32, "cos"); _chk(sin, torch.float32, "sin")
    M, K1, N = a1.shape[0], a1.shape[1], b1.shape[0]
    assert b1.shape[1] == K1 and out.shape[0] == M and out.shape[1] == N and cos.shape[0] >= min(L, seg[0] + seg[1] if seg[1] else L)
    K2 = 0
    if a2 is not None:
        _chk(a2, torch.bfloat16, "a2"); _chk(b2, torch.bfloat16, "b2")
        K2 = b2.shape[1]
        assert b2.shape[0] == N and a2.shape[0] == M
    if PROFILE is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("opadpo_gemm_nt_rope", ptr(a1), a1.stride(0), ptr(b1), b1.stride(0), K1,
         ptr(a2), a2.stride(0) if a2 is not None else 0, ptr(b2), b2.stride(0) if b2 is not None else 0, K2,
         a2_group_n, a2_group_stride, ptr(out), out.stride(0), M, N, ptr(cos), ptr(sin), int(L), int(rope_cols), int(seg[0]), int(seg[1]), stream())
    if PROFILE is not None:
        e1.record()
        PROFILE.append((2.0 * M * N * (K1 + K2), e0, e1))
    return out


def gemm_tn(p: torch.Tensor, q: torch.Tensor, c: torch.Tensor, *, n2: Optional[int] = None, q_group_n1: int = 0,
            q_group_stride: int = 0, alpha: float = 1.0, splits: int = 0) -> torch.Tensor:
    """c[N1,N2] (fp32) += alpha * p[M,N1]^T @ q[M,N2]."""
    _chk(p, torch.bfloat16, "p"); _chk(q, torch.bfloat16, "q"); _chk(c, torch.float32, "c")
    M, N1 = p.shape
    N2 = c.shape[1] if n2 is None else n2
    assert c.shape[0] == N1 and q.shape[0] == M
    call("opadpo_gemm_tn", ptr(p), p.stride(0), ptr(q), q.stride(0), ptr(c), c.stride(0), M, N1, N2,
         q_group_n1, q_group_stride, float(alpha), splits, stream())
    return c
