// extern "C" boundary of libopadpo_hip.so (declared in include/opadpo_hip.h).
// Host-side validation only; every kernel launch is asynchronous on the caller's stream.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include "kernels.h"
#include <dlfcn.h>

namespace {
thread_local char g_err[512] = "";

int fail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
  return (int)e;
}
int bad(const char* where, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, what);
  return (int)hipErrorInvalidValue;
}
inline int done(hipError_t e, const char* where) { return e == hipSuccess ? 0 : fail(e, where); }
inline hipStream_t S(void* s) { return (hipStream_t)s; }
}  // namespace

extern "C" {

int opadpo_abi_version(void) { return OPADPO_ABI_VERSION; }
const char* opadpo_last_error(void) { return g_err; }
void opadpo_set_flags(int use_glds, int use_tr) { opadpo_set_flags_impl(use_glds, use_tr); }

int opadpo_gemm_nt(const uint16_t* A1, int lda1, const uint16_t* B1, int ldb1, int K1,
                   const uint16_t* A2, int lda2, const uint16_t* B2, int ldb2, int K2,
                   int a2_group_n, int a2_group_stride, int a1_group_n, int a1_group_stride,
                   void* C, int ldc, int out_f32, const void* R, int ldr, int res_f32, const uint16_t* bias,
                   int M, int N, float alpha, int act, void* stream) {
  if (M < 0 || N <= 0 || N % 128) return bad("opadpo_gemm_nt", "N must be a positive multiple of 128");
  if (K1 < 0 || K2 < 0 || K1 % 64 || K2 % 64 || K1 + K2 == 0) return bad("opadpo_gemm_nt", "K1/K2 must be multiples of 64");
  if ((K1 && (!A1 || !B1)) || (K2 && (!A2 || !B2)) || !C) return bad("opadpo_gemm_nt", "null operand");
  if (lda1 % 8 || ldb1 % 8 || (K2 && (lda2 % 8 || ldb2 % 8)) || ldc % 4 || (R && ldr % 4))
    return bad("opadpo_gemm_nt", "leading dimensions must keep 16-byte operand / 8-byte output alignment");
  if ((a2_group_n && a2_group_n % 128) || (a1_group_n && a1_group_n % 128))
    return bad("opadpo_gemm_nt", "group widths must be multiples of the 128-column tile");
  GemmNTArgs a;
  a.A1 = A1; a.B1 = B1; a.A2 = A2; a.B2 = B2; a.C = C; a.R = R; a.bias = bias;
  a.M = M; a.N = N; a.K1 = K1; a.K2 = K2;
  a.lda1 = lda1; a.ldb1 = ldb1; a.lda2 = lda2; a.ldb2 = ldb2; a.ldc = ldc; a.ldr = ldr;
  a.a2_group_n = a2_group_n; a.a2_group_stride = a2_group_stride;
  a.a1_group_n = a1_group_n; a.a1_group_stride = a1_group_stride;
  a.alpha = alpha; a.act = act; a.out_f32 = out_f32; a.r_f32 = res_f32;
  return done(launch_gemm_nt(a, S(stream)), "opadpo_gemm_nt");
}

int opadpo_gemm_nt_rope(const uint16_t* A1, int lda1, const uint16_t* B1, int ldb1, int K1,
                        const uint16_t* A2, int lda2, const uint16_t* B2, int ldb2, int K2, int a2_group_n, int a2_group_stride,
                        uint16_t* C, int ldc, int M, int N, const float* cos_tab, const float* sin_tab, int L, int rope_cols,
                        int seg_prefix, int seg_len, void* stream) {
  if (M < 0 || N <= 0 || N % 256) return bad("opadpo_gemm_nt_rope", "N must be a positive multiple of 256");
  if (K1 <= 0 || K2 < 0 || K1 % 64 || K2 % 64) return bad("opadpo_gemm_nt_rope", "K1/K2 must be multiples of 64");
  if (!A1 || !B1 || (K2 && (!A2 || !B2)) || !C || !cos_tab || !sin_tab) return bad("opadpo_gemm_nt_rope", "null operand");
  if (lda1 % 8 || ldb1 % 8 || (K2 && (lda2 % 8 || ldb2 % 8)) || ldc % 8) return bad("opadpo_gemm_nt_rope", "leading dimensions must keep 16-byte alignment");
  if (a2_group_n && a2_group_n % 256) return bad("opadpo_gemm_nt_rope", "group width must be a multiple of the 256-column tile");
  if (L < 4 || rope_cols < 0 || rope_cols > N || rope_cols % 128 || seg_prefix < 0 || seg_len < 0 || (seg_len > 0 && seg_len < 4))
    return bad("opadpo_gemm_nt_rope", "rope_cols must be whole heads of 128 within N; L >= 4; seg_len 0 or >= 4");
  GemmNTArgs a;
  a.A1 = A1; a.B1 = B1; a.A2 = A2; a.B2 = B2; a.C = C; a.R = nullptr; a.bias = nullptr;
  a.M = M; a.N = N; a.K1 = K1; a.K2 = K2;
  a.lda1 = lda1; a.ldb1 = ldb1; a.lda2 = lda2; a.ldb2 = ldb2; a.ldc = ldc; a.ldr = 0;
  a.a2_group_n = a2_group_n; a.a2_group_stride = a2_group_stride; a.a1_group_n = 0; a.a1_group_stride = 0;
  a.alpha = 1.0f; a.act = 0; a.out_f32 = 0; a.r_f32 = 0;
  a.rope_cos = cos_tab; a.rope_sin = sin_tab; a.rope_L = L; a.rope_cols = rope_cols; a.rope_seg_prefix = seg_prefix; a.rope_seg_len = seg_len;
  return done(launch_gemm_nt(a, S(stream)), "opadpo_gemm_nt_rope");
}

int opadpo_gemm_nt_rope_pos(const uint16_t* A1, int lda1, const uint16_t* B1, int ldb1, int K1,
                            const uint16_t* A2, int lda2, const uint16_t* B2, int ldb2, int K2, int a2_group_n, int a2_group_stride,
                            uint16_t* C, int ldc, int M, int N, const int32_t* row_pos, float theta, int rope_cols, void* stream) {
  if (M < 0 || N <= 0 || N % 256) return bad("opadpo_gemm_nt_rope_pos", "N must be a positive multiple of 256");
  if (K1 <= 0 || K2 < 0 || K1 % 64 || K2 % 64) return bad("opadpo_gemm_nt_rope_pos", "K1/K2 must be multiples of 64");
  if (!A1 || !B1 || (K2 && (!A2 || !B2)) || !C || !row_pos) return bad("opadpo_gemm_nt_rope_pos", "null operand");
  if (lda1 % 8 || ldb1 % 8 || (K2 && (lda2 % 8 || ldb2 % 8)) || ldc % 8) return bad("opadpo_gemm_nt_rope_pos", "leading dimensions must keep 16-byte alignment");
  if (a2_group_n && a2_group_n % 256) return bad("opadpo_gemm_nt_rope_pos", "group width must be a multiple of the 256-column tile");
  if (rope_cols < 0 || rope_cols > N || rope_cols % 128 || !(theta > 1.0f)) return bad("opadpo_gemm_nt_rope_pos", "rope_cols must be whole heads of 128 within N; theta > 1");
  GemmNTArgs a;
  a.A1 = A1; a.B1 = B1; a.A2 = A2; a.B2 = B2; a.C = C; a.R = nullptr; a.bias = nullptr;
  a.M = M; a.N = N; a.K1 = K1; a.K2 = K2;
  a.lda1 = lda1; a.ldb1 = ldb1; a.lda2 = lda2; a.ldb2 = ldb2; a.ldc = ldc; a.ldr = 0;
  a.a2_group_n = a2_group_n; a.a2_group_stride = a2_group_stride; a.a1_group_n = 0; a.a1_group_stride = 0;
  a.alpha = 1.0f; a.act = 0; a.out_f32 = 0; a.r_f32 = 0;
  a.rope_pos = row_pos; a.rope_l2theta = log2f(theta); a.rope_cols = rope_cols;
  return done(launch_gemm_nt(a, S(stream)), "opadpo_gemm_nt_rope_pos");
}

int opadpo_gemm_nt_decode(const uint16_t* A, int lda, const uint16_t* B, int ldb, int K, void* C, int ldc, int mode, int M, int N, int splits,
                          void* stream) {
  if (M < 0 || M > 64) return bad("opadpo_gemm_nt_decode", "M must be 0..64 (one token per sequence)");
  if (N <= 0 || N % 128 || K <= 0 || K % 64) return bad("opadpo_gemm_nt_decode", "N must be a multiple of 128, K of 64");
  if (!A || !B || !C || lda % 8 || ldb % 8 || ldc % 4 || mode < 0 || mode > 2) return bad("opadpo_gemm_nt_decode", "null operand, misaligned leading dimension or bad mode");
  GemmNTArgs a;
  a.A1 = A; a.B1 = B; a.A2 = nullptr; a.B2 = nullptr; a.C = C; a.R = nullptr; a.bias = nullptr;
  a.M = M; a.N = N; a.K1 = K; a.K2 = 0; a.lda1 = lda; a.ldb1 = ldb; a.lda2 = 0; a.ldb2 = 0; a.ldc = ldc; a.ldr = 0;
  a.a2_group_n = 0; a.a2_group_stride = 0; a.a1_group_n = 0; a.a1_group_stride = 0; a.alpha = 1.f; a.act = 0; a.out_f32 = mode == 1; a.r_f32 = 0;
  return done(launch_gemm_nt_dec64(a, mode, splits, S(stream)), "opadpo_gemm_nt_decode");
}
int opadpo_gemm_nt_decode_splits(int N, int K, int splits) { return (N > 0 && K >= 64) ? gemm_nt_dec64_splits(N, K, splits) : 1; }
int opadpo_rmsnorm_sum_fwd(const void* resid, int resid_f32, const float* partials, int n_partials, size_t partial_stride, const uint16_t* w,
                           float* x_out, uint16_t* y, float* rstd, int rows, int H, float eps, void* stream) {
  if (!resid || !w || !x_out) return bad("opadpo_rmsnorm_sum_fwd", "null operand");
  return done(launch_rmsnorm_sum_fwd(resid, resid_f32, partials, n_partials, partial_stride, w, x_out, y, rstd, rows, H, eps, S(stream)),
              "opadpo_rmsnorm_sum_fwd");
}
int opadpo_gemm_tn(const uint16_t* P, int ldp, const uint16_t* Q, int ldq, float* C, int ldc,
                   int M, int N1, int N2, int q_group_n1, int q_group_stride, float alpha, int splits,
                   void* stream) {
  if (N1 <= 0 || N2 <= 0 || N1 % 128 || N2 % 128) return bad("opadpo_gemm_tn", "N1/N2 must be positive multiples of 128");
  if (!P || !Q || !C || ldp % 8 || ldq % 8) return bad("opadpo_gemm_tn", "null operand or misaligned leading dimension");
  if (q_group_n1 && q_group_n1 % 128) return bad("opadpo_gemm_tn", "q_group_n1 must be a multiple of 128");
  GemmTNArgs a;
  a.P = P; a.Q = Q; a.C = C; a.M = M; a.N1 = N1; a.N2 = N2; a.ldp = ldp; a.ldq = ldq; a.ldc = ldc;
  a.q_group_n1 = q_group_n1; a.q_group_stride = q_group_stride; a.alpha = alpha; a.splits = splits;
  return done(launch_gemm_tn(a, S(stream)), "opadpo_gemm_tn");
}

int opadpo_gemm_tn_group(int n, const uint16_t* const* P, const int* ldp, const uint16_t* const* Q, const int* ldq, float* const* C, const int* ldc,
                         int M, const int* N1, const int* N2, const int* q_group_n1, const int* q_group_stride, float alpha, void* stream) {
  if (n < 0 || n > 8 || (n && (!P || !ldp || !Q || !ldq || !C || !ldc || !N1 || !N2))) return bad("opadpo_gemm_tn_group", "0..8 problems, non-null arrays");
  GemmTNArgs list[8];
  for (int i = 0; i < n; ++i) {
    if (N1[i] <= 0 || N2[i] <= 0 || N1[i] % 128 || N2[i] % 128) return bad("opadpo_gemm_tn_group", "N1/N2 must be positive multiples of 128");
    if (!P[i] || !Q[i] || !C[i] || ldp[i] % 8 || ldq[i] % 8) return bad("opadpo_gemm_tn_group", "null operand or misaligned leading dimension");
    GemmTNArgs& a = list[i];
    a.P = P[i]; a.Q = Q[i]; a.C = C[i]; a.M = M; a.N1 = N1[i]; a.N2 = N2[i]; a.ldp = ldp[i]; a.ldq = ldq[i]; a.ldc = ldc[i];
    a.q_group_n1 = q_group_n1 ? q_group_n1[i] : 0; a.q_group_stride = q_group_stride ? q_group_stride[i] : 0; a.alpha = alpha; a.splits = 0;
    if (a.q_group_n1 && a.q_group_n1 % 128) return bad("opadpo_gemm_tn_group", "q_group_n1 must be a multiple of 128");
  }
  return done(launch_gemm_tn_group(list, n, S(stream)), "opadpo_gemm_tn_group");
}

static int tn_group_list(const char* fn, GemmTNArgs* list, int n, const uint16_t* const* P, const int* ldp, const uint16_t* const* Q, const int* ldq, float* const* C,
                         const int* ldc, int M, const int* N1, const int* N2, const int* q_group_n1, const int* q_group_stride, float alpha, bool shapes_only) {
  if (n < 0 || n > 8 || (n && (!N1 || !N2 || (!shapes_only && (!P || !ldp || !Q || !ldq || !C || !ldc))))) return bad(fn, "0..8 problems, non-null arrays");
  for (int i = 0; i < n; ++i) {
    if (N1[i] <= 0 || N2[i] <= 0 || N1[i] % 128 || N2[i] % 128) return bad(fn, "N1/N2 must be positive multiples of 128");
    GemmTNArgs& a = list[i];
    a.M = M; a.N1 = N1[i]; a.N2 = N2[i];
    a.q_group_n1 = q_group_n1 ? q_group_n1[i] : 0; a.q_group_stride = q_group_stride ? q_group_stride[i] : 0; a.alpha = alpha; a.splits = 0;
    if (a.q_group_n1 && a.q_group_n1 % 128) return bad(fn, "q_group_n1 must be a multiple of 128");
    if (shapes_only) { a.P = a.Q = nullptr; a.C = nullptr; a.ldp = N1[i]; a.ldq = N2[i] * (a.q_group_n1 ? N1[i] / a.q_group_n1 : 1); a.ldc = N2[i]; continue; }
    if (!P[i] || !Q[i] || !C[i] || ldp[i] % 8 || ldq[i] % 8) return bad(fn, "null operand or misaligned leading dimension");
    a.P = P[i]; a.Q = Q[i]; a.C = C[i]; a.ldp = ldp[i]; a.ldq = ldq[i]; a.ldc = ldc[i];
  }
  return 0;
}

size_t opadpo_gemm_tn_group_workspace_bytes(int n, int M, const int* N1, const int* N2, const int* q_group_n1) {
  GemmTNArgs list[8];
  if (M <= 0 || tn_group_list("opadpo_gemm_tn_group_workspace_bytes", list, n, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, M, N1, N2, q_group_n1, nullptr, 1.f, true)) return 0;
  return gemm_tn_group_workspace_bytes(list, n);
}

int opadpo_gemm_tn_group_det(int n, const uint16_t* const* P, const int* ldp, const uint16_t* const* Q, const int* ldq, float* const* C, const int* ldc,
                             int M, const int* N1, const int* N2, const int* q_group_n1, const int* q_group_stride, float alpha,
                             void* workspace, size_t workspace_bytes, void* stream) {
  GemmTNArgs list[8];
  if (const int rc = tn_group_list("opadpo_gemm_tn_group_det", list, n, P, ldp, Q, ldq, C, ldc, M, N1, N2, q_group_n1, q_group_stride, alpha, false)) return rc;
  const size_t need = gemm_tn_group_workspace_bytes(list, n);
  if (need == 0) return bad("opadpo_gemm_tn_group_det", "these problems do not run on the 256x256 kernel (N1, N2, q_group_n1 % 256): no deterministic form");
  if (!workspace || workspace_bytes < need) return bad("opadpo_gemm_tn_group_det", "workspace smaller than opadpo_gemm_tn_group_workspace_bytes");
  return done(launch_gemm_tn_group(list, n, S(stream), workspace, workspace_bytes), "opadpo_gemm_tn_group_det");
}

int opadpo_attn_fwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int ld, uint16_t* o, int ldo,
                    float* lse, const uint8_t* key_mask, int S_, int L, int nh, int hd, int causal, float scale,
                    int seg_prefix, int seg_len, void* stream) {
  if (hd != 64 && hd != 128) return bad("opadpo_attn_fwd", "head_dim must be 64 or 128");
  // head_dim 128 runs the 32-rows-per-wave kernel, which writes O (also the zeros of an all-padding tile) as 16-byte pieces: ldo % 8
  if (!q || !k || !v || !o || ld % 8 || ldo % (hd == 128 ? 8 : 4)) return bad("opadpo_attn_fwd", "null operand or misaligned leading dimension (ld % 8, ldo % 8 at head_dim 128, ldo % 4 at 64)");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.key_mask = key_mask;
  a.S = S_; a.L = L; a.nh = nh; a.hd = hd; a.ld = ld; a.ldo = ldo; a.causal = causal; a.scale = scale;
  if (seg_len < 0 || seg_prefix < 0 || (seg_len > 0 && !causal)) return bad("opadpo_attn_fwd", "packed responses need causal attention and seg_prefix, seg_len >= 0");
  a.seg_prefix = seg_prefix; a.seg_len = seg_len; a.use_tr = -1;
  return done(launch_attn_fwd(a, S(stream)), "opadpo_attn_fwd");
}

int opadpo_attn_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int ld, const uint16_t* o,
                    const uint16_t* dout, int ldo, const float* lse, const uint8_t* key_mask,
                    uint16_t* dq, uint16_t* dk, uint16_t* dv, float* dq_f32, float* delta,
                    int S_, int L, int nh, int hd, int causal, float scale, int seg_prefix, int seg_len, void* stream) {
  if (hd != 64 && hd != 128) return bad("opadpo_attn_bwd", "head_dim must be 64 or 128");
  if (seg_len < 0 || seg_prefix < 0 || (seg_len > 0 && !causal)) return bad("opadpo_attn_bwd", "packed responses need causal attention and seg_prefix, seg_len >= 0");
  if (!q || !k || !v || !o || !dout || !lse || (!dq && !dq_f32) || !dk || !dv || !delta || ld % 8 || ldo % 8)
    return bad("opadpo_attn_bwd", "null operand or misaligned leading dimension");
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.q = q; a.k = k; a.v = v; a.o = (uint16_t*)o; a.lse = (float*)lse; a.key_mask = key_mask;
  a.S = S_; a.L = L; a.nh = nh; a.hd = hd; a.ld = ld; a.ldo = ldo; a.causal = causal; a.scale = scale;
  a.dout = dout; a.dq_acc = dq_f32; a.dq = dq; a.dk = dk; a.dv = dv; a.delta = delta;
  a.seg_prefix = seg_prefix; a.seg_len = seg_len; a.use_tr = -1;
  return done(launch_attn_bwd(a, S(stream)), "opadpo_attn_bwd");
}

int opadpo_rmsnorm_fwd(const void* x, int x_f32, const uint16_t* w, uint16_t* y, float* rstd, int rows, int H, float eps, void* stream) {
  return done(launch_rmsnorm_fwd(x, x_f32, w, y, rstd, rows, H, eps, S(stream)), "opadpo_rmsnorm_fwd");
}
int opadpo_rmsnorm_bwd(const uint16_t* dy, const void* x, int x_f32, const uint16_t* w, const float* rstd,
                       const void* dres, int dres_f32, float* dx_f32, uint16_t* dx_bf16, int rows, int H, void* stream) {
  if (!dx_f32 && !dx_bf16) return bad("opadpo_rmsnorm_bwd", "no output buffer");
  return done(launch_rmsnorm_bwd(dy, x, x_f32, w, rstd, dres, dres_f32, dx_f32, dx_bf16, rows, H, S(stream)), "opadpo_rmsnorm_bwd");
}
int opadpo_layernorm_fwd(const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* y, int rows, int H, float eps, void* stream) {
  return done(launch_layernorm_fwd(x, w, b, y, rows, H, eps, S(stream)), "opadpo_layernorm_fwd");
}
int opadpo_layernorm_fwd_f32(const float* x, const uint16_t* w, const uint16_t* b, void* y, int y_f32, int rows, int H, float eps, void* stream) {
  if (!x || !w || !b || !y) return bad("opadpo_layernorm_fwd_f32", "null operand");
  return done(launch_layernorm_fwd(x, w, b, y, rows, H, eps, S(stream), 1, y_f32 ? 1 : 0), "opadpo_layernorm_fwd_f32");
}
int opadpo_layernorm_bwd(const uint16_t* dy, const uint16_t* x, const uint16_t* w, const uint16_t* dres, uint16_t* dx, int rows, int H,
                         float eps, void* stream) {
  return done(launch_layernorm_bwd(dy, x, w, dres, dx, rows, H, eps, S(stream)), "opadpo_layernorm_bwd");
}
int opadpo_act_fwd(const uint16_t* z, uint16_t* out, size_t n, int act, void* stream) {
  return done(launch_act_fwd(z, out, n, act, S(stream)), "opadpo_act_fwd");
}
int opadpo_act_bwd(const uint16_t* dout, const uint16_t* z, uint16_t* dz, size_t n, int act, void* stream) {
  return done(launch_act_bwd(dout, z, dz, n, act, S(stream)), "opadpo_act_bwd");
}
int opadpo_rope(uint16_t* qk, int ld, const float* cos_tab, const float* sin_tab, int rows, int L, int n_heads, int hd,
                int inverse, const int32_t* pos_base, int seg_prefix, int seg_len, void* stream) {
  if (ld % 8) return bad("opadpo_rope", "misaligned leading dimension");
  return done(launch_rope(qk, ld, cos_tab, sin_tab, rows, L, n_heads, hd, inverse, pos_base, seg_prefix, seg_len, S(stream)), "opadpo_rope");
}
int opadpo_silu_mul_fwd(const uint16_t* gu, uint16_t* act, int rows, int F, void* stream) {
  return done(launch_silu_mul_fwd(gu, act, rows, F, S(stream)), "opadpo_silu_mul_fwd");
}
int opadpo_silu_mul_bwd(const uint16_t* dact, const uint16_t* gu, uint16_t* dgu, int rows, int F, void* stream) {
  return done(launch_silu_mul_bwd(dact, gu, dgu, rows, F, S(stream)), "opadpo_silu_mul_bwd");
}
int opadpo_embed_splice(const int32_t* ids, const uint8_t* text_mask, const uint16_t* embed, const uint16_t* feats,
                        const int32_t* feat_row, const uint8_t* image_mask, void* x, int x_f32, uint8_t* key_mask,
                        int S_, int n_txt, int P, int H, int image_token, void* stream) {
  if (!ids || !text_mask || !embed || !feats || !feat_row || !x || !key_mask) return bad("opadpo_embed_splice", "null operand");
  return done(launch_embed_splice(ids, text_mask, embed, feats, feat_row, image_mask, x, x_f32, key_mask, S_, n_txt, P, H,
                                  image_token, S(stream)), "opadpo_embed_splice");
}
int opadpo_im2col(const uint16_t* pixels, uint16_t* out, int B, int image_size, int patch, int kpad, void* stream) {
  if (image_size % patch || kpad < 3 * patch * patch) return bad("opadpo_im2col", "bad geometry");
  return done(launch_im2col(pixels, out, B, image_size, patch, kpad, S(stream)), "opadpo_im2col");
}
int opadpo_vision_embed(const uint16_t* patches, const uint16_t* cls, const uint16_t* pos, uint16_t* x, int B, int P, int h, void* stream) {
  return done(launch_vision_embed(patches, cls, pos, x, B, P, h, S(stream)), "opadpo_vision_embed");
}
int opadpo_vision_embed_f32(const uint16_t* patches, const uint16_t* cls, const uint16_t* pos, float* x, int B, int P, int h, void* stream) {
  return done(launch_vision_embed(patches, cls, pos, x, B, P, h, S(stream), 1), "opadpo_vision_embed_f32");
}
int opadpo_gather_rows(const uint16_t* src, int ld_src, const int32_t* rows_idx, uint16_t* dst, int n, int H, void* stream) {
  if (ld_src % 8) return bad("opadpo_gather_rows", "misaligned leading dimension");
  return done(launch_gather_rows(src, ld_src, rows_idx, dst, n, H, S(stream)), "opadpo_gather_rows");
}
int opadpo_scatter_rows(const uint16_t* src, const int32_t* rows_idx, uint16_t* dst, int ld_dst, int n, int H, void* stream) {
  if (ld_dst % 8) return bad("opadpo_scatter_rows", "misaligned leading dimension");
  return done(launch_scatter_rows(src, rows_idx, dst, ld_dst, n, H, S(stream)), "opadpo_scatter_rows");
}
int opadpo_scatter_add_rows_f32(const float* src, const int32_t* rows_idx, float* dst, int ld_dst, int n, int H, void* stream) {
  return done(launch_scatter_add_rows_f32(src, rows_idx, dst, ld_dst, n, H, S(stream)), "opadpo_scatter_add_rows_f32");
}
int opadpo_transpose(const uint16_t* in, uint16_t* out, int R, int C, void* stream) {
  return done(launch_transpose(in, out, R, C, S(stream)), "opadpo_transpose");
}
int opadpo_transpose_batched(const uint16_t* src, uint16_t* dst, const int64_t* jobs, int n_jobs, int max_tiles, void* stream) {
  if (n_jobs > 0 && (!src || !dst || !jobs)) return bad("opadpo_transpose_batched", "null operand");
  return done(launch_transpose_batched(src, dst, (const long long*)jobs, n_jobs, max_tiles, S(stream)), "opadpo_transpose_batched");
}
int opadpo_f32_to_bf16(const float* in, uint16_t* out, size_t n, void* stream) {
  return done(launch_f32_to_bf16(in, out, n, S(stream)), "opadpo_f32_to_bf16");
}
int opadpo_bf16_to_f32(const uint16_t* in, float* out, size_t n, void* stream) {
  if (n && ((uintptr_t)in % 16 || (uintptr_t)out % 16)) return bad("opadpo_bf16_to_f32", "buffers must be 16-byte aligned");
  return done(launch_bf16_to_f32(in, out, n, S(stream)), "opadpo_bf16_to_f32");
}
int opadpo_f32_to_bf16_strided(const float* in, uint16_t* out, size_t rows, int C, int ld, void* stream) {
  if (ld % 4) return bad("opadpo_f32_to_bf16_strided", "misaligned leading dimension");
  return done(launch_f32_to_bf16_strided(in, out, rows, C, ld, S(stream)), "opadpo_f32_to_bf16_strided");
}
int opadpo_head_fwd(const float* logits, int ldl, const int32_t* labels, float inv_temp, float* logp, float* ent,
                    float* lse, int rows, int V, void* stream) {
  if (!logits || !labels || !logp || !ent || !lse) return bad("opadpo_head_fwd", "null operand");
  return done(launch_head_fwd(logits, ldl, labels, inv_temp, logp, ent, lse, rows, V, S(stream)), "opadpo_head_fwd");
}
int opadpo_head_bwd(const float* logits, int ldl, const int32_t* labels, const float* lse, const float* dlogp,
                    const float* ent, const float* dent, float inv_temp, uint16_t* dz, int ldz, int rows, int V, void* stream) {
  if (!logits || !labels || !lse || !dlogp || !dz || (dent && !ent)) return bad("opadpo_head_bwd", "null operand");
  return done(launch_head_bwd(logits, ldl, labels, lse, dlogp, ent, dent, inv_temp, dz, ldz, rows, V, S(stream)), "opadpo_head_bwd");
}
int opadpo_sumsq(const float* g, size_t n, float* out, void* stream) {
  return done(launch_sumsq(g, n, out, S(stream)), "opadpo_sumsq");
}
int opadpo_adamw(float* p, const float* g, float* m, float* v, uint16_t* p_bf16, size_t n, double lr, double beta1,
                 double beta2, double eps, double weight_decay, int step, const float* sumsq, double max_norm,
                 double grad_div, void* stream) {
  if (step < 1) return bad("opadpo_adamw", "step is 1-based");
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  return done(launch_adamw(p, g, m, v, p_bf16, n, (float)lr, (float)beta1, (float)beta2, (float)eps, (float)weight_decay,
                           (float)bc1, (float)sqrt(bc2), sumsq, (float)max_norm, (float)grad_div, S(stream)), "opadpo_adamw");
}
int opadpo_attn_decode(const uint16_t* q, int ldq, const uint16_t* k_cache, const uint16_t* v_cache, uint16_t* o,
                       const uint8_t* key_mask, int B, int nh, int hd, int ctx, const int32_t* ctx_ptr, int max_ctx,
                       float scale, void* workspace, size_t workspace_bytes, void* stream) {
  if (ctx > max_ctx) return bad("opadpo_attn_decode", "ctx > max_ctx");
  if (ldq % 8) return bad("opadpo_attn_decode", "misaligned leading dimension");
  return done(launch_attn_decode(q, k_cache, v_cache, o, key_mask, B, nh, hd, ctx, ctx_ptr, max_ctx, ldq, scale, workspace,
                                 workspace_bytes, S(stream)),
              "opadpo_attn_decode");
}
size_t opadpo_attn_decode_workspace_bytes(int B, int nh, int hd, int max_ctx) {
  return (B > 0 && nh > 0) ? attn_decode_workspace_bytes(B, nh, hd, max_ctx) : 0;
}
int opadpo_attn_decode_fused(const uint16_t* qkv, int ld, const float* cos_tab, const float* sin_tab, uint16_t* k_cache, uint16_t* v_cache,
                             uint16_t* o, const uint8_t* key_mask, int B, int nh, int hd, const int32_t* pos_ptr, int max_ctx, float scale,
                             void* workspace, size_t workspace_bytes, void* stream) {
  if (!pos_ptr) return bad("opadpo_attn_decode_fused", "pos_ptr is required");
  if (ld % 8 || ld < 3 * nh * hd) return bad("opadpo_attn_decode_fused", "qkv rows must hold [q|k|v] with an aligned leading dimension");
  return done(launch_attn_decode_fused(qkv, ld, cos_tab, sin_tab, k_cache, v_cache, o, key_mask, B, nh, hd, pos_ptr, max_ctx, scale,
                                       workspace, workspace_bytes, S(stream)),
              "opadpo_attn_decode_fused");
}
int opadpo_rope_kv_append(uint16_t* qkv, int ld, const float* cos_tab, const float* sin_tab, uint16_t* k_cache, uint16_t* v_cache,
                          int B, int nh, int hd, const int32_t* pos_ptr, int max_ctx, void* stream) {
  return done(launch_rope_kv_append(qkv, ld, cos_tab, sin_tab, k_cache, v_cache, B, nh, hd, pos_ptr, max_ctx, S(stream)),
              "opadpo_rope_kv_append");
}
int opadpo_sample(const float* logits, int ldl, int rows, int V, float temperature, int top_k, float top_p,
                  uint64_t seed, uint64_t step, const int32_t* step_ptr, uint8_t* finished, int pad_id, int eos_id,
                  int32_t* out, int32_t* history, void* stream) {
  if (temperature <= 0.f) return bad("opadpo_sample", "temperature must be > 0");
  return done(launch_sample(logits, ldl, rows, V, temperature, top_k, top_p, seed, step, step_ptr, finished, pad_id, eos_id, out,
                            history, S(stream)),
              "opadpo_sample");
}

// ---- gradient exchange for binders that do not go through torch.distributed (opadpo_allreduce_grads) -----------------------
// The collective library is NOT linked: the entry points resolve ncclAllReduce / ncclReduceScatter / ncclAllGather at first use from whatever RCCL the
// process has loaded (dlsym on the global scope; librccl.so by name as the fall-back), so a host that already carries an RCCL - PyTorch ships its own -
// keeps exactly one copy.  comm is the caller's ncclComm_t; one process per GPU, every call asynchronous on `stream`.
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_reducescatter_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*nccl_allgather_fn)(const void*, void*, size_t, int, void*, hipStream_t);
static void* rccl_sym(const char* name) {
  void* f = dlsym(RTLD_DEFAULT, name);
  if (!f) {
    static void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (h) f = dlsym(h, name);
  }
  return f;
}
static int rccl_dtype(int dtype) { return dtype == 0 ? 7 /* ncclFloat32 */ : dtype == 1 ? 9 /* ncclBfloat16 */ : -1; }
int opadpo_allreduce_grads(void* nccl_comm, void* flat_grad, size_t count, int dtype, void* stream) {
  static nccl_allreduce_fn fn = (nccl_allreduce_fn)rccl_sym("ncclAllReduce");
  if (!fn) return bad("opadpo_allreduce_grads", "ncclAllReduce not found: no RCCL loaded in this process and librccl.so not on the loader path");
  if (!nccl_comm || !flat_grad || rccl_dtype(dtype) < 0) return bad("opadpo_allreduce_grads", "null communicator / buffer, or dtype not 0 (fp32) / 1 (bf16)");
  const int rc = fn(flat_grad, flat_grad, count, rccl_dtype(dtype), 0 /* ncclSum */, nccl_comm, S(stream));
  if (rc != 0) { snprintf(g_err, sizeof(g_err), "opadpo_allreduce_grads: ncclAllReduce returned %d", rc); return rc; }
  return 0;
}
int opadpo_reduce_scatter_grads(void* nccl_comm, const void* flat_grad, void* shard, size_t shard_count, int dtype, void* stream) {
  static nccl_reducescatter_fn fn = (nccl_reducescatter_fn)rccl_sym("ncclReduceScatter");
  if (!fn) return bad("opadpo_reduce_scatter_grads", "ncclReduceScatter not found: no RCCL loaded in this process and librccl.so not on the loader path");
  if (!nccl_comm || !flat_grad || !shard || rccl_dtype(dtype) < 0) return bad("opadpo_reduce_scatter_grads", "null communicator / buffer, or dtype not 0 / 1");
  const int rc = fn(flat_grad, shard, shard_count, rccl_dtype(dtype), 0, nccl_comm, S(stream));
  if (rc != 0) { snprintf(g_err, sizeof(g_err), "opadpo_reduce_scatter_grads: ncclReduceScatter returned %d", rc); return rc; }
  return 0;
}
int opadpo_all_gather_params(void* nccl_comm, const void* shard, void* flat, size_t shard_count, int dtype, void* stream) {
  static nccl_allgather_fn fn = (nccl_allgather_fn)rccl_sym("ncclAllGather");
  if (!fn) return bad("opadpo_all_gather_params", "ncclAllGather not found: no RCCL loaded in this process and librccl.so not on the loader path");
  if (!nccl_comm || !shard || !flat || rccl_dtype(dtype) < 0) return bad("opadpo_all_gather_params", "null communicator / buffer, or dtype not 0 / 1");
  const int rc = fn(shard, flat, shard_count, rccl_dtype(dtype), nccl_comm, S(stream));
  if (rc != 0) { snprintf(g_err, sizeof(g_err), "opadpo_all_gather_params: ncclAllGather returned %d", rc); return rc; }
  return 0;
}

}  // extern "C"
