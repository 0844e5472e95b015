// Internal C++ launcher declarations shared by the kernel translation units and capi.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/opadpo_hip.h"

typedef uint16_t bf16_t;

struct GemmNTArgs {
  const bf16_t* A1; const bf16_t* B1;   // [M,K1] lda1 ; [N,K1] ldb1
  const bf16_t* A2; const bf16_t* B2;   // LoRA tail: [M,*] lda2 ; [N,K2] ldb2 (may be null, K2 = 0)
  void* C;                              // [M,N] ldc, bf16 or fp32
  const void* R;                        // residual [M,N] ldr (nullable), bf16 or fp32 (r_f32)
  const bf16_t* bias;                   // [N] (nullable)
  int M, N, K1, K2;
  int lda1, ldb1, lda2, ldb2, ldc, ldr;
  int a2_group_n, a2_group_stride;
  int a1_group_n, a1_group_stride;      // grouped dgrad: A1 column offset = (n0 / a1_group_n) * a1_group_stride
  float alpha;
  int act;
  int out_f32;
  int r_f32;
  // fused rotary embedding (opadpo_gemm_nt_rope): columns [0, rope_cols) are heads of 128, rotated with the position of the
  // output row (row % rope_L, packed responses restart at rope_seg_prefix); tables [pos][64] fp32.  Null = off.
  const float* rope_cos = nullptr;
  const float* rope_sin = nullptr;
  int rope_L = 0, rope_cols = 0, rope_seg_prefix = 0, rope_seg_len = 0;
  // table-free form (round 3): per-row positions (int32 [M], e.g. the ragged pass's row_pos) and log2(theta); the epilogue computes the
  // angles itself: hardware sin / cos of the fractional revolution pos * theta^(-2i/128) / 2pi, evaluated per (row, frequency) - a function of
  // those two numbers only (an angle-addition recurrence along the rows was tried and dropped: it made a row depend on its tile offset)
  const int32_t* rope_pos = nullptr;
  float rope_l2theta = 0.f;
  int group_m = 8;                      // row tiles per group of the grouped tile order (256x256 4-wave kernel)
  int variant = -1;                     // kernel-variant override of this call (opadpo_ctx_set_flags); -1 = process default (opadpo_set_flags)
  int tile0 = 0;
  // quarter-tile tail launch of the 128x128 kernel (set by launch_gemm_nt only): block b computes quarter (b & 3) of the 256x256 tile
  // tile0 + (b >> 2) of the 4-wave kernel's grouped tile order - the full K range, so every output element keeps the summation order
  // it has inside a 256x256 tile (results do not depend on which tiles fall into the tail, i.e. on the row count of the batch)
  int quarter = 0;
  int swiglu_bwd_staged = 0;            // OPADPO_ACT_SWIGLU_BWD through the LDS-staged epilogue of rounds 3-4 (cross-check of the direct form)
  int store_nt = 0;                     // direct epilogue of the 256x256 4-wave kernels: non-temporal C stores (set by launch_gemm_nt)
  // K-FOLDED problem (set by launch_gemm_nt only; round 6): the product's K range is cut into N / b1_fold_n slices that run as column groups of ONE launch -
  // output columns [g * b1_fold_n, (g + 1) * b1_fold_n) are the partial product of slice g: A1 columns from g * a1_group_stride (the grouped-A1 rule),
  // B1 = rows 0 .. b1_fold_n - 1 of the caller's matrix read from column g * b1_fold_koff on.  gemm_nt_w4_kernel only.
  int b1_fold_n = 0, b1_fold_koff = 0;
  int xcd_cyclic = 0;                   // 256x256 4-wave kernels: 32-tile blocks dealt to the XCDs block-cyclically (xcd_remap_cyclic; set by launch_gemm_nt)
};

struct GemmTNArgs {
  const bf16_t* P; const bf16_t* Q;     // [M,N1] ldp ; [M,*] ldq
  float* C;                             // [N1,N2] ldc fp32, accumulated with atomics
  int M, N1, N2;
  int ldp, ldq, ldc;
  int q_group_n1, q_group_stride;       // Q column offset = (n1_0 / q_group_n1) * q_group_stride
  float alpha;
  int splits;                           // <=0: auto
  int use_tr = -1;                      // per-call override of the transposed-LDS-read switch; -1 = process default
};

struct AttnArgs {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;   // element (s, pos, head, d) at ptr[(s*L+pos)*ld + head*hd + d]
  bf16_t* o;                                            // same addressing with ldo
  float* lse;                                           // [S, nh, L]
  const uint8_t* key_mask;                              // [S, L] (nullable = all valid)
  int S, L, nh, hd;
  int ld, ldo;
  int causal;
  float scale;
  int seg_prefix, seg_len;                              // packed responses sharing a prefix (seg_len = 0: off)
  // backward only
  const bf16_t* dout;                                   // ldo addressing
  float* dq_acc;                                        // optional fp32 copy of dQ [S*L, nh*hd] (nullable)
  bf16_t* dq; bf16_t* dk; bf16_t* dv;                   // ld addressing (same as q / k / v)
  float* delta;                                         // [S, nh, L] scratch
  int use_tr;                                           // per-call override of the transposed-LDS-read switch; < 0 = process default
  // RAGGED rows (nullable; opadpo_ctx only): the padding rows are removed from the flat [rows, ld] buffers.  seq_meta[s * meta_stride
  // + 0] = first row of sequence s, [1 .. 1 + n_seg] = boundaries relative to it: b_0 = end of the prefix (= start of response 0),
  // b_a = start of response a, b_nseg = length of the sequence.  L = longest sequence (grid), lse / delta are [nh, rows_total],
  // key_mask is per row.
  const int32_t* seq_meta;
  int meta_stride, n_seg, rows_total;
  // backward only, nullable: per-row rotary positions (int32 [rows], the ragged pass's row_pos) + log2(theta).  When given, dQ and dK
  // leave the kernels ALREADY rotated back (the gradient of apply_rotary_pos_emb: g*cos - rot(g)*sin) - no separate pass over dq | dk
  const int32_t* rope_pos;
  float rope_l2theta;
};

// up to 8 gemm_tn problems with the same M run as ONE launch of the 256x256 kernel (tile lists concatenated)
struct GemmTNGroup {
  GemmTNArgs g[8];
  int n;
  int splits;
  int tile_end[8];      // cumulative 256x256 tile counts
  // deterministic flush (round 4): a run's partial tiles go to ws[(run * smax + segment) * 65536 floats] as plain stores and
  // gemm_tn_reduce_kernel adds the segments of every tile in a fixed order; ws = nullptr keeps the fp32-atomic flush
  float* ws = nullptr;
  int smax = 0;
};

void opadpo_set_flags_impl(int use_glds, int use_tr);
bool opadpo_flag_tr();
void opadpo_set_attn_dma(bool on);

hipError_t launch_gemm_nt(const GemmNTArgs& a, hipStream_t st);
hipError_t launch_gemm_tn(const GemmTNArgs& a, hipStream_t st);
// all problems must share M and be eligible for the 256x256 kernel (N1, N2 % 256 == 0, q_group_n1 % 256 == 0); otherwise the
// problems are launched one by one
void opadpo_set_sample_compact(int on);      // 1 / 0 force, -1 = OPADPO_SAMPLE_COMPACT (default 1)
hipError_t launch_gemm_tn_group(const GemmTNArgs* list, int n, hipStream_t st, void* workspace = nullptr, size_t workspace_bytes = 0, int* all_ordered = nullptr);
// bytes of workspace that make a grouped launch of these problems deterministic (0: the problems do not run on the 256x256 kernel)
size_t gemm_tn_group_workspace_bytes(const GemmTNArgs* list, int n);

hipError_t launch_attn_fwd(const AttnArgs& a, hipStream_t st);
hipError_t launch_attn_bwd(const AttnArgs& a, hipStream_t st);

hipError_t launch_rmsnorm_fwd(const void* x, int x_f32, const bf16_t* w, bf16_t* y, float* rstd, int rows, int H, float eps, hipStream_t st);
hipError_t launch_rmsnorm_sum_fwd(const void* resid, int resid_f32, const float* partials, int n_partials, size_t partial_stride, const bf16_t* w,
                                  float* x_out, bf16_t* y, float* rstd, int rows, int H, float eps, hipStream_t st);
hipError_t launch_gemm_nt_dec64(const GemmNTArgs& a, int mode, int splits, hipStream_t st);
int gemm_nt_dec64_splits(int N, int K, int splits);
hipError_t launch_rmsnorm_bwd(const bf16_t* dy, const void* x, int x_f32, const bf16_t* w, const float* rstd, const void* dres,
                              int dres_f32, float* dx_f32, bf16_t* dx_bf16, int rows, int H, hipStream_t st);
hipError_t launch_layernorm_fwd(const void* x, const bf16_t* w, const bf16_t* b, void* y, int rows, int H, float eps, hipStream_t st, int x_f32 = 0, int y_f32 = 0);
hipError_t launch_layernorm_bwd(const bf16_t* dy, const bf16_t* x, const bf16_t* w, const bf16_t* dres, bf16_t* dx, int rows, int H,
                                float eps, hipStream_t st);
hipError_t launch_act_fwd(const bf16_t* z, bf16_t* out, size_t n, int act, hipStream_t st);
hipError_t launch_act_bwd(const bf16_t* dout, const bf16_t* z, bf16_t* dz, size_t n, int act, hipStream_t st);
hipError_t launch_rope(bf16_t* qk, int ld, const float* cosb, const float* sinb, int rows, int L, int n_heads, int hd,
                       int inverse, const int32_t* pos_base, int seg_prefix, int seg_len, hipStream_t st, const int32_t* row_pos = nullptr,
                       float l2theta = 0.f);      // l2theta > 0 (with row_pos): table-free angles, the fused rotary epilogue's definition
hipError_t launch_silu_mul_fwd(const bf16_t* gu, bf16_t* act, int rows, int F, hipStream_t st);
hipError_t launch_silu_mul_bwd(const bf16_t* dact, const bf16_t* gu, bf16_t* dgu, int rows, int F, hipStream_t st);
hipError_t launch_embed_splice(const int32_t* ids, const uint8_t* text_mask, const bf16_t* embed, const bf16_t* feats,
                               const int32_t* feat_row, const uint8_t* image_mask, void* x, int x_f32, uint8_t* key_mask,
                               int S, int n_txt, int P, int H, int image_token, hipStream_t st);
hipError_t launch_im2col(const bf16_t* pixels, bf16_t* out, int B, int image_size, int patch, int kpad, hipStream_t st);
hipError_t launch_vision_embed(const bf16_t* patches, const bf16_t* cls, const bf16_t* pos, void* x, int B, int P, int h, hipStream_t st, int x_f32 = 0);
hipError_t launch_gather_rows(const bf16_t* src, int ld_src, const int32_t* rows_idx, bf16_t* dst, int n, int H, hipStream_t st);
hipError_t launch_scatter_add_rows_f32(const float* src, const int32_t* rows_idx, float* dst, int ld_dst, int n, int H, hipStream_t st);
hipError_t launch_scatter_rows(const bf16_t* src, const int32_t* rows_idx, bf16_t* dst, int ld_dst, int n, int H, hipStream_t st);
hipError_t launch_transpose(const bf16_t* in, bf16_t* out, int R, int C, hipStream_t st);
hipError_t launch_transpose_batched(const bf16_t* src, bf16_t* dst, const long long* jobs, int n_jobs, int max_tiles, hipStream_t st);
hipError_t launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t st);
hipError_t launch_bf16_to_f32(const bf16_t* in, float* out, size_t n, hipStream_t st);
hipError_t launch_f32_to_bf16_strided(const float* in, bf16_t* out, size_t rows, int C, int ld, hipStream_t st);

hipError_t launch_head_fwd(const float* logits, int ldl, const int32_t* labels, float inv_temp, float* logp, float* ent,
                           float* lse, int rows, int V, hipStream_t st);
hipError_t launch_head_bwd(const float* logits, int ldl, const int32_t* labels, const float* lse, const float* dlogp,
                           const float* ent, const float* dent,
                           float inv_temp, bf16_t* dz, int ldz, int rows, int V, hipStream_t st, int col0 = 0);
// chunked head (no [rows, vocab] buffer): fold the logits of vocabulary columns [col0, col0 + n) into the running (max, sum exp, sum z exp,
// label logit) of every row; finish -> log-prob, entropy, log-sum-exp
hipError_t launch_head_fwd_chunk(const float* logits, int ldl, const int32_t* labels, float inv_temp, int col0, int n, int first, float* m,
                                 float* s, float* t, float* zl, int rows, hipStream_t st);
hipError_t launch_head_fwd_finish(const int32_t* labels, const float* m, const float* s, const float* t, const float* zl, float* logp, float* ent,
                                  float* lse, int rows, hipStream_t st);

hipError_t launch_sumsq(const float* g, size_t n, float* out, hipStream_t st);
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, size_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt, const float* sumsq,
                        float max_norm, float grad_div, hipStream_t st);

hipError_t launch_sample(const float* logits, int ldl, int rows, int V, float temperature, int top_k, float top_p,
                         uint64_t seed, uint64_t step, const int32_t* step_ptr, uint8_t* finished, int pad_id, int eos_id,
                         int32_t* out, int32_t* history, hipStream_t st);
hipError_t launch_attn_decode(const bf16_t* q, const bf16_t* kc, const bf16_t* vc, bf16_t* o, const uint8_t* key_mask,
                              int B, int nh, int hd, int ctx, const int32_t* ctx_ptr, int max_ctx, int ldq, float scale,
                              void* workspace, size_t workspace_bytes, hipStream_t st);
size_t attn_decode_workspace_bytes(int B, int nh, int hd, int max_ctx);
hipError_t launch_attn_decode_fused(const bf16_t* qkv, int ld, const float* cosb, const float* sinb, bf16_t* kc, bf16_t* vc, bf16_t* o,
                                    const uint8_t* key_mask, int B, int nh, int hd, const int32_t* pos_ptr, int max_ctx, float scale,
                                    void* workspace, size_t workspace_bytes, hipStream_t st);
hipError_t launch_rope_kv_append(bf16_t* qkv, int ld, const float* cosb, const float* sinb, bf16_t* kc, bf16_t* vc, int B, int nh,
                                 int hd, const int32_t* pos_ptr, int max_ctx, hipStream_t st);
