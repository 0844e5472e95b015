/* libopadpo_hip.so — C ABI of the MI355X (gfx950) OPA-DPO hot path.
 *
 * The reference (zhyang2226/OPA-DPO) has no FFI: its hot path is the Python call
 * `self.base_model(**inputs)` inside AutoregressivePolicy.forward
 * (opadpo/dpo_models/rl_models.py:114-120) plus loss.backward()/optimizer.step()
 * (opadpo/dpo_models/rl_trainer.py:155-175), all of which lands in third-party CUDA wheels
 * (torch/cuBLAS, flash-attn, peft, bitsandbytes).  This header is the seam a maintainer binds
 * instead (ctypes stub: INTEGRATION.md).  Every entry point:
 *   - is plain `extern "C"`: raw device pointers + sizes, no torch / C++ types;
 *   - is asynchronous on the `hipStream_t` passed as `void* stream` (torch's current stream);
 *   - borrows all memory (the caller — torch-ROCm — owns and keeps it alive);
 *   - returns 0 on success, a non-zero hipError_t otherwise (`opadpo_last_error()` = text);
 *     it never aborts and never falls back to a CPU path.
 * bf16 tensors are `uint16_t` bit patterns (torch.bfloat16).  "ld*" = leading dimension in
 * elements.  Token ids are int32, pad id 0, image token -200 (utils/constants.py:28).
 */
#ifndef OPADPO_HIP_H
#define OPADPO_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPADPO_ABI_VERSION 2      /* 2 (round 5): + opadpo_ctx_wgrad_deterministic, opadpo_allreduce_grads / _reduce_scatter_grads / _all_gather_params; opadpo_gemm_tn_group_workspace_bytes is 0 for MIXED lists too; context flag bits 13 / 14 (residual adds / SwiGLU backward as their own passes) and bit 6 = the STAGED SwiGLU-backward epilogue */
#define OPADPO_ACT_NONE 0
#define OPADPO_ACT_QUICK_GELU 1 /* CLIP MLP  (transformers activations.py quick_gelu) */
#define OPADPO_ACT_GELU 2       /* mm_projector mlp2x_gelu (erf GELU) */
#define OPADPO_ACT_SWIGLU_PAIR 3 /* gemm_nt only: B rows arranged per 128 as [64 gate | 64 up]; C gets N/2 columns
                                 * silu(gate) * up (SwiGLU of modeling_llama.LlamaMLP fused into the gate|up projection; used
                                 * for the merged, no-grad reference pass and, with OPADPO_GEMM_STREAM, for decode with a
                                 * merged adapter).  Needs N % 256 == 0, bf16 C, no bias / residual / LoRA tail, alpha = 1. */
#define OPADPO_ATTN_SKIP_MASKED_Q 2 /* OR into `causal` of opadpo_attn_fwd / _bwd: a tile of 64 query positions that are ALL masked as keys
                                    * (the trailing padding of a right-padded response) writes zeros (forward: O, backward: dQ) and is left
                                    * out of the dK / dV accumulation - exact for the LLM, where such rows are padding whose outputs nobody
                                    * reads and whose output gradient is zero; needs key_mask */
#define OPADPO_ACT_SWIGLU_BWD 4 /* gemm_nt only (N a multiple of 256, bf16 out, alpha 1, no bias): the product is d_act [M,N], R (bf16, ldr) holds the
                                * stored pre-activations [gate | up] = [M,2N] and C (ldc) receives [d_gate | d_up] = [M,2N]: opadpo_silu_mul_bwd applied
                                * in the epilogue on the bf16-rounded d_act tile (bit-identical to the two-call form, d_act never written) */
#define OPADPO_GEMM_STREAM 0x100 /* OR into `act`: M <= 64 (one token per sequence, KV-cache decode) -> weight-streaming
                                  * schedule (one workgroup per 16/32 weight rows, K split over its waves); from 8 token rows a
                                  * problem with nothing fused (no LoRA tail, bias, residual or scale; plain or SWIGLU_PAIR store) runs on
                                  * the whole-line kernel of opadpo_gemm_nt_decode, bit-identical to that entry point */

int opadpo_abi_version(void);
const char* opadpo_last_error(void);
/* process-default kernel-variant switches of the op-level entry points (a context carries its own: opadpo_ctx_set_flags):
 * use_glds = gemm_nt variant (10 = auto (default), 4 = 128x128 kernel, 17 = 8-wave 256x256 kernel, 31 = 4-wave 256x256 kernel forced,
 * 15 = M <= 64 weight-streaming kernel); use_tr bit 0 = ds_read_b64_tr_b16 transposed LDS reads in attention / gemm_tn, bit 1 =
 * attention forward through a direct-to-LDS double-buffered K/V ring (default: register-staged single buffer, which keeps 3 blocks
 * per CU), bit 3 = 128x128 gemm_tn kernel instead of the default 256x256 one, bit 4 = 16-row weight-streaming decode GEMM also for
 * M <= 16 (default there: the whole-cache-line 8-row form), bits 5-6 = kernel behind opadpo_gemm_nt_decode (0 = the library's choice,
 * 1 = the LDS-ring kernel of rounds 2-4, 3 = the whole-line streaming kernel gemm_nt_dec64x = the library's choice; A/B runs and tests), bits 7-8 = weight rows per workgroup of that kernel (0 = by shape, 1 / 2 / 3 =
 * 48 / 64 / 128 rows; tests), bit 9 = opadpo_sample runs its full vocabulary sweeps instead of the one-wave tail on the kept tokens (identical draws; the
 * exactness test's yardstick; a process switch, so eager and graph-captured launches always agree), bit 10 = the streaming 256x256 GEMM runs on 8
 * workgroups instead of one per CU (tests: long tile walks per workgroup on small problems; results are bit-identical for any workgroup count),
 * bit 11 = the gemm_nt products of >= 128 K-tiles (down projection, the dgrads) keep the default K-loop text of the 256x256 kernel; clear (default since round 6):
 * they run its DEEP text - another placement of the same loads, barriers and MFMAs, bit-identical results, +2-3 % there (OPADPO_W4_DEEP=0: process default off).
 * (Round 5 used bit 11 for an experimental 64-rows-per-wave attention forward, measured slower and removed.)
 * bit 12 = the 256x256 gemm_nt kernels deal their tile order to the XCDs in contiguous chunks (rounds 1-5); clear (default since round 6): 32-tile blocks dealt
 * block-cyclically, so that the eight XCDs of a round share one row group's A panels through the Infinity Cache - bit-identical results, -1.0 % per training step
 * (OPADPO_XCD_CYCLIC=0: process default off). */
void opadpo_set_flags(int use_glds, int use_tr);

/* ---- Linear layers: base GEMM with the LoRA branch fused by K-concatenation ----------------
 * C[M,N] = act(alpha * (A1[M,K1].B1[N,K1]^T + A2[M,K2].B2[N,K2]^T) + bias) + R
 * replaces nn.Linear + peft.tuners.lora.Linear.forward (y = xW^T + (alpha/r)(xA^T)B^T) reached
 * from rl_models.py:120, and their dgrad in backward (rl_trainer.py:162).
 * A2's column offset for output column n0 is (n0 / a2_group_n) * a2_group_stride (fused q|k|v and
 * gate|up projections); pass a2_group_n = 0 for a single group.  A1 can be grouped the same way
 * (a1_group_n / a1_group_stride: block-diagonal dT = dY_g . B_g of the fused projections in one launch).  N % 128 == 0, K1 % 64 == 0,
 * K2 % 64 == 0; M arbitrary.  out_f32: C is float32 instead of bf16; res_f32: R is float32 (the LLM
 * residual stream is kept in fp32 so bf16 rounding does not accumulate over 2*n_layers additions). */
int opadpo_gemm_nt(const uint16_t* A1, int lda1, const uint16_t* B1, int ldb1, int K1,
                   const uint16_t* A2, int lda2, const uint16_t* B2, int ldb2, int K2,
                   int a2_group_n, int a2_group_stride, int a1_group_n, int a1_group_stride,
                   void* C, int ldc, int out_f32, const void* R, int ldr, int res_f32, const uint16_t* bias,
                   int M, int N, float alpha, int act, void* stream);

/* The fused q|k|v projection with the rotary embedding applied in its epilogue (modeling_llama.LlamaAttention: q_proj / k_proj /
 * v_proj followed by apply_rotary_pos_emb on q and k): C[M,N] bf16 = A1 B1^T (+ A2[:, group] B2^T), then columns [0, rope_cols)
 * - heads of 128 - are rotated with the position of their row: pos = row % L, and with seg_len > 0 every response of a packed
 * row restarts at seg_prefix (same convention as opadpo_rope / opadpo_attn_fwd); cos_tab / sin_tab [>= L][64] fp32.  The
 * rotation reads the bf16-rounded projection, as opadpo_rope does on the stored tensor.  N % 256 == 0; alpha = 1, no bias,
 * residual or activation. */
int opadpo_gemm_nt_rope(const uint16_t* A1, int lda1, const uint16_t* B1, int ldb1, int K1,
                        const uint16_t* A2, int lda2, const uint16_t* B2, int ldb2, int K2, int a2_group_n, int a2_group_stride,
                        uint16_t* C, int ldc, int M, int N, const float* cos_tab, const float* sin_tab, int L, int rope_cols,
                        int seg_prefix, int seg_len, void* stream);

/* The same projection with a TABLE-FREE rotary epilogue (round 3; what the context's ragged passes run): the position of every output
 * row comes from row_pos [M] int32 (device), the angles are computed in the epilogue - v_sin / v_cos of the fractional revolution
 * pos * theta^(-2i/128) / 2pi per (row, frequency), a function of those two numbers only (a row's bits do not depend on its place in the
 * batch).  No cos / sin traffic, no position arithmetic: ~3 us per 256x256 block instead of 7.7; angles within 2e-4 rad of HF's fp32
 * tables (the bf16 rounding of the result is 4e-3 relative). */
int opadpo_gemm_nt_rope_pos(const uint16_t* A1, int lda1, const uint16_t* B1, int ldb1, int K1,
                            const uint16_t* A2, int lda2, const uint16_t* B2, int ldb2, int K2, int a2_group_n, int a2_group_stride,
                            uint16_t* C, int ldc, int M, int N, const int32_t* row_pos, float theta, int rope_cols, void* stream);

/* Decode projection for up to 64 tokens (rollout at 9..64 sequences per device, online_generator.py:292-309): C = A[M,K] . B[N,K]^T,
 * no bias / residual / LoRA tail (adapter-free or merged adapter).  A weight stream: 48 / 64 / 128 weight rows (by shape) x <= 64 tokens x
 * one K-slice per workgroup, the weights loaded global -> registers in whole 128-byte lines, the activations shared through LDS in
 * 256-deep chunks by two loader waves (gemm_nt_dec64x_kernel, round 4; opadpo_set_flags use_tr bits 5-6 = 1 selects the LDS-ring kernel
 * of rounds 2-4).  mode 0: bf16 C[M,N]; mode 1: fp32 partial tiles
 * C[splits][M,N] (ldc = row stride of one slice) - K is split over `splits` workgroups (<= 0: chosen by the library, query it with
 * opadpo_gemm_nt_decode_splits) and the consumer adds the slices (opadpo_rmsnorm_sum_fwd); mode 2: OPADPO_ACT_SWIGLU_PAIR weight
 * layout -> bf16 C[M, N/2] = silu(gate) * up.  M <= 64, N % 128 == 0, K % 64 == 0. */
int opadpo_gemm_nt_decode(const uint16_t* A, int lda, const uint16_t* B, int ldb, int K, void* C, int ldc, int mode, int M, int N, int splits,
                          void* stream);
int opadpo_gemm_nt_decode_splits(int N, int K, int splits);
/* x_out = resid + sum_s partials[s] (fp32 [rows,H]; resid fp32 or bf16), y (nullable) = RMSNorm(x_out) * w in bf16, rstd (nullable):
 * the residual add + LlamaRMSNorm behind a K-split decode projection; fixed summation order (deterministic). */
int opadpo_rmsnorm_sum_fwd(const void* resid, int resid_f32, const float* partials, int n_partials, size_t partial_stride, const uint16_t* w,
                           float* x_out, uint16_t* y, float* rstd, int rows, int H, float eps, void* stream);

/* LoRA weight gradients: C[N1,N2] (fp32) += alpha * sum_m P[m,N1] * Q[m,N2]   (dB = dY^T t,
 * dA = dT^T x; autograd of peft lora_A / lora_B, rl_trainer.py:162).  Q column offset for output
 * row n1 is (n1 / q_group_n1) * q_group_stride.  N1 % 128 == 0, N2 % 128 == 0. splits<=0: auto. */
int opadpo_gemm_tn(const uint16_t* P, int ldp, const uint16_t* Q, int ldq, float* C, int ldc,
                   int M, int N1, int N2, int q_group_n1, int q_group_stride, float alpha, int splits,
                   void* stream);
/* Up to 8 such products that share M (the 8 LoRA wgrads of one decoder layer) as ONE launch: their 256x256 tile lists are
 * concatenated, so a workgroup's run is long and the fp32-atomic flushes are rare (an r-wide output launched alone is cut into 16
 * K-chunks whose flush takes as long as their K-loop).  Host arrays of n entries; problems that do not fit the 256x256 kernel
 * (N1, N2, q_group_n1 % 256) make the call fall back to n single launches. */
int opadpo_gemm_tn_group(int n, const uint16_t* const* P, const int* ldp, const uint16_t* const* Q, const int* ldq, float* const* C, const int* ldc,
                         int M, const int* N1, const int* N2, const int* q_group_n1, const int* q_group_stride, float alpha, void* stream);
/* DETERMINISTIC form (round 4; what opadpo_seq_logprobs_bwd uses): a workgroup's partial tiles go to `workspace` as plain stores and a
 * second launch adds the partial tiles of every output tile in a fixed order - no fp32 atomics, C bit-reproducible run to run (accelerate /
 * PEFT gradients of the reference are deterministic too unless cuBLAS split-K is in play; dpo_trainer.py:847-849 accumulates them in place).
 * The problems must all run on the 256x256 kernel (N1, N2, q_group_n1 % 256 == 0); `_workspace_bytes` returns 0 otherwise.  The workspace
 * is scratch: free again when the call's kernels have run (stream order). */
size_t opadpo_gemm_tn_group_workspace_bytes(int n, int M, const int* N1, const int* N2, const int* q_group_n1);
int opadpo_gemm_tn_group_det(int n, const uint16_t* const* P, const int* ldp, const uint16_t* const* Q, const int* ldq, float* const* C, const int* ldc,
                             int M, const int* N1, const int* N2, const int* q_group_n1, const int* q_group_stride, float alpha,
                             void* workspace, size_t workspace_bytes, void* stream);

/* ---- attention (flash-attn 2.5.3 LlamaFlashAttention2 / CLIP eager attention) -------------------
 * q,k,v: element (s,pos,head,d) at ptr[(s*L+pos)*ld + head*hd + d]; o/dout with ldo.
 * key_mask [S,L] bytes (NULL = all keys valid); causal: key pos <= query pos.  hd in {64,128}.
 * lse [S,nh,L] fp32 (may be NULL in forward when no backward follows).
 * seg_len > 0 (needs causal): every row is [prefix of seg_prefix positions | response 0 | response 1 | ...], each response
 * seg_len long; a response attends the prefix and itself only — the K responses of a DPO sample (rl_models.py:95-112 stacks
 * them as K separate sequences) share ONE pass over the image + query prefix.  seg_len = 0: plain sequences. */
int opadpo_attn_fwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int ld, uint16_t* o, int ldo,
                    float* lse, const uint8_t* key_mask, int S, int L, int nh, int hd, int causal, float scale,
                    int seg_prefix, int seg_len, void* stream);
/* dq/dk/dv: bf16 with the q/k/v addressing (ld); dq_f32: optional fp32 copy of dQ [S*L, nh*hd]
 * (NULL in the product path); delta: fp32 scratch [S,nh,L].  Atomic-free (two passes: dK/dV, dQ). */
int opadpo_attn_bwd(const uint16_t* q, const uint16_t* k, const uint16_t* v, int ld, const uint16_t* o,
                    const uint16_t* dout, int ldo, const float* lse, const uint8_t* key_mask,
                    uint16_t* dq, uint16_t* dk, uint16_t* dv, float* dq_f32, float* delta,
                    int S, int L, int nh, int hd, int causal, float scale, int seg_prefix, int seg_len, void* stream);

/* ---- norms / rotary / SwiGLU (transformers modeling_llama.py / modeling_clip.py) -------------- */
/* x: bf16 or (x_f32) float32 rows; y bf16; rstd fp32 [rows] (nullable). */
int opadpo_rmsnorm_fwd(const void* x, int x_f32, const uint16_t* w, uint16_t* y, float* rstd, int rows, int H, float eps, void* stream);
/* dx = rmsnorm'(dy) + dres; written as float32 (dx_f32, nullable) and/or bf16 (dx_bf16, nullable). */
int opadpo_rmsnorm_bwd(const uint16_t* dy, const void* x, int x_f32, const uint16_t* w, const float* rstd,
                       const void* dres, int dres_f32, float* dx_f32, uint16_t* dx_bf16, int rows, int H, void* stream);
int opadpo_layernorm_fwd(const uint16_t* x, const uint16_t* w, const uint16_t* b, uint16_t* y, int rows, int H, float eps, void* stream);
/* fp32 input (the CLIP tower's residual stream is fp32 in the DPO path, round 4); y bf16 (y_f32 = 0: the operand of the next GEMM) or fp32
 * (y_f32 = 1: the pre-LayerNorm, whose output IS the residual stream) */
int opadpo_layernorm_fwd_f32(const float* x, const uint16_t* w, const uint16_t* b, void* y, int y_f32, int rows, int H, float eps, void* stream);
/* CLIP / projector training path (OPA LoRA-SFT stage, opadpo/opa_train.py: the vision tower and mm_projector carry trainable
 * LoRA there): LayerNorm backward w.r.t. x (affine parameters frozen; mean / rstd recomputed; dres nullable = residual-path
 * gradient, added), and the activation applied / differentiated on a stored PRE-activation tensor (act = OPADPO_ACT_*). */
int opadpo_layernorm_bwd(const uint16_t* dy, const uint16_t* x, const uint16_t* w, const uint16_t* dres, uint16_t* dx, int rows, int H,
                         float eps, void* stream);
int opadpo_act_fwd(const uint16_t* z, uint16_t* out, size_t n, int act, void* stream);
int opadpo_act_bwd(const uint16_t* dout, const uint16_t* z, uint16_t* dz, size_t n, int act, void* stream);
/* in-place half-split rotary on n_heads heads starting at column 0 of qk (row r has position
 * pos_base[0] + r % L; pos_base is a device int32 or NULL = 0 — device-resident so that a captured decode step can be
 * replayed); cos/sin: fp32 [max_pos, hd/2]; inverse=1 applies the transposed rotation (gradient).  seg_len > 0: packed
 * responses (see opadpo_attn_fwd) — positions >= seg_prefix + seg_len wrap back so every response starts at seg_prefix. */
int opadpo_rope(uint16_t* qk, int ld, const float* cos_tab, const float* sin_tab, int rows, int L, int n_heads, int hd,
                int inverse, const int32_t* pos_base, int seg_prefix, int seg_len, void* stream);
int opadpo_silu_mul_fwd(const uint16_t* gu, uint16_t* act, int rows, int F, void* stream);   /* gu = [gate | up] */
int opadpo_silu_mul_bwd(const uint16_t* dact, const uint16_t* gu, uint16_t* dgu, int rows, int F, void* stream);

/* ---- embedding gather + multimodal splice (LLaVA prepare_inputs_labels_for_multimodal) ------------
 * ids/text_mask [S,n_txt]; each row holds exactly one image_token, replaced by the P rows
 * feats[feat_row[s]] -> x [S, n_txt+P-1, H] (bf16, or float32 when x_f32), key_mask [S, n_txt+P-1].
 * image_mask [S,P] or NULL. */
int opadpo_embed_splice(const int32_t* ids, const uint8_t* text_mask, const uint16_t* embed, const uint16_t* feats,
                        const int32_t* feat_row, const uint8_t* image_mask, void* x, int x_f32, uint8_t* key_mask,
                        int S, int n_txt, int P, int H, int image_token, void* stream);

/* ---- CLIP patch embedding (Conv2d k=s=patch as im2col + gemm_nt) ---------------------------------- */
int opadpo_im2col(const uint16_t* pixels, uint16_t* out, int B, int image_size, int patch, int kpad, void* stream);
int opadpo_vision_embed(const uint16_t* patches, const uint16_t* cls, const uint16_t* pos, uint16_t* x, int B, int P, int h, void* stream);
int opadpo_vision_embed_f32(const uint16_t* patches, const uint16_t* cls, const uint16_t* pos, float* x, int B, int P, int h, void* stream);   /* fp32 x */

/* ---- data movement helpers ---------------------------------------------------------------------- */
int opadpo_gather_rows(const uint16_t* src, int ld_src, const int32_t* rows_idx, uint16_t* dst, int n, int H, void* stream);
int opadpo_scatter_rows(const uint16_t* src, const int32_t* rows_idx, uint16_t* dst, int ld_dst, int n, int H, void* stream);
/* dst[rows_idx[r], :H] += src[r, :H] (fp32; duplicate indices accumulate) */
int opadpo_scatter_add_rows_f32(const float* src, const int32_t* rows_idx, float* dst, int ld_dst, int n, int H, void* stream);
int opadpo_transpose(const uint16_t* in, uint16_t* out, int R, int C, void* stream);
/* n_jobs transposes in one launch: jobs[j] = {src offset, dst offset, rows, cols} (elements, device memory, int64); max_tiles =
 * max over jobs of ceil(rows/64)*ceil(cols/64).  Used for the K-major copies of all LoRA blocks after an optimizer step. */
int opadpo_transpose_batched(const uint16_t* src, uint16_t* dst, const int64_t* jobs, int n_jobs, int max_tiles, void* stream);
int opadpo_f32_to_bf16(const float* in, uint16_t* out, size_t n, void* stream);
/* bf16 -> fp32 widening copy (exact); 16-byte aligned buffers.  Gradient slices received in bf16 from the ZeRO-1 reduce-scatter. */
int opadpo_bf16_to_f32(const uint16_t* in, float* out, size_t n, void* stream);
int opadpo_f32_to_bf16_strided(const float* in, uint16_t* out, size_t rows, int C, int ld, void* stream);

/* ---- head: compute_logprobs (utils/common_utils.py:112-118) + entropy (rl_models.py:128,132) -------
 * logits fp32 [rows,V] (ldl), labels int32 (0 = pad -> logp = -0.0, ent = 0), z = logits*inv_temp. */
int opadpo_head_fwd(const float* logits, int ldl, const int32_t* labels, float inv_temp, float* logp, float* ent,
                    float* lse, int rows, int V, void* stream);
/* dz = [dlogp * (onehot(label) - p) - dent * p * (log p + ent)] * inv_temp as bf16 [rows,V] (ldz), p = softmax(z).  ent (the
 * row entropies from head_fwd) and dent (their gradient) are nullable: only the OPA-SFT entropy regulariser
 * (opa_models/opa_trainer.py:64-90) differentiates the entropy; the DPO path passes NULL. */
int opadpo_head_bwd(const float* logits, int ldl, const int32_t* labels, const float* lse, const float* dlogp,
                    const float* ent, const float* dent, float inv_temp, uint16_t* dz, int ldz, int rows, int V, void* stream);

/* ---- clip_grad_norm_ + AdamW (rl_trainer.py:164-175; utils/trainer_utils.py:35) ------------------- */
int opadpo_sumsq(const float* g, size_t n, float* out, void* stream);  /* out[0] += sum g^2 */
/* p,m,v fp32 [n]; g fp32; p_bf16 (nullable) refreshed; step is 1-based; sumsq (nullable) device
 * scalar holding ||g||^2; effective gradient = g * grad_div * min(1, max_norm/(||g||*grad_div+1e-6)). */
int opadpo_adamw(float* p, const float* g, float* m, float* v, uint16_t* p_bf16, size_t n, double lr, double beta1,
                 double beta2, double eps, double weight_decay, int step, const float* sumsq, double max_norm,
                 double grad_div, void* stream);

/* ---- gradient exchange without torch.distributed (SURVEY.md section 8b "opadpo_allreduce_grads"; reference: accelerate / DDP behind
 * rl_trainer.py:155-175, which never syncs - Quirk Q1 - while north_star mandates the exchange) ------------------------------------------------
 * For a binder that owns its ncclComm_t (one process per GPU, ncclCommInitRank with the RCCL already loaded in the process; the library does
 * NOT link RCCL - it resolves the three collectives with dlsym at first use, librccl.so by name as the fall-back).  dtype 0 = fp32, 1 = bf16 (the
 * wire format of the Python host's ZeRO-1 path); SUM over the ranks, in place for the all-reduce; divide by the world size in opadpo_adamw's
 * grad_div.  ZeRO-1 = opadpo_reduce_scatter_grads (flat [world * shard_count] -> this rank's shard) -> opadpo_sumsq + scalar all-reduce through
 * opadpo_allreduce_grads(count = 1) -> opadpo_adamw on the shard -> opadpo_all_gather_params.  Asynchronous on `stream`; return code = the
 * RCCL result (0 = ncclSuccess), text through opadpo_last_error.  The Python host of this repository keeps using torch.distributed (same RCCL). */
int opadpo_allreduce_grads(void* nccl_comm, void* flat_grad, size_t count, int dtype, void* stream);
int opadpo_reduce_scatter_grads(void* nccl_comm, const void* flat_grad, void* shard, size_t shard_count, int dtype, void* stream);
int opadpo_all_gather_params(void* nccl_comm, const void* shard, void* flat, size_t shard_count, int dtype, void* stream);

/* ---- rollout (online_generator.py:292-323: policy.generate(do_sample=True, top_k, top_p, temperature)) ----
 * KV cache layout is head-major: k_cache / v_cache [B, nh, max_ctx, hd] (one (sequence, head) = one contiguous stream).
 * attn_decode: one query token per sequence, q row b at q + b*ldq ([nh*hd] used), o [B, nh*hd];
 * keys 0..ctx-1 valid where key_mask[b*max_ctx + j] != 0; ctx_ptr (device int32, nullable) overrides ctx with
 * ctx_ptr[0] + 1 (position of the newest key) for graph replay.  workspace (nullable): device scratch of at least
 * opadpo_attn_decode_workspace_bytes(B, nh, hd, max_ctx) bytes; with it the key range is split over several blocks
 * (+ one merge launch) where that fills the 256 CUs better: up to 256 (sequence, head) pairs as many ranges as fit ONE round of 16-wave
 * workgroups, two ranges of 4-wave workgroups for 257-767 pairs, and from 768 pairs two ranges only where they cut the rounds of the 1024
 * resident 4-wave workgroups by a fifth or more (max_ctx >= 512; the byte count returned is 0 when the geometry has one range).
 * Cache slots >= ctx are never read. */
int opadpo_attn_decode(const uint16_t* q, int ldq, const uint16_t* k_cache, const uint16_t* v_cache, uint16_t* o,
                       const uint8_t* key_mask, int B, int nh, int hd, int ctx, const int32_t* ctx_ptr, int max_ctx,
                       float scale, void* workspace, size_t workspace_bytes, void* stream);
size_t opadpo_attn_decode_workspace_bytes(int B, int nh, int hd, int max_ctx);
/* decode step: RoPE at position pos_ptr[0] (device int32) on the q and k thirds of qkv rows [q|k|v] (ld elements apart);
 * q is rotated in place, the rotated k and the v are appended to the caches at [b, h, pos, :]. */
int opadpo_rope_kv_append(uint16_t* qkv, int ld, const float* cos_tab, const float* sin_tab, uint16_t* k_cache, uint16_t* v_cache,
                          int B, int nh, int hd, const int32_t* pos_ptr, int max_ctx, void* stream);
/* rope_kv_append + attn_decode in ONE launch (decode step of the generator): qkv rows [q|k|v] of the new token as the
 * projection wrote them (not modified); q and k are rotated at position pos_ptr[0] inside the kernel, k and v appended to the
 * caches at [b, h, pos, :], attention over keys 0..pos (key_mask as above).  Same rounding points as the two calls. */
int opadpo_attn_decode_fused(const uint16_t* qkv, int ld, const float* cos_tab, const float* sin_tab, uint16_t* k_cache, uint16_t* v_cache,
                             uint16_t* o, const uint8_t* key_mask, int B, int nh, int hd, const int32_t* pos_ptr, int max_ctx, float scale,
                             void* workspace, size_t workspace_bytes, void* stream);
/* temperature -> top-k -> top-p -> multinomial (HF logits processors order); one draw per row from
 * a counter-based generator keyed on (seed, step, row).  finished rows emit pad_id; a row that draws eos_id (>= 0) is
 * marked finished.  step_ptr (device int32, nullable) overrides `step`; history (nullable) [steps, rows] receives the
 * token at row step. */
int opadpo_sample(const float* logits, int ldl, int rows, int V, float temperature, int top_k, float top_p,
                  uint64_t seed, uint64_t step, const int32_t* step_ptr, uint8_t* finished, int pad_id, int eos_id,
                  int32_t* out, int32_t* history, void* stream);

/* =====================================================================================================================
 * Context API: the SEQUENCE-LEVEL entry points.
 *
 * The reference's seam is one Python call per pass - `self.base_model(**inputs)` (opadpo/dpo_models/rl_models.py:114-120), its
 * backward `accelerator.backward(loss)` (rl_trainer.py:162) and `policy.generate(...)` (rl_models.py:166-183,
 * opadpo/generator_models/online_generator.py:292-309).  An `opadpo_ctx` puts the whole schedule of such a pass below the C ABI:
 *   - one context per process / GPU (one process per GPU, run/train_opa_dpo.sh:96-99); not re-entrant per context;
 *   - weights, adapters, token ids, masks and outputs are BORROWED raw device pointers (the caller keeps them alive);
 *   - the workspace, the saved activations of a training forward and the KV cache of a rollout are OWNED by the context
 *     (hipMalloc by default, or the caller's stream-ordered allocator - opadpo_ctx_set_allocator - so that a torch host keeps ONE
 *     memory pool); opadpo_ctx_destroy frees everything;
 *   - every call takes the caller's hipStream_t and is asynchronous on it (opadpo_decode_begin synchronises twice for two tiny
 *     host->device counters); int return code, 0 = ok, text through opadpo_ctx_last_error(ctx); never aborts, never a CPU path;
 *   - kernel-variant switches are per context (opadpo_ctx_set_flags), not process-global.
 * ===================================================================================================================== */
#define OPADPO_MAX_ADAPTERS 8
#define OPADPO_IMAGE_TOKEN (-200) /* utils/constants.py:28 IMAGE_TOKEN_INDEX */

typedef struct opadpo_ctx opadpo_ctx;
typedef struct opadpo_saved opadpo_saved; /* activations of one training forward, owned by the context */

typedef struct opadpo_dims {
  int hidden, n_layers, n_heads, head_dim, ffn, vocab; /* Llama-2 decoder (7B: 4096, 32, 32, 128, 11008, 32000) */
  float rms_eps, rope_theta;
  int v_hidden, v_used_layers, v_heads, v_ffn, image_size, patch; /* CLIP-ViT-L/14-336: 1024, 23 (hidden_states[-2]), 16, 4096, 336, 14 */
  float v_eps;
  int lora_r; /* 256 in the OPA-DPO recipe */
  float lora_alpha;
} opadpo_dims;

/* One decoder layer, bf16, row-major [out, in]: q|k|v and gate|up fused along the output dimension.  *_t = K-major transposed
 * copies ([in, out]) used by the dgrad GEMMs (NULL for inference-only contexts). */
typedef struct opadpo_layer_weights {
  const uint16_t *wqkv, *wo, *wgu, *wd, *ln1, *ln2;
  const uint16_t *wqkv_t, *wo_t, *wgu_t, *wd_t;
} opadpo_layer_weights;

typedef struct opadpo_vision_layer_weights {
  const uint16_t *ln1_w, *ln1_b, *ln2_w, *ln2_b, *wqkv, *bqkv, *wo, *bo, *fc1, *b1, *fc2, *b2;
} opadpo_vision_layer_weights;

typedef struct opadpo_vision_weights {
  const uint16_t *patch_w;              /* [v_hidden, ceil64(3*patch*patch)] zero-padded conv weight */
  const uint16_t *cls, *pos, *pre_ln_w, *pre_ln_b;
  const uint16_t *proj0, *proj0_b, *proj2, *proj2_b; /* mm_projector mlp2x_gelu */
} opadpo_vision_weights;

/* stream-ordered allocator hooks: alloc(bytes, stream, user) -> device pointer (NULL on failure); free(ptr, user) */
typedef void* (*opadpo_alloc_fn)(size_t bytes, void* stream, void* user);
typedef void (*opadpo_free_fn)(void* ptr, void* user);

int opadpo_ctx_create(const opadpo_dims* dims, int device, opadpo_ctx** out);
void opadpo_ctx_destroy(opadpo_ctx* ctx);
const char* opadpo_ctx_last_error(const opadpo_ctx* ctx);
int opadpo_ctx_set_allocator(opadpo_ctx* ctx, opadpo_alloc_fn alloc, opadpo_free_fn free_fn, void* user);
/* gemm_variant as in opadpo_set_flags, for this context only; -1 = process default.  use_tr is a SEPARATE bit space from the process flags: only
 * bits 0-4 mean what they mean in opadpo_set_flags (passed on to the kernels of this context); bits 5-14 are the context switches listed here
 * (process bits 5-10 - decode GEMM kernel, sampler sweep, 8-workgroup walk - have no per-context form and are NOT read from this value).
 * Context bits: use_tr bit 5 = keep the
 * 16/32-row streaming GEMMs for rollouts of 33..64 sequences (default there: the LDS-ring decode GEMM, opadpo_gemm_nt_decode);
 * bit 6 = SwiGLU backward in the LDS-STAGED epilogue of the down projection's dgrad (the form of rounds 3-4, which measured 0.35 % slower per step
 * than its own launch; the default since round 5 is the direct-epilogue form, see bit 14); bit 7 = top decoder layer on every row (default on ragged rows: its o-projection and MLP run only
 * on the rows the head reads - the last prefix row and the response rows -, forward and backward; exact, nothing else reads the rest);
 * bit 8 = the 16-rows-per-wave attention forward and dQ kernels (default at head_dim 128: 32 rows per wave on v_mfma_f32_32x32x16_bf16);
 * bit 9 / bit 10 = force / forbid the CHUNKED head (lm_head + online log-sum-exp + label gather + entropy over 4096 vocabulary columns at
 * a time, logits recomputed per chunk in the backward: no [rows, vocab] buffer; default: chunked when the fp32 logits of the batch shape S*K*T reach 4 GiB);
 * bit 11 = rotary embedding as its own in-place kernel (default on ragged rows: inside the q|k|v projection's epilogue, opadpo_gemm_nt_rope_pos);
 * bit 12 = LoRA wgrads flushed with fp32 atomics (default: partial tiles to a workspace + ordered reduce, bit-reproducible gradients).
 *   PRECONDITION of the default: every wgrad of a layer runs on the 256x256 kernel, i.e. lora_r % 256 == 0 (the shipped DPO recipe: r = 256).
 *   With r = 64 / 128 (the reference's online_generation / opa_train defaults) the wgrads run on the 128x128 kernel and flush with fp32
 *   atomics whatever this bit says: results correct to fp32 summation order, not bit-reproducible - opadpo_ctx_wgrad_deterministic tells.
 * bit 13 = residual adds of the full-sequence passes deferred to the RMSNorm that follows (rmsnorm_sum_fwd: x = res + y, 14 B per element; the form of
 *   rounds 2-4).  Default since round 5: the o / down projections add their fp32 residual rows in the 256x256 kernels' DIRECT epilogue (rows requested
 *   one row block ahead of the accumulator read-out) and the norm pass reads 6 B per element - the same fp32 addition of the same operands, so h / x
 *   keep their bits; -1.0 % per step.  OPADPO_FUSE_RESID=0 makes the deferred form the process default (A/B runs).
 * bit 14 = SwiGLU backward as its own kernel (silu_mul_bwd; rounds 1-4).  Default since round 5: inside the direct epilogue of the down projection's
 *   dgrad (OPADPO_ACT_SWIGLU_BWD, gate / up operands requested one row block ahead; bit-identical to the two-kernel form, d_act never reaches HBM;
 *   -0.8 % per step); bit 6 selects the LDS-staged form of that epilogue (rounds 3-4; cross-check).  OPADPO_FUSE_SWIGLU_BWD=0: process default off. */
int opadpo_ctx_set_flags(opadpo_ctx* ctx, int gemm_variant, int use_tr);
/* how the LAST opadpo_seq_logprobs_bwd of this context flushed its LoRA wgrads: 1 = ordered reduce (bit-reproducible), 0 = fp32 atomics
 * (bit 12 set, or lora_r % 256 != 0), -1 = no backward yet */
int opadpo_ctx_wgrad_deterministic(const opadpo_ctx* ctx);
/* return cached arenas and the workspace to the allocator, and forget the largest-arena-so-far hints (the next pass sizes its arena for its own batch) */
int opadpo_ctx_trim(opadpo_ctx* ctx);
size_t opadpo_ctx_bytes_peak(const opadpo_ctx* ctx);
/* live measurement of the dominant kernel: while enabled every gemm_nt launch of the context is bracketed by HIP events on the launch
 * stream; _read waits for them and returns (and clears) the sum of algorithmic FLOPs 2*M*N*(K1+K2), the sum of launch durations in ms
 * and the launch count */
int opadpo_ctx_profile(opadpo_ctx* ctx, int enable);
int opadpo_ctx_profile_read(opadpo_ctx* ctx, double* flops, double* ms, int64_t* launches);

/* make_models (dpo_trainer.py:958-1038): ONE frozen base shared by every adapter */
int opadpo_ctx_set_llm_weights(opadpo_ctx* ctx, const uint16_t* embed, const uint16_t* final_norm, const uint16_t* lm_head,
                               const uint16_t* lm_head_t, const opadpo_layer_weights* layers, int n_layers);
int opadpo_ctx_set_vision_weights(opadpo_ctx* ctx, const opadpo_vision_weights* w, const opadpo_vision_layer_weights* layers, int n_layers);
/* optional: rotary tables [n_pos][head_dim/2] fp32 from the host (default: the context computes HF Llama's own) */
int opadpo_ctx_set_rope_tables(opadpo_ctx* ctx, const float* cos_tab, const float* sin_tab, int n_pos);
/* set_adapter (rl_models.py:84-85): adapter `id` = flat bf16 LoRA buffer (layer-major; per layer a_qkv [3r,H] | b_qkv [3H,r] | a_o |
 * b_o | a_gu [2r,H] | b_gu [2F,r] | a_d [r,F] | b_d [H,r]); work_t = its K-major copy, grad = flat fp32 gradient (both NULL for a
 * frozen adapter); work = NULL selects the bare base model */
int opadpo_ctx_set_adapter(opadpo_ctx* ctx, int id, const uint16_t* work, const uint16_t* work_t, float* grad);
/* a FROZEN adapter folded into its own copy of the projections (PEFT merge); swiglu_pair: wgu rows per 128 = [64 gate | 64 up] */
int opadpo_ctx_set_merged_adapter(opadpo_ctx* ctx, int id, const opadpo_layer_weights* merged, int n_layers, int swiglu_pair);

/* get_vision_tower() + mm_projector: pixels [B,3,S,S] bf16 -> feats [B, 576, hidden] bf16 */
int opadpo_vision_encode(opadpo_ctx* ctx, const uint16_t* pixels, int B, uint16_t* feats, void* stream);

/* base_model(**inputs) + logits[:, -T-1:-1] / temperature + compute_logprobs + entropy (rl_models.py:114-132), for S rows
 * [query | response_0 | .. | response_{K-1}] of n_txt ids each (one OPADPO_IMAGE_TOKEN per row; K > 1: the responses share one pass
 * over the image + query prefix).  feat_row[s] = which image's features row s uses; image_mask [S,576] (nullable) = CoPO
 * 'attention' key mask.  logp / ent: [K*S*T] fp32 in the reference's stacking order [k][s][t].  train = 1 keeps the activations
 * (handle in *saved); train = 0 releases them (*saved = NULL).
 * row_plan (HOST memory, nullable): int32 [S][K+1] = per row the number of LEADING masked query positions to drop and the valid
 * length of every response (positions behind it are padding).  When given, the pass runs on RAGGED rows: padding positions are not
 * rows of any GEMM / norm / SwiGLU / RoPE / attention tile (the reference computes them and throws the result away: 24 % of the rows
 * of a synthetic seq512 pair), outputs on valid tokens are unchanged, pad cells still read -0.0 / 0.  NULL = the padded layout.  Any S
 * and K (the per-sequence geometry travels to the device in as many 960-int kernel-argument blocks as it needs). */
int opadpo_seq_logprobs_fwd(opadpo_ctx* ctx, int adapter_id, const int32_t* ids, const uint8_t* text_mask, const int32_t* feat_row,
                            const uint8_t* image_mask, const uint16_t* feats, int S, int n_txt, int T, int K, float temperature, int train,
                            float* logp, float* ent, opadpo_saved** saved, const int32_t* row_plan, void* stream);
/* accelerator.backward(loss): accumulates d loss / d LoRA into the adapter's flat fp32 grad buffer for decoder layers
 * layer_hi .. layer_lo (top-down).  The first call of a backward starts at n_layers - 1 (it runs the head backward from dlogp
 * [K*S*T] and the optional entropy gradient dent); later calls continue below - a data-parallel host launches the exchange of a
 * finished bucket of layers between two calls.  d_feats (nullable, fp32 [n_images,576,hidden], accumulated): gradient w.r.t.
 * the image features (OPA LoRA-SFT stage), written when layer_lo == 0. */
int opadpo_seq_logprobs_bwd(opadpo_ctx* ctx, opadpo_saved* saved, const float* dlogp, const float* dent, float* d_feats, int layer_hi,
                            int layer_lo, void* stream);
int opadpo_saved_release(opadpo_ctx* ctx, opadpo_saved* saved);
/* diagnostics (parity tests: the per-layer drift curve of the residual stream): copies the fp32 residual stream ENTERING decoder
 * layer `layer` of a training forward, [rows, hidden], into dst (device memory, rows*hidden floats) and returns the row count in
 * *rows (ragged passes: the compact valid rows, sequence by sequence; padded: S * L).  dst = NULL only queries *rows. */
int opadpo_saved_residual(opadpo_ctx* ctx, const opadpo_saved* saved, int layer, float* dst, int* rows, void* stream);

/* policy.generate(do_sample=True, ...) (online_generator.py:292-309): prefill of B left-padded queries [B,Q] + token 0; the KV
 * cache ([n_layers, B, heads, Q+575+max_new_tokens, head_dim] x2) lives in the context.  history [max_new_tokens, B] int32
 * (caller-owned) receives the tokens; finished rows emit pad_id.  suppress_eos: benchmarks (never sample EOS). */
int opadpo_decode_begin(opadpo_ctx* ctx, int adapter_id, const int32_t* ids, const uint8_t* text_mask, const uint16_t* feats, int B, int Q,
                        int max_new_tokens, float temperature, int top_k, float top_p, uint64_t seed, int eos_id, int pad_id, int suppress_eos,
                        int32_t* history, void* stream);
/* one more token for every row: all launches of a step, asynchronous */
int opadpo_decode_step(opadpo_ctx* ctx, void* stream);
/* n_steps more tokens.  use_graph = 0: the launches of every step are issued from the library's own loop (no host work between
 * them; measured faster than graph replay on MI355X: 3.45 vs 3.72 ms per step at 7B / 4 sequences); use_graph = 1: one step is
 * captured into a hipGraph once and replayed per token (same tokens, bit for bit) */
int opadpo_decode_run(opadpo_ctx* ctx, int n_steps, int use_graph, void* stream);
/* synchronous: *all_finished = every row has emitted EOS */
int opadpo_decode_all_finished(opadpo_ctx* ctx, int* all_finished, void* stream);
int opadpo_decode_end(opadpo_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* OPADPO_HIP_H */
