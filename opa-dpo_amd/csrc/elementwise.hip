// HBM-bound row / elementwise kernels (bf16 storage, fp32 math, 16-byte vector accesses).
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

namespace {
// 16-byte accesses of tensors that are streamed exactly once by these HBM-bound kernels; `nt` (launch_* : OPADPO_EW_NT) marks them
// non-temporal so that they do not push the next GEMM's operand panels out of the 4-MiB L2s
typedef __attribute__((ext_vector_type(4))) unsigned ew_u4_t;
typedef __attribute__((ext_vector_type(4))) float ew_f4_t;
__device__ __forceinline__ uint4 ew_ld16(const void* p, int nt) {
  const ew_u4_t v = nt ? __builtin_nontemporal_load((const ew_u4_t*)p) : *(const ew_u4_t*)p;
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void ew_st16(void* p, const uint4& v, int nt) {
  const ew_u4_t w = {v.x, v.y, v.z, v.w};
  if (nt) __builtin_nontemporal_store(w, (ew_u4_t*)p); else *(ew_u4_t*)p = w;
}
__device__ __forceinline__ void ew_st16f(float* p, float a, float b, float c, float d, int nt) {
  const ew_f4_t w = {a, b, c, d};
  if (nt) __builtin_nontemporal_store(w, (ew_f4_t*)p); else *(ew_f4_t*)p = w;
}


// ---- RMSNorm ------------------------------------------------------------------------------
// one block (256 threads) per row; H % 8 == 0.  y = x * rsqrt(mean(x^2) + eps) * w.  The residual
// stream x may be fp32 (XF) — the LLM keeps it in fp32 so that bf16 rounding does not accumulate
// across the 2*n_layers residual additions.
template <bool XF>
__device__ __forceinline__ void load8(const void* base, size_t idx, float* f) {
  if constexpr (XF) {
    const float4 a = *(const float4*)((const float*)base + idx);
    const float4 b = *(const float4*)((const float*)base + idx + 4);
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
  } else {
    unpack8(*(const uint4*)((const bf16_t*)base + idx), f);
  }
}

template <bool XF>
__global__ __launch_bounds__(256) void rmsnorm_fwd_kernel(const void* x, const bf16_t* w, bf16_t* y, float* rstd,
                                                           int H, float eps) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  float ss = 0.f;
  if (H <= 256 * 8 * 3) {
    // row held in registers between the two passes, norm weight fetched before the reduction: two dependent memory round
    // trips instead of four (decode: 8 rows per launch, the kernel is nothing but latency; training: x is read once)
    float f[3][8], g[3][8];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int i = threadIdx.x * 8 + it * 2048;
      if (i < H) {
        load8<XF>(x, row * H + i, f[it]);
        unpack8(*(const uint4*)(w + i), g[it]);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += f[it][j] * f[it][j];
      }
    }
    ss = block_sum_256(ss, red);
    const float r = rsqrtf(ss / H + eps);
    if (rstd && threadIdx.x == 0) rstd[row] = r;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
      const int i = threadIdx.x * 8 + it * 2048;
      if (i < H) {
#pragma unroll
        for (int j = 0; j < 8; ++j) f[it][j] = f[it][j] * r * g[it][j];
        *(uint4*)(y + row * H + i) = pack8(f[it]);
      }
    }
    return;
  }
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float f[8];
    load8<XF>(x, row * H + i, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += f[j] * f[j];
  }
  ss = block_sum_256(ss, red);
  const float r = rsqrtf(ss / H + eps);
  if (rstd && threadIdx.x == 0) rstd[row] = r;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float f[8], g[8];
    load8<XF>(x, row * H + i, f);
    unpack8(*(const uint4*)(w + i), g);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = f[j] * r * g[j];
    *(uint4*)(y + row * H + i) = pack8(f);
  }
}

// dx = rstd * (g - xhat * mean(g * xhat)) (+ dres),  g = dy * w, xhat = x * rstd.
// Writes the fp32 gradient residual stream (dx_f32) and/or its bf16 copy (dx_bf16, GEMM operand).
template <bool XF, bool RF>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const bf16_t* dy, const void* x, const bf16_t* w,
                                                           const float* rstd, const void* dres, float* dx_f32,
                                                           bf16_t* dx_bf16, int H, int nt) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  const float r = rstd[row];
  float dot = 0.f;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float a[8], b[8], c[8];
    unpack8(*(const uint4*)(dy + row * H + i), a);
    load8<XF>(x, row * H + i, b);
    unpack8(*(const uint4*)(w + i), c);
#pragma unroll
    for (int j = 0; j < 8; ++j) dot += a[j] * c[j] * b[j] * r;
  }
  dot = block_sum_256(dot, red) / H;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float a[8], b[8], c[8], o[8];
    unpack8(*(const uint4*)(dy + row * H + i), a);
    load8<XF>(x, row * H + i, b);
    unpack8(*(const uint4*)(w + i), c);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = r * (a[j] * c[j] - b[j] * r * dot);
    if (dres) {
      float d[8];
      load8<RF>(dres, row * H + i, d);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += d[j];
    }
    if (dx_f32) {
      ew_st16f(dx_f32 + row * H + i, o[0], o[1], o[2], o[3], nt);
      ew_st16f(dx_f32 + row * H + i + 4, o[4], o[5], o[6], o[7], nt);
    }
    if (dx_bf16) ew_st16(dx_bf16 + row * H + i, pack8(o), nt);
  }
}

// ---- LayerNorm (CLIP) ---------------------------------------------------------------------
// XF32 / YF32: the CLIP tower's residual stream is fp32 in the DPO path (round 4: 46 bf16 roundings of x per image removed - the
// ablation of round 3 put a quarter of the 8-layer log-prob error on the tower); the pre-LayerNorm reads and writes that stream, the
// two LayerNorms of a block read it and write the bf16 operand of the next GEMM.  (bf16 in / out: the OPA-SFT stage's trainable tower.)
template <bool XF32, bool YF32>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const void* x_, const bf16_t* w, const bf16_t* b,
                                                             void* y_, int H, float eps) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  auto load8 = [&](int i, float* f) {
    if constexpr (XF32) {
      const float4 a = *(const float4*)((const float*)x_ + row * H + i), c = *(const float4*)((const float*)x_ + row * H + i + 4);
      f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = c.x; f[5] = c.y; f[6] = c.z; f[7] = c.w;
    } else {
      unpack8(*(const uint4*)((const bf16_t*)x_ + row * H + i), f);
    }
  };
  float s = 0.f;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float f[8];
    load8(i, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j];
  }
  const float mean = block_sum_256(s, red) / H;
  float v = 0.f;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float f[8];
    load8(i, f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = f[j] - mean; v += d * d; }
  }
  const float r = rsqrtf(block_sum_256(v, red) / H + eps);
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float f[8], g[8], c[8];
    load8(i, f);
    unpack8(*(const uint4*)(w + i), g);
    unpack8(*(const uint4*)(b + i), c);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (f[j] - mean) * r * g[j] + c[j];
    if constexpr (YF32) {
      *(float4*)((float*)y_ + row * H + i) = make_float4(f[0], f[1], f[2], f[3]);
      *(float4*)((float*)y_ + row * H + i + 4) = make_float4(f[4], f[5], f[6], f[7]);
    } else {
      *(uint4*)((bf16_t*)y_ + row * H + i) = pack8(f);
    }
  }
}

// ---- RoPE (half-split rotate, HF Llama) ------------------------------------------------------
// buffer qk: row r (= s*L + pos), head-major columns; rotates n_heads heads starting at column 0
// (call once for the q section and once for the k section, or with 2*n_heads when contiguous).
// forward: x' = x*cos + rot(x)*sin ; inverse (gradient): g' = g*cos - rot(g)*sin  with rot(x) = [-x2, x1]
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* qk, int ld, const float* cosb, const float* sinb,
                                                    size_t total, int L, int n_heads, int hd, int inverse, const int32_t* pos_base,
                                                    int seg_prefix, int seg_len, const int32_t* row_pos, float l2theta) {
  const int half = hd / 2;
  const int pos0 = pos_base ? pos_base[0] : 0;     // device-resident position offset (graph-replayed decode step)
  const int per_row = n_heads * (half / 8);     // 8 (x1,x2) pairs per thread
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t row = idx / per_row;
    const int rem = (int)(idx % per_row);
    const int head = rem / (half / 8), i0 = (rem % (half / 8)) * 8;
    int pos;
    if (row_pos) {                       // ragged rows (padding removed): the position of every row is given
      pos = row_pos[row];
    } else {
      pos = pos0 + (int)(row % L);
      // packed responses sharing a prefix: every response restarts at position seg_prefix
      if (seg_len > 0 && pos >= seg_prefix + seg_len) pos -= ((pos - seg_prefix) / seg_len) * seg_len;
    }
    bf16_t* base = qk + row * ld + head * hd;
    float x1[8], x2[8];
    unpack8(*(const uint4*)(base + i0), x1);
    unpack8(*(const uint4*)(base + half + i0), x2);
    float o1[8], o2[8];
    if (l2theta > 0.f) {
      // TABLE-FREE angles (ragged context passes): the hardware sin / cos of the fractional revolution pos * theta^(-2i/hd) / 2pi - the angle
      // definition of the q|k|v projection's rotary epilogue (gemm_nt_w4_kernel) and of the attention backward's inverse rotation, in the
      // SAME arithmetic form as that epilogue (product, sign, fused multiply-add), so a pass that is too small for the fused epilogue
      // rotates q / k to the same bits and the backward of every ragged pass is the transpose of its forward whatever its size
      const float posf = (float)pos;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float frev = __builtin_amdgcn_exp2f(-(float)(2 * (i0 + j)) * (1.0f / (float)hd) * l2theta) * 0.15915494309189535f;
        const float x = __builtin_amdgcn_fractf(posf * frev);
        const float c = __builtin_amdgcn_cosf(x), s = inverse ? -__builtin_amdgcn_sinf(x) : __builtin_amdgcn_sinf(x);
        const float t1 = x2[j] * s, t2 = x1[j] * s;
        o1[j] = __builtin_fmaf(x1[j], c, -t1);
        o2[j] = __builtin_fmaf(x2[j], c, t2);
      }
    } else {
      const float* cp = cosb + (size_t)pos * half + i0;
      const float* sp = sinb + (size_t)pos * half + i0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float c = cp[j], s = inverse ? -sp[j] : sp[j];
        o1[j] = x1[j] * c - x2[j] * s;
        o2[j] = x2[j] * c + x1[j] * s;
      }
    }
    *(uint4*)(base + i0) = pack8(o1);
    *(uint4*)(base + half + i0) = pack8(o2);
  }
}

// LayerNorm backward w.r.t. x (CLIP blocks; the affine parameters are frozen under LoRA): mean / rstd are recomputed
// from x (one row = one block, three passes over <= 8 KiB that stay in L1/registers), dx = rstd * (g - mean(g) - xhat * mean(g xhat)),
// g = dy * w; `dres` (nullable) is the gradient arriving over the residual connection and is added.
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* dy, const bf16_t* x, const bf16_t* w, const bf16_t* dres,
                                                             bf16_t* dx, int H, float eps) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  const bf16_t* xr = x + row * H;
  const bf16_t* dyr = dy + row * H;
  float s = 0.f;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float f[8];
    unpack8(*(const uint4*)(xr + i), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) s += f[j];
  }
  const float mean = block_sum_256(s, red) / H;
  float v = 0.f;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float f[8];
    unpack8(*(const uint4*)(xr + i), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float d = f[j] - mean; v += d * d; }
  }
  const float r = rsqrtf(block_sum_256(v, red) / H + eps);
  float sg = 0.f, sgx = 0.f;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float f[8], g[8], ww[8];
    unpack8(*(const uint4*)(xr + i), f);
    unpack8(*(const uint4*)(dyr + i), g);
    unpack8(*(const uint4*)(w + i), ww);
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float gj = g[j] * ww[j]; sg += gj; sgx += gj * (f[j] - mean) * r; }
  }
  const float mg = block_sum_256(sg, red) / H;
  const float mgx = block_sum_256(sgx, red) / H;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float f[8], g[8], ww[8], o[8];
    unpack8(*(const uint4*)(xr + i), f);
    unpack8(*(const uint4*)(dyr + i), g);
    unpack8(*(const uint4*)(w + i), ww);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = r * (g[j] * ww[j] - mg - (f[j] - mean) * r * mgx);
    if (dres) {
      float d[8];
      unpack8(*(const uint4*)(dres + row * H + i), d);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += d[j];
    }
    *(uint4*)(dx + row * H + i) = pack8(o);
  }
}

// element-wise activation on a pre-activation tensor kept for backward (training forward of the CLIP MLP / projector)
__device__ __forceinline__ float ew_act(float z, int act) {
  if (act == OPADPO_ACT_QUICK_GELU) return z / (1.0f + __expf(-1.702f * z));
  return 0.5f * z * (1.0f + erff(z * 0.70710678118654752f));
}
__device__ __forceinline__ float ew_act_grad(float z, int act) {
  if (act == OPADPO_ACT_QUICK_GELU) {
    const float sg = 1.0f / (1.0f + __expf(-1.702f * z));
    return sg + 1.702f * z * sg * (1.0f - sg);
  }
  return 0.5f * (1.0f + erff(z * 0.70710678118654752f)) + z * 0.3989422804014327f * __expf(-0.5f * z * z);
}
__global__ __launch_bounds__(256) void act_fwd_kernel(const bf16_t* z, bf16_t* out, size_t n8, int act) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float f[8];
    unpack8(*(const uint4*)(z + i * 8), f);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = ew_act(f[j], act);
    *(uint4*)(out + i * 8) = pack8(f);
  }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(const bf16_t* dout, const bf16_t* z, bf16_t* dz, size_t n8, int act) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    float f[8], g[8];
    unpack8(*(const uint4*)(z + i * 8), f);
    unpack8(*(const uint4*)(dout + i * 8), g);
#pragma unroll
    for (int j = 0; j < 8; ++j) g[j] *= ew_act_grad(f[j], act);
    *(uint4*)(dz + i * 8) = pack8(g);
  }
}

// ---- SwiGLU ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void silu_mul_fwd_kernel(const bf16_t* gu, bf16_t* act, size_t rows, int F, int nt) {
  const int per_row = F / 8;
  const size_t total = rows * per_row;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t row = idx / per_row;
    const int i = (int)(idx % per_row) * 8;
    float g[8], u[8], o[8];
    unpack8(*(const uint4*)(gu + row * 2 * F + i), g);
    unpack8(*(const uint4*)(gu + row * 2 * F + F + i), u);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = g[j] / (1.0f + __expf(-g[j])) * u[j];
    ew_st16(act + row * F + i, pack8(o), nt);
  }
}

__global__ __launch_bounds__(256) void silu_mul_bwd_kernel(const bf16_t* dact, const bf16_t* gu, bf16_t* dgu,
                                                            size_t rows, int F, int nt) {
  const int per_row = F / 8;
  const size_t total = rows * per_row;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t row = idx / per_row;
    const int i = (int)(idx % per_row) * 8;
    float g[8], u[8], d[8], dg[8], du[8];
    unpack8(ew_ld16(gu + row * 2 * F + i, nt), g);
    unpack8(ew_ld16(gu + row * 2 * F + F + i, nt), u);
    unpack8(ew_ld16(dact + row * F + i, nt), d);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float sg = 1.0f / (1.0f + __expf(-g[j]));
      const float silu = g[j] * sg;
      du[j] = d[j] * silu;
      dg[j] = d[j] * u[j] * sg * (1.0f + g[j] * (1.0f - sg));
    }
    ew_st16(dgu + row * 2 * F + i, pack8(dg), nt);
    ew_st16(dgu + row * 2 * F + F + i, pack8(du), nt);
  }
}

// ---- embedding gather + multimodal splice --------------------------------------------------
// Row s: ids[s, 0..n_txt) with exactly one image_token at position p -> output row layout
// [text before | P image features | text after], L = n_txt + P - 1.  One block per output
// position.  Also emits the key mask [S,L].
__global__ __launch_bounds__(256) void embed_splice_kernel(const int32_t* ids, const uint8_t* text_mask, const bf16_t* embed,
                                                            const bf16_t* feats, const int32_t* feat_row,
                                                            const uint8_t* image_mask, void* x, int x_f32, uint8_t* key_mask,
                                                            int n_txt, int P, int H, int image_token) {
  __shared__ int img_pos;
  const int s = blockIdx.y, pos = blockIdx.x;
  const int L = n_txt + P - 1;
  if (threadIdx.x == 0) img_pos = n_txt;   // "no image token" -> plain text row (never for valid input)
  __syncthreads();
  for (int i = threadIdx.x; i < n_txt; i += 256)
    if (ids[(size_t)s * n_txt + i] == image_token) img_pos = i;
  __syncthreads();
  const int ip = img_pos;
  const bf16_t* src;
  uint8_t m;
  if (pos < ip) {
    const int t = pos;
    const int id = ids[(size_t)s * n_txt + t];
    src = embed + (size_t)max(id, 0) * H;
    m = text_mask[(size_t)s * n_txt + t];
  } else if (pos < ip + P && ip < n_txt) {
    const int j = pos - ip;
    src = feats + ((size_t)feat_row[s] * P + j) * H;
    m = image_mask ? image_mask[(size_t)s * P + j] : (uint8_t)1;
  } else {
    const int t = pos - (ip < n_txt ? P - 1 : 0);
    const int id = ids[(size_t)s * n_txt + min(t, n_txt - 1)];
    src = embed + (size_t)max(id, 0) * H;
    m = text_mask[(size_t)s * n_txt + min(t, n_txt - 1)];
  }
  const size_t orow = ((size_t)s * L + pos) * H;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    const uint4 v = *(const uint4*)(src + i);
    if (x_f32) {
      float f[8];
      unpack8(v, f);
      *(float4*)((float*)x + orow + i) = make_float4(f[0], f[1], f[2], f[3]);
      *(float4*)((float*)x + orow + i + 4) = make_float4(f[4], f[5], f[6], f[7]);
    } else {
      *(uint4*)((bf16_t*)x + orow + i) = v;
    }
  }
  if (threadIdx.x == 0) key_mask[(size_t)s * L + pos] = m;
}

// ---- CLIP patch embedding helpers -------------------------------------------------------------
// im2col for the stride==kernel conv: out[(b*P + py*G + px), k = c*patch*patch + dy*patch + dx], zero padded to kpad
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* pixels, bf16_t* out, int image_size, int patch, int kpad) {
  const int G = image_size / patch;
  const int P = G * G;
  const size_t prow = blockIdx.x;           // b*P + p
  const int b = (int)(prow / P), pp = (int)(prow % P);
  const int py = pp / G, px = pp % G;
  const int K = 3 * patch * patch;
  for (int k = threadIdx.x; k < kpad; k += 256) {
    bf16_t v = 0;
    if (k < K) {
      const int ch = k / (patch * patch), r = k % (patch * patch);
      const int dy = r / patch, dx = r % patch;
      v = pixels[(((size_t)b * 3 + ch) * image_size + (py * patch + dy)) * image_size + px * patch + dx];
    }
    out[prow * kpad + k] = v;
  }
}

// x[b, 0] = cls + pos[0] ; x[b, 1+p] = patches[b*P+p] + pos[1+p]
template <bool XF32>
__global__ __launch_bounds__(256) void vision_embed_kernel(const bf16_t* patches, const bf16_t* cls, const bf16_t* pos,
                                                            void* x, int P, int h) {
  const int b = blockIdx.y, t = blockIdx.x;   // t in [0, P]
  const bf16_t* src = (t == 0) ? cls : patches + ((size_t)b * P + t - 1) * h;
  for (int i = threadIdx.x * 8; i < h; i += 256 * 8) {
    float a[8], c[8];
    unpack8(*(const uint4*)(src + i), a);
    unpack8(*(const uint4*)(pos + (size_t)t * h + i), c);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += c[j];
    if constexpr (XF32) {
      float* xo = (float*)x + ((size_t)b * (P + 1) + t) * h + i;
      *(float4*)xo = make_float4(a[0], a[1], a[2], a[3]);
      *(float4*)(xo + 4) = make_float4(a[4], a[5], a[6], a[7]);
    } else {
      *(uint4*)((bf16_t*)x + ((size_t)b * (P + 1) + t) * h + i) = pack8(a);
    }
  }
}

// ---- row gather / scatter ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const bf16_t* src, int ld_src, const int32_t* rows_idx,
                                                           bf16_t* dst, int H) {
  const size_t r = blockIdx.x;
  const bf16_t* s = src + (size_t)rows_idx[r] * ld_src;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) *(uint4*)(dst + r * H + i) = *(const uint4*)(s + i);
}
// dst[rows_idx[r]] += src[r] (fp32, hardware float atomics: duplicate row indices accumulate)
__global__ __launch_bounds__(256) void scatter_add_rows_f32_kernel(const float* src, const int32_t* rows_idx, float* dst, int ld_dst, int H) {
  const size_t r = blockIdx.x;
  float* d = dst + (size_t)rows_idx[r] * ld_dst;
  for (int i = threadIdx.x; i < H; i += 256) atomicAdd(d + i, src[r * H + i]);
}
__global__ __launch_bounds__(256) void scatter_rows_kernel(const bf16_t* src, const int32_t* rows_idx, bf16_t* dst,
                                                            int ld_dst, int H) {
  const size_t r = blockIdx.x;
  bf16_t* d = dst + (size_t)rows_idx[r] * ld_dst;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) *(uint4*)(d + i) = *(const uint4*)(src + r * H + i);
}

// ---- bf16 transpose [R,C] -> [C,R] through a 64x64 LDS tile -----------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* in, bf16_t* out, int R, int C) {
  __shared__ bf16_t tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(size_t)(r0 + r) * C + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < R && c0 + c < C) out[(size_t)(c0 + c) * R + r0 + r] = tile[r][c];
  }
}

// many small transposes in one launch (the K-major copies of every LoRA block after an optimizer step were 350 launches of
// ~10 us): job j = {src offset, dst offset, rows, cols} in elements of the two flat buffers; blockIdx.y = job, blockIdx.x = tile
__global__ __launch_bounds__(256) void transpose_batched_kernel(const bf16_t* src, bf16_t* dst, const long long* jobs) {
  __shared__ bf16_t tile[64][66];
  const long long* jb = jobs + (size_t)blockIdx.y * 4;
  const int R = (int)jb[2], C = (int)jb[3];
  const int tiles_c = (C + 63) / 64, tiles = tiles_c * ((R + 63) / 64);
  if ((int)blockIdx.x >= tiles) return;
  const bf16_t* in = src + jb[0];
  bf16_t* out = dst + jb[1];
  const int r0 = ((int)blockIdx.x / tiles_c) * 64, c0 = ((int)blockIdx.x % tiles_c) * 64;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int r = i >> 6, c = i & 63;
    tile[r][c] = (r0 + r < R && c0 + c < C) ? in[(size_t)(r0 + r) * C + c0 + c] : (bf16_t)0;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = i >> 6, r = i & 63;
    if (r0 + r < R && c0 + c < C) out[(size_t)(c0 + c) * R + r0 + r] = tile[r][c];
  }
}

__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* in, bf16_t* out, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 256 * 4) {
    if (i + 3 < n) {
      const float4 v = *(const float4*)(in + i);
      uint2 o;
      o.x = pack_bf2(v.x, v.y);
      o.y = pack_bf2(v.z, v.w);
      *(uint2*)(out + i) = o;
    } else {
      for (size_t j = i; j < n; ++j) out[j] = f2bf(in[j]);
    }
  }
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const bf16_t* in, float* out, size_t n) {
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (size_t)gridDim.x * 256 * 8) {
    if (i + 7 < n) {
      const uint4 v = *(const uint4*)(in + i);
      float4 a, b;
      a.x = __uint_as_float(v.x << 16); a.y = __uint_as_float(v.x & 0xffff0000u);
      a.z = __uint_as_float(v.y << 16); a.w = __uint_as_float(v.y & 0xffff0000u);
      b.x = __uint_as_float(v.z << 16); b.y = __uint_as_float(v.z & 0xffff0000u);
      b.z = __uint_as_float(v.w << 16); b.w = __uint_as_float(v.w & 0xffff0000u);
      *(float4*)(out + i) = a;
      *(float4*)(out + i + 4) = b;
    } else {
      for (size_t j = i; j < n; ++j) out[j] = __uint_as_float((uint32_t)in[j] << 16);
    }
  }
}

// fp32 [rows, C] contiguous -> bf16 columns [0, C) of a wider buffer with leading dimension ld
__global__ __launch_bounds__(256) void f32_to_bf16_strided_kernel(const float* in, bf16_t* out, size_t rows, int C, int ld) {
  const int per_row = C / 4;
  const size_t total = rows * per_row;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const size_t row = idx / per_row;
    const int i = (int)(idx % per_row) * 4;
    const float4 v = *(const float4*)(in + row * C + i);
    uint2 o;
    o.x = pack_bf2(v.x, v.y);
    o.y = pack_bf2(v.z, v.w);
    *(uint2*)(out + row * ld + i) = o;
  }
}

// x_out = resid + sum_s partials[s] (fp32), y = RMSNorm(x_out) * w (bf16): the residual add + RMSNorm that follows a K-split decode
// projection (gemm_nt_dec64_kernel MODE 1 writes one fp32 partial tile per K-slice; adding them here keeps the sum order fixed -
// deterministic, no atomics - and costs no launch).  One block per row, row held in registers between the two passes.
template <bool RF>
__global__ __launch_bounds__(256) void rmsnorm_sum_fwd_kernel(const void* resid, const float* partials, int n_partials, size_t partial_stride,
                                                               const bf16_t* w, float* x_out, bf16_t* y, float* rstd, int H, float eps, int nt) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  float f[3][8], g[3][8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int i = threadIdx.x * 8 + it * 2048;
    if (i < H) {
      load8<RF>(resid, row * H + i, f[it]);
      unpack8(*(const uint4*)(w + i), g[it]);
      for (int sidx = 0; sidx < n_partials; ++sidx) {
        const float* pp = partials + (size_t)sidx * partial_stride + row * H + i;
        const float4 a = *(const float4*)pp, b = *(const float4*)(pp + 4);
        f[it][0] += a.x; f[it][1] += a.y; f[it][2] += a.z; f[it][3] += a.w;
        f[it][4] += b.x; f[it][5] += b.y; f[it][6] += b.z; f[it][7] += b.w;
      }
      ew_st16f(x_out + row * H + i, f[it][0], f[it][1], f[it][2], f[it][3], nt);
      ew_st16f(x_out + row * H + i + 4, f[it][4], f[it][5], f[it][6], f[it][7], nt);
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += f[it][j] * f[it][j];
    }
  }
  ss = block_sum_256(ss, red);
  const float r = rsqrtf(ss / H + eps);
  if (rstd && threadIdx.x == 0) rstd[row] = r;
  if (!y) return;
#pragma unroll
  for (int it = 0; it < 3; ++it) {
    const int i = threadIdx.x * 8 + it * 2048;
    if (i < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) f[it][j] = f[it][j] * r * g[it][j];
      ew_st16(y + row * H + i, pack8(f[it]), nt);
    }
  }
}

inline int ew_grid(size_t total_threads) {
  size_t b = (total_threads + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

hipError_t launch_rmsnorm_fwd(const void* x, int x_f32, const bf16_t* w, bf16_t* y, float* rstd, int rows, int H, float eps, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  if (H % 8) return hipErrorInvalidValue;
  if (x_f32) hipLaunchKernelGGL(rmsnorm_fwd_kernel<true>, dim3(rows), dim3(256), 0, st, x, w, y, rstd, H, eps);
  else hipLaunchKernelGGL(rmsnorm_fwd_kernel<false>, dim3(rows), dim3(256), 0, st, x, w, y, rstd, H, eps);
  return hipGetLastError();
}
// OPADPO_EW_NT=1: non-temporal stores (and streamed-once loads) in the four HBM-bound passes between the GEMMs (A/B switch, see ew_st16)
static int ew_nt() { static const int v = getenv("OPADPO_EW_NT") ? atoi(getenv("OPADPO_EW_NT")) : 0; return v; }
hipError_t launch_rmsnorm_sum_fwd(const void* resid, int resid_f32, const float* partials, int n_partials, size_t partial_stride, const bf16_t* w,
                                  float* x_out, bf16_t* y, float* rstd, int rows, int H, float eps, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  if (H % 8 || H > 256 * 8 * 3 || n_partials < 0 || (n_partials > 0 && !partials) || !x_out) return hipErrorInvalidValue;
  const int nt = rows >= 4096 ? ew_nt() : 0;       // decode-sized calls keep their rows in cache
  if (resid_f32) hipLaunchKernelGGL(rmsnorm_sum_fwd_kernel<true>, dim3(rows), dim3(256), 0, st, resid, partials, n_partials, partial_stride, w, x_out, y, rstd, H, eps, nt);
  else hipLaunchKernelGGL(rmsnorm_sum_fwd_kernel<false>, dim3(rows), dim3(256), 0, st, resid, partials, n_partials, partial_stride, w, x_out, y, rstd, H, eps, nt);
  return hipGetLastError();
}
hipError_t launch_rmsnorm_bwd(const bf16_t* dy, const void* x, int x_f32, const bf16_t* w, const float* rstd, const void* dres,
                              int dres_f32, float* dx_f32, bf16_t* dx_bf16, int rows, int H, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  if (H % 8) return hipErrorInvalidValue;
  const int nt = rows >= 4096 ? ew_nt() : 0;
#define RB(XF, RF) hipLaunchKernelGGL((rmsnorm_bwd_kernel<XF, RF>), dim3(rows), dim3(256), 0, st, dy, x, w, rstd, dres, dx_f32, dx_bf16, H, nt)
  if (x_f32 && dres_f32) RB(true, true);
  else if (x_f32) RB(true, false);
  else if (dres_f32) RB(false, true);
  else RB(false, false);
#undef RB
  return hipGetLastError();
}
hipError_t launch_layernorm_fwd(const void* x, const bf16_t* w, const bf16_t* b, void* y, int rows, int H, float eps, hipStream_t st, int x_f32, int y_f32) {
  if (rows <= 0) return hipSuccess;
  if (H % 8 || (y_f32 && !x_f32)) return hipErrorInvalidValue;
  if (x_f32 && y_f32) hipLaunchKernelGGL((layernorm_fwd_kernel<true, true>), dim3(rows), dim3(256), 0, st, x, w, b, y, H, eps);
  else if (x_f32) hipLaunchKernelGGL((layernorm_fwd_kernel<true, false>), dim3(rows), dim3(256), 0, st, x, w, b, y, H, eps);
  else hipLaunchKernelGGL((layernorm_fwd_kernel<false, false>), dim3(rows), dim3(256), 0, st, x, w, b, y, H, eps);
  return hipGetLastError();
}
hipError_t launch_layernorm_bwd(const bf16_t* dy, const bf16_t* x, const bf16_t* w, const bf16_t* dres, bf16_t* dx, int rows, int H,
                                float eps, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  if (H % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(rows), dim3(256), 0, st, dy, x, w, dres, dx, H, eps);
  return hipGetLastError();
}
hipError_t launch_act_fwd(const bf16_t* z, bf16_t* out, size_t n, int act, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (n % 8 || (act != OPADPO_ACT_QUICK_GELU && act != OPADPO_ACT_GELU)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(act_fwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, st, z, out, n / 8, act);
  return hipGetLastError();
}
hipError_t launch_act_bwd(const bf16_t* dout, const bf16_t* z, bf16_t* dz, size_t n, int act, hipStream_t st) {
  if (n == 0) return hipSuccess;
  if (n % 8 || (act != OPADPO_ACT_QUICK_GELU && act != OPADPO_ACT_GELU)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(act_bwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, st, dout, z, dz, n / 8, act);
  return hipGetLastError();
}
hipError_t launch_rope(bf16_t* qk, int ld, const float* cosb, const float* sinb, int rows, int L, int n_heads, int hd,
                       int inverse, const int32_t* pos_base, int seg_prefix, int seg_len, hipStream_t st, const int32_t* row_pos, float l2theta) {
  if (rows <= 0) return hipSuccess;
  if (hd % 16) return hipErrorInvalidValue;
  if (l2theta > 0.f && !row_pos) return hipErrorInvalidValue;      // table-free angles need the per-row positions
  const size_t total = (size_t)rows * n_heads * (hd / 16);
  hipLaunchKernelGGL(rope_kernel, dim3(ew_grid(total)), dim3(256), 0, st, qk, ld, cosb, sinb, total, L, n_heads, hd, inverse, pos_base, seg_prefix, seg_len, row_pos, l2theta);
  return hipGetLastError();
}
hipError_t launch_silu_mul_fwd(const bf16_t* gu, bf16_t* act, int rows, int F, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  if (F % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(silu_mul_fwd_kernel, dim3(ew_grid((size_t)rows * (F / 8))), dim3(256), 0, st, gu, act, (size_t)rows, F, rows >= 4096 ? ew_nt() : 0);
  return hipGetLastError();
}
hipError_t launch_silu_mul_bwd(const bf16_t* dact, const bf16_t* gu, bf16_t* dgu, int rows, int F, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  if (F % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(silu_mul_bwd_kernel, dim3(ew_grid((size_t)rows * (F / 8))), dim3(256), 0, st, dact, gu, dgu, (size_t)rows, F, rows >= 4096 ? ew_nt() : 0);
  return hipGetLastError();
}
hipError_t launch_embed_splice(const int32_t* ids, const uint8_t* text_mask, const bf16_t* embed, const bf16_t* feats,
                               const int32_t* feat_row, const uint8_t* image_mask, void* x, int x_f32, uint8_t* key_mask,
                               int S, int n_txt, int P, int H, int image_token, hipStream_t st) {
  if (S <= 0) return hipSuccess;
  if (H % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(embed_splice_kernel, dim3(n_txt + P - 1, S), dim3(256), 0, st, ids, text_mask, embed, feats, feat_row,
                     image_mask, x, x_f32, key_mask, n_txt, P, H, image_token);
  return hipGetLastError();
}
hipError_t launch_im2col(const bf16_t* pixels, bf16_t* out, int B, int image_size, int patch, int kpad, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  const int G = image_size / patch;
  hipLaunchKernelGGL(im2col_kernel, dim3(B * G * G), dim3(256), 0, st, pixels, out, image_size, patch, kpad);
  return hipGetLastError();
}
hipError_t launch_vision_embed(const bf16_t* patches, const bf16_t* cls, const bf16_t* pos, void* x, int B, int P, int h, hipStream_t st, int x_f32) {
  if (B <= 0) return hipSuccess;
  if (h % 8) return hipErrorInvalidValue;
  if (x_f32) hipLaunchKernelGGL(vision_embed_kernel<true>, dim3(P + 1, B), dim3(256), 0, st, patches, cls, pos, x, P, h);
  else hipLaunchKernelGGL(vision_embed_kernel<false>, dim3(P + 1, B), dim3(256), 0, st, patches, cls, pos, x, P, h);
  return hipGetLastError();
}
hipError_t launch_gather_rows(const bf16_t* src, int ld_src, const int32_t* rows_idx, bf16_t* dst, int n, int H, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (H % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(n), dim3(256), 0, st, src, ld_src, rows_idx, dst, H);
  return hipGetLastError();
}
hipError_t launch_scatter_rows(const bf16_t* src, const int32_t* rows_idx, bf16_t* dst, int ld_dst, int n, int H, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (H % 8) return hipErrorInvalidValue;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3(n), dim3(256), 0, st, src, rows_idx, dst, ld_dst, H);
  return hipGetLastError();
}
hipError_t launch_scatter_add_rows_f32(const float* src, const int32_t* rows_idx, float* dst, int ld_dst, int n, int H, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(scatter_add_rows_f32_kernel, dim3(n), dim3(256), 0, st, src, rows_idx, dst, ld_dst, H);
  return hipGetLastError();
}
hipError_t launch_transpose(const bf16_t* in, bf16_t* out, int R, int C, hipStream_t st) {
  if (R <= 0 || C <= 0) return hipSuccess;
  hipLaunchKernelGGL(transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, st, in, out, R, C);
  return hipGetLastError();
}
hipError_t launch_transpose_batched(const bf16_t* src, bf16_t* dst, const long long* jobs, int n_jobs, int max_tiles, hipStream_t st) {
  if (n_jobs <= 0 || max_tiles <= 0) return hipSuccess;
  if (n_jobs > 65535) return hipErrorInvalidValue;
  hipLaunchKernelGGL(transpose_batched_kernel, dim3(max_tiles, n_jobs), dim3(256), 0, st, src, dst, jobs);
  return hipGetLastError();
}
hipError_t launch_f32_to_bf16(const float* in, bf16_t* out, size_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(ew_grid((n + 3) / 4)), dim3(256), 0, st, in, out, n);
  return hipGetLastError();
}
hipError_t launch_bf16_to_f32(const bf16_t* in, float* out, size_t n, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(ew_grid((n + 7) / 8)), dim3(256), 0, st, in, out, n);
  return hipGetLastError();
}
hipError_t launch_f32_to_bf16_strided(const float* in, bf16_t* out, size_t rows, int C, int ld, hipStream_t st) {
  if (rows == 0) return hipSuccess;
  if (C % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(f32_to_bf16_strided_kernel, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, st, in, out, rows, C, ld);
  return hipGetLastError();
}
