// Head ops (per-token log-prob / entropy over the vocabulary, and its gradient) and the fused
// clip + AdamW update over the flat LoRA parameter buffer.
#include "common.h"
#include "kernels.h"

namespace {

// One block per response row.  z = logits * inv_temp (fp32 logits straight from the lm_head GEMM
// accumulators).  logp = z[label] - lse ; H = lse - sum softmax(z) * z.  Rows whose label is the
// pad id (0) return logp = -0.0, H = 0 (utils/common_utils.py:112-118 + rl_models.py:127,132).
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* logits, int ldl, const int32_t* labels, float inv_temp,
                                                        float* logp, float* ent, float* lse_out, int V) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  const float* z = logits + row * ldl;
  float mx = -3.0e38f;
  for (int i = threadIdx.x * 4; i < V; i += 256 * 4) {
    const float4 v = *(const float4*)(z + i);
    mx = fmaxf(mx, fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)) * inv_temp);
  }
  mx = block_max_256(mx, red);
  float se = 0.f, sz = 0.f;
  for (int i = threadIdx.x * 4; i < V; i += 256 * 4) {
    const float4 v = *(const float4*)(z + i);
    const float a = v.x * inv_temp, b = v.y * inv_temp, c = v.z * inv_temp, d = v.w * inv_temp;
    const float ea = __expf(a - mx), eb = __expf(b - mx), ec = __expf(c - mx), ed = __expf(d - mx);
    se += ea + eb + ec + ed;
    sz += ea * a + eb * b + ec * c + ed * d;
  }
  se = block_sum_256(se, red);
  sz = block_sum_256(sz, red);
  if (threadIdx.x == 0) {
    const float lse = mx + __logf(se);
    const int lab = labels[row];
    lse_out[row] = lse;
    if (lab == 0) {
      logp[row] = -0.0f;
      ent[row] = 0.0f;
    } else {
      logp[row] = z[lab] * inv_temp - lse;
      ent[row] = lse - sz / se;
    }
  }
}

// dz[v] = dlogp * (1[v == label] - softmax(z)[v]) * inv_temp   (0 for pad rows), written as bf16
// dz = inv_temp * [ dlogp * (onehot(label) - p)  +  dent * (-p * (log p + H)) ]   (H = -sum p log p, its gradient is only
// needed by the OPA-SFT entropy regulariser: opa_trainer.py:64-90; ent / dent nullable)
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* logits, int ldl, const int32_t* labels, const float* lse,
                                                        const float* dlogp, const float* ent, const float* dent, float inv_temp,
                                                        bf16_t* dz, int ldz, int V, int col0) {
  // col0 > 0: `logits` / `dz` hold the vocabulary CHUNK [col0, col0 + V) of every row (chunked head: the logits are recomputed chunk
  // by chunk in the backward, nothing of size [rows, vocab] exists); lse / ent are the statistics of the WHOLE row
  const size_t row = blockIdx.x;
  const float* z = logits + row * ldl;
  const int lab_g = labels[row];
  const int lab = lab_g == 0 ? 0 : lab_g - col0;          // pad rows stay pad; other labels become chunk-local (may fall outside [0, V))
  const float g = (lab_g == 0) ? 0.f : dlogp[row];
  const float ge = (lab_g == 0 || !dent) ? 0.f : dent[row];
  const float Hrow = ent ? ent[row] : 0.f;
  const float l = lse[row];
  bf16_t* out = dz + row * ldz;
  for (int i = threadIdx.x * 4; i < V; i += 256 * 4) {
    const float4 v = *(const float4*)(z + i);
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float lp = o[j] * inv_temp - l;
      const float pr = __expf(lp);
      o[j] = (g * (((i + j) == lab && lab_g != 0 ? 1.0f : 0.0f) - pr) - ge * pr * (lp + Hrow)) * inv_temp;
    }
    uint2 w;
    w.x = pack_bf2(o[0], o[1]);
    w.y = pack_bf2(o[2], o[3]);
    *(uint2*)(out + i) = w;
  }
}

__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, size_t n, float* out) {
  __shared__ float red[4];
  float s = 0.f;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (size_t)gridDim.x * 256 * 4) {
    if (i + 3 < n) {
      const float4 v = *(const float4*)(g + i);
      s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    } else {
      for (size_t j = i; j < n; ++j) s += g[j] * g[j];
    }
  }
  s = block_sum_256(s, red);
  if (threadIdx.x == 0) atomicAdd(out, s);
}

// torch.optim.AdamW semantics (decoupled decay, bias correction, fp32 state) on g * grad_div *
// clip, clip = min(1, max_norm / (||g * grad_div|| + 1e-6)) with ||g||^2 read from `sumsq`
// (device scalar, already all-reduced when data parallel).  Also refreshes the bf16 working copy.
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, size_t n,
                                                     float lr, float beta1, float beta2, float eps, float weight_decay,
                                                     float bc1, float bc2_sqrt, const float* sumsq, float max_norm,
                                                     float grad_div) {
  float scale = grad_div;
  if (sumsq && max_norm > 0.f) {
    const float norm = sqrtf(sumsq[0]) * grad_div;
    scale *= fminf(1.0f, max_norm / (norm + 1e-6f));
  }
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * scale;
    float pi = p[i] * (1.0f - lr * weight_decay);
    const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
    const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    pi -= (lr / bc1) * (mi / denom);
    p[i] = pi; m[i] = mi; v[i] = vi;
    if (p_bf16) p_bf16[i] = f2bf(pi);
  }
}

inline int grid_for(size_t n) {
  size_t b = (n + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

namespace {
// ---- chunked head (lm_head + online log-sum-exp + label gather + entropy without a [rows, vocab] buffer) ----
// The lm_head GEMM runs over one vocabulary chunk [col0, col0 + n) at a time into a [rows, n] fp32 buffer that is REUSED by every
// chunk; this kernel folds the chunk into the row's running statistics (flash-attention's online softmax over the vocabulary):
//   m = running max of z, s = sum exp(z - m), t = sum z * exp(z - m), zl = z[label] once the label's chunk has passed
// One block per row, the chunk row lives in registers (n <= 256 * 4 * HC_PER): one read of the logits.
constexpr int HC_PER = 4;                 // float4 per thread -> chunks of up to 4096 columns
__global__ __launch_bounds__(256) void head_fwd_chunk_kernel(const float* logits, int ldl, const int32_t* labels, float inv_temp, int col0, int n,
                                                              int first, float* m_io, float* s_io, float* t_io, float* zl_io) {
  __shared__ float red[4];
  const size_t row = blockIdx.x;
  const float* z = logits + row * ldl;
  float4 v[HC_PER];
  float mx = -3.0e38f;
#pragma unroll
  for (int q = 0; q < HC_PER; ++q) {
    const int i = (q * 256 + threadIdx.x) * 4;
    if (i < n) {
      v[q] = *(const float4*)(z + i);
      v[q].x *= inv_temp; v[q].y *= inv_temp; v[q].z *= inv_temp; v[q].w *= inv_temp;
      mx = fmaxf(mx, fmaxf(fmaxf(v[q].x, v[q].y), fmaxf(v[q].z, v[q].w)));
    }
  }
  mx = block_max_256(mx, red);
  const float m_old = first ? -3.0e38f : m_io[row];
  const float m_new = fmaxf(m_old, mx);
  float se = 0.f, sz = 0.f;
#pragma unroll
  for (int q = 0; q < HC_PER; ++q) {
    const int i = (q * 256 + threadIdx.x) * 4;
    if (i < n) {
      const float ea = __expf(v[q].x - m_new), eb = __expf(v[q].y - m_new), ec = __expf(v[q].z - m_new), ed = __expf(v[q].w - m_new);
      se += ea + eb + ec + ed;
      sz += ea * v[q].x + eb * v[q].y + ec * v[q].z + ed * v[q].w;
    }
  }
  se = block_sum_256(se, red);
  sz = block_sum_256(sz, red);
  if (threadIdx.x == 0) {
    const float f = first ? 0.f : __expf(m_old - m_new);
    m_io[row] = m_new;
    s_io[row] = (first ? 0.f : s_io[row]) * f + se;
    t_io[row] = (first ? 0.f : t_io[row]) * f + sz;
    const int lab = labels[row] - col0;
    if (first) zl_io[row] = 0.f;
    if (labels[row] != 0 && lab >= 0 && lab < n) zl_io[row] = z[lab] * inv_temp;
  }
}
// logp = z[label] - lse, H = lse - t / s (pad rows: -0.0 / 0, utils/common_utils.py:112-118 + rl_models.py:127,132)
__global__ void head_fwd_finish_kernel(const int32_t* labels, const float* m, const float* s, const float* t, const float* zl, float* logp,
                                       float* ent, float* lse_out, int rows) {
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= rows) return;
  const float lse = m[row] + __logf(s[row]);
  lse_out[row] = lse;
  if (labels[row] == 0) { logp[row] = -0.0f; ent[row] = 0.0f; }
  else { logp[row] = zl[row] - lse; ent[row] = lse - t[row] / s[row]; }
}
}  // namespace

hipError_t launch_head_fwd_chunk(const float* logits, int ldl, const int32_t* labels, float inv_temp, int col0, int n, int first, float* m,
                                 float* s, float* t, float* zl, int rows, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  if (n <= 0 || n % 4 || ldl % 4 || n > 256 * 4 * HC_PER) return hipErrorInvalidValue;
  hipLaunchKernelGGL(head_fwd_chunk_kernel, dim3(rows), dim3(256), 0, st, logits, ldl, labels, inv_temp, col0, n, first, m, s, t, zl);
  return hipGetLastError();
}
hipError_t launch_head_fwd_finish(const int32_t* labels, const float* m, const float* s, const float* t, const float* zl, float* logp, float* ent,
                                  float* lse, int rows, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(head_fwd_finish_kernel, dim3((rows + 255) / 256), dim3(256), 0, st, labels, m, s, t, zl, logp, ent, lse, rows);
  return hipGetLastError();
}

hipError_t launch_head_fwd(const float* logits, int ldl, const int32_t* labels, float inv_temp, float* logp, float* ent,
                           float* lse, int rows, int V, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  if (V % 4 || ldl % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(head_fwd_kernel, dim3(rows), dim3(256), 0, st, logits, ldl, labels, inv_temp, logp, ent, lse, V);
  return hipGetLastError();
}
hipError_t launch_head_bwd(const float* logits, int ldl, const int32_t* labels, const float* lse, const float* dlogp,
                           const float* ent, const float* dent,
                           float inv_temp, bf16_t* dz, int ldz, int rows, int V, hipStream_t st, int col0) {
  if (rows <= 0) return hipSuccess;
  if (V % 4 || ldl % 4 || ldz % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(rows), dim3(256), 0, st, logits, ldl, labels, lse, dlogp, ent, dent, inv_temp, dz, ldz, V, col0);
  return hipGetLastError();
}
hipError_t launch_sumsq(const float* g, size_t n, float* out, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(sumsq_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, st, g, n, out);
  return hipGetLastError();
}
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, bf16_t* p_bf16, size_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, float bc1, float bc2_sqrt, const float* sumsq,
                        float max_norm, float grad_div, hipStream_t st) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n)), dim3(256), 0, st, p, g, m, v, p_bf16, n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2_sqrt, sumsq, max_norm, grad_div);
  return hipGetLastError();
}
