// On-policy rollout kernels: single-token attention over a KV cache and the on-device
// temperature -> top-k -> top-p -> multinomial sampler (HF logits-processor order; the reference
// calls policy.generate(do_sample=True, top_k=30, top_p=0.95): online_generator.py:292-309).
#include "common.h"
#include "kernels.h"

namespace {

// block = (head, batch row); 256 threads.  Phase 1: thread t scores keys t, t+256, ...
// Phase 2: softmax over the block.  Phase 3: 4 key groups x 64 lanes (hd/64 dims per lane).
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* q, int ldq, const bf16_t* kc, const bf16_t* vc,
                                                           bf16_t* o, const uint8_t* key_mask, int nh, int ctx_arg,
                                                           const int32_t* ctx_ptr, int max_ctx, float scale) {
  const int ctx = ctx_ptr ? min(ctx_ptr[0] + 1, max_ctx) : ctx_arg;   // device-resident: keys 0..pos (graph replay)
  extern __shared__ float sm[];          // [ctx] probabilities + [4*HD] partial outputs + [4] reduce
  float* prob = sm;
  float* part = sm + max_ctx;
  float* red = part + 4 * HD;
  const int h = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
  const size_t row_stride = (size_t)nh * HD;
  const bf16_t* qp = q + (size_t)b * ldq + h * HD;
  float qf[HD];
#pragma unroll
  for (int i = 0; i < HD / 8; ++i) unpack8(*(const uint4*)(qp + i * 8), qf + i * 8);
  float mx = -1.0e30f;
  for (int j = tid; j < ctx; j += 256) {
    float s = -1.0e30f;
    if (!key_mask || key_mask[(size_t)b * max_ctx + j]) {
      const bf16_t* kp = kc + ((size_t)b * max_ctx + j) * row_stride + h * HD;
      float acc = 0.f;
#pragma unroll
      for (int i = 0; i < HD / 8; ++i) {
        float kf[8];
        unpack8(*(const uint4*)(kp + i * 8), kf);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += kf[e] * qf[i * 8 + e];
      }
      s = acc * scale;
    }
    prob[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = block_max_256(mx, red);
  float se = 0.f;
  for (int j = tid; j < ctx; j += 256) {
    const float s = prob[j];
    const float e = (s > -1.0e29f) ? __expf(s - mx) : 0.f;
    prob[j] = e;
    se += e;
  }
  se = block_sum_256(se, red);
  __syncthreads();
  const int grp = tid >> 6, lane = tid & 63;
  constexpr int DPL = HD / 64;           // dims per lane
  float acc[DPL];
#pragma unroll
  for (int e = 0; e < DPL; ++e) acc[e] = 0.f;
  for (int j = grp; j < ctx; j += 4) {
    const float pj = prob[j];
    const bf16_t* vp = vc + ((size_t)b * max_ctx + j) * row_stride + h * HD + lane * DPL;
#pragma unroll
    for (int e = 0; e < DPL; ++e) acc[e] += pj * bf2f(vp[e]);
  }
#pragma unroll
  for (int e = 0; e < DPL; ++e) part[grp * HD + lane * DPL + e] = acc[e];
  __syncthreads();
  if (tid < HD) {
    const float v = (part[tid] + part[HD + tid] + part[2 * HD + tid] + part[3 * HD + tid]) / (se > 0.f ? se : 1.f);
    o[(size_t)b * row_stride + h * HD + tid] = f2bf(v);
  }
}

__device__ __forceinline__ uint32_t fkey(float f) {   // monotone float -> uint map
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ uint64_t mix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// step_ptr (nullable): device-resident step counter (graph replay) used for the RNG stream and, with `history`
// ([max_steps, rows]), for the slot the token is appended to.  eos_id >= 0: a row that samples EOS is marked finished.
__global__ __launch_bounds__(256) void sample_kernel(const float* logits, int ldl, int V, float inv_temp, int top_k, float top_p,
                                                      uint64_t seed, uint64_t step_arg, const int32_t* step_ptr, uint8_t* finished,
                                                      int pad_id, int eos_id, int32_t* out, int32_t* history) {
  const uint64_t step = step_ptr ? (uint64_t)step_ptr[0] : step_arg;
  __shared__ float red[4];
  __shared__ float scan[256];
  __shared__ uint32_t s_thr;
  const int row = blockIdx.x, tid = threadIdx.x;
  if (finished && finished[row]) {
    if (tid == 0) {
      out[row] = pad_id;
      if (history) history[step * gridDim.x + row] = pad_id;
    }
    return;
  }
  const float* z = logits + (size_t)row * ldl;
  float mx = -3.0e38f;
  for (int i = tid; i < V; i += 256) mx = fmaxf(mx, z[i] * inv_temp);
  mx = block_max_256(mx, red);

  uint32_t thr = 0;   // keep tokens with key >= thr
  if (top_k > 0 && top_k < V) {
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t cand = thr | (1u << bit);
      float cnt = 0.f;
      for (int i = tid; i < V; i += 256) cnt += (fkey(z[i] * inv_temp) >= cand) ? 1.f : 0.f;
      cnt = block_sum_256(cnt, red);
      if (cnt >= (float)top_k) thr = cand;
    }
  }
  if (top_p < 1.0f) {
    float zk = 0.f;
    for (int i = tid; i < V; i += 256) {
      const float v = z[i] * inv_temp;
      if (fkey(v) >= thr) zk += __expf(v - mx);
    }
    zk = block_sum_256(zk, red);
    const float budget = (1.0f - top_p) * zk;
    uint32_t t2 = 0;
    for (int bit = 31; bit >= 0; --bit) {
      const uint32_t cand = t2 | (1u << bit);
      float mass = 0.f;
      for (int i = tid; i < V; i += 256) {
        const float v = z[i] * inv_temp;
        const uint32_t k = fkey(v);
        if (k >= thr && k < cand) mass += __expf(v - mx);
      }
      mass = block_sum_256(mass, red);
      if (mass <= budget) t2 = cand;
    }
    thr = max(thr, t2);
  }
  // multinomial over the kept set, cumulative in index order; thread t owns a contiguous chunk
  const int chunk = (V + 255) / 256;
  const int lo = tid * chunk, hi = min(V, lo + chunk);
  float local = 0.f;
  for (int i = lo; i < hi; ++i) {
    const float v = z[i] * inv_temp;
    if (fkey(v) >= thr) local += __expf(v - mx);
  }
  scan[tid] = local;
  __syncthreads();
  const uint64_t r = mix64(mix64(seed ^ (step * 0xD1B54A32D192ED03ull)) + (uint64_t)row);
  const float u = ((float)(r >> 40) + 0.5f) * (1.0f / 16777216.0f);
  if (tid == 0) {
    float total = 0.f;
    for (int i = 0; i < 256; ++i) total += scan[i];
    const float target = u * total;
    float run = 0.f;
    int owner = 0;
    float owner_start = 0.f;
    for (int i = 0; i < 256; ++i) {          // last non-empty chunk whose start is <= target
      if (scan[i] > 0.f && run <= target) { owner = i; owner_start = run; }
      run += scan[i];
    }
    s_thr = (uint32_t)owner;
    red[0] = target;
    red[1] = owner_start;
  }
  __syncthreads();
  if (tid == (int)s_thr) {
    const float target = red[0];
    float run = red[1];
    int pick = pad_id;
    for (int i = lo; i < hi; ++i) {
      const float v = z[i] * inv_temp;
      if (fkey(v) >= thr) {
        run += __expf(v - mx);
        pick = i;
        if (run > target) break;
      }
    }
    out[row] = pick;
    if (history) history[step * gridDim.x + row] = pick;
    if (finished && eos_id >= 0 && pick == eos_id) finished[row] = 1;
  }
}

}  // namespace

hipError_t launch_attn_decode(const bf16_t* q, const bf16_t* kc, const bf16_t* vc, bf16_t* o, const uint8_t* key_mask,
                              int B, int nh, int hd, int ctx, const int32_t* ctx_ptr, int max_ctx, int ldq, float scale,
                              hipStream_t st) {
  if (B <= 0) return hipSuccess;
  const size_t smem = (size_t)(max_ctx + 4 * hd + 4) * sizeof(float);
  if (smem > 64 * 1024) return hipErrorInvalidValue;
  if (hd == 128)
    hipLaunchKernelGGL((attn_decode_kernel<128>), dim3(nh, B), dim3(256), smem, st, q, ldq, kc, vc, o, key_mask, nh, ctx, ctx_ptr, max_ctx, scale);
  else if (hd == 64)
    hipLaunchKernelGGL((attn_decode_kernel<64>), dim3(nh, B), dim3(256), smem, st, q, ldq, kc, vc, o, key_mask, nh, ctx, ctx_ptr, max_ctx, scale);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

hipError_t launch_sample(const float* logits, int ldl, int rows, int V, float temperature, int top_k, float top_p,
                         uint64_t seed, uint64_t step, const int32_t* step_ptr, uint8_t* finished, int pad_id, int eos_id,
                         int32_t* out, int32_t* history, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(sample_kernel, dim3(rows), dim3(256), 0, st, logits, ldl, V, 1.0f / temperature, top_k, top_p, seed, step,
                     step_ptr, finished, pad_id, eos_id, out, history);
  return hipGetLastError();
}
