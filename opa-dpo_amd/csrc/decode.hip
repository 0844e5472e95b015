// On-policy rollout kernels: single-token attention over a KV cache and the on-device
// temperature -> top-k -> top-p -> multinomial sampler (HF logits-processor order; the reference
// calls policy.generate(do_sample=True, top_k=30, top_p=0.95): online_generator.py:292-309).
#include <stdlib.h>
#include "common.h"
#include "kernels.h"
#include <algorithm>

namespace {

// ---- single-token attention over the KV cache ("flash decoding") ----------------------------------------------------
// Cache layout is head-major: k_cache / v_cache [B, nh, max_ctx, HD], so one (sequence, head) is a contiguous stream
// of HD*2-byte rows -> every wave load instruction covers 1 KiB of consecutive bytes.  HBM-bound: 2*HD*2 bytes per key.
// grid = (nh, B, splits), NW waves.  HD/8 lanes share one key (16 B of K and of V each), so a wave covers 64/(HD/8)
// keys per step and the block NW x that; U steps are issued back to back (U K-loads + U V-loads in flight per lane).
// Each lane group keeps an online softmax (running max / sum / 8 output dims per lane); groups merge through LDS.
// NW = 4 when B*nh alone oversubscribes the 256 CUs, 16 (one fat block per CU, 128 KiB of loads in flight) otherwise.
// splits > 1 (B*nh < 256): the key range is cut into pieces, every block writes (max, sum, out[HD]) to the workspace
// and attn_decode_merge_kernel combines them (a kernel boundary is the cheapest device-wide release/acquire here: a
// per-block agent-scope fence costs an L2 write-back per block on a multi-XCD part).
// FUSE (decode step of the generator): q points at the [q|k|v] rows of the NEW token straight out of the projection.  Every
// block rotates its q chunk itself (RoPE at position pos = ctx_ptr[0]; a lane fetches its 8 elements and the 8 of its
// rotation partner 64 columns away); the block whose key range contains pos also rotates the new k, appends k and v to the
// caches and folds that key into its online softmax from registers (the loop skips slot pos, so there is no
// write -> read dependence through memory inside the launch).  Replaces rope_kv_append + attn_decode: one launch less per
// layer and step.  Rounding points are those of the two-kernel path (rotated q and k rounded to bf16 before use).
template <int HD, int NW, bool FUSE = false>
__global__ __launch_bounds__(NW * 64) void attn_decode_kernel(const bf16_t* q, int ldq, const bf16_t* kc, const bf16_t* vc,
                                                               bf16_t* o, const uint8_t* key_mask, int nh, int ctx_arg,
                                                               const int32_t* ctx_ptr, int max_ctx, float scale_log2e, float* ws_part,
                                                               const float* cosb = nullptr, const float* sinb = nullptr,
                                                               bf16_t* kc_w = nullptr, bf16_t* vc_w = nullptr, int kv_nt = 0) {
  constexpr int LPK = HD / 8;            // lanes per key
  constexpr int KPW = 64 / LPK;          // keys per wave step
  constexpr int NG = NW * KPW;           // lane groups (= keys per block step)
  constexpr int U = 4;
  __shared__ float sm_m[NG], sm_l[NG];
  __shared__ __attribute__((aligned(16))) float sm_acc[NG][HD];
  const int ctx = ctx_ptr ? min(ctx_ptr[0] + 1, max_ctx) : ctx_arg;   // device-resident: keys 0..pos (graph replay)
  const int h = blockIdx.x, b = blockIdx.y, split = blockIdx.z, splits = gridDim.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int g = lane / LPK, d = lane % LPK;
  const int per = (((ctx + splits - 1) / splits) + NG - 1) / NG * NG;
  const int s0 = split * per, s1 = min(ctx, s0 + per);

  float qf[8];
  const int pos = FUSE ? ctx - 1 : -1;                 // slot of the token this step appends
  // rotate 8 elements of a head row: own chunk d*8.. and the partner chunk 64 columns away (first half: x1 c - x2 s, second: x2 c + x1 s)
  auto rope8 = [&](const bf16_t* row, float (&out)[8]) {
    constexpr int half = HD / 2;
    const bool lo = d * 8 < half;
    const int i0 = lo ? d * 8 : d * 8 - half;
    float own[8], oth[8];
    unpack8(*(const uint4*)(row + d * 8), own);
    unpack8(*(const uint4*)(row + (lo ? d * 8 + half : d * 8 - half)), oth);
    const float* cp = cosb + (size_t)pos * half + i0;
    const float* sp = sinb + (size_t)pos * half + i0;
#pragma unroll
    for (int e = 0; e < 8; ++e) out[e] = lo ? own[e] * cp[e] - oth[e] * sp[e] : own[e] * cp[e] + oth[e] * sp[e];
    const uint4 rb = pack8(out);                       // the two-kernel path stores the rotated row as bf16
    unpack8(rb, out);
  };
  if constexpr (FUSE) rope8(q + (size_t)b * ldq + h * HD, qf);
  else unpack8(*(const uint4*)(q + (size_t)b * ldq + h * HD + d * 8), qf);
#pragma unroll
  for (int e = 0; e < 8; ++e) qf[e] *= scale_log2e;
  const size_t seq = ((size_t)b * nh + h) * max_ctx;
  const bf16_t* kb = kc + seq * HD + d * 8;
  const bf16_t* vb = vc + seq * HD + d * 8;
  const uint8_t* km = key_mask ? key_mask + (size_t)b * max_ctx : nullptr;

  float m = -1.0e30f, l = 0.f, acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = 0.f;
  for (int jb = s0 + wave * KPW; jb < s1; jb += NG * U) {
    uint4 kk[U], vv[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int j = jb + u * NG + g;
      ok[u] = j < s1 && (!km || km[j]) && j != pos;
      kk[u] = make_uint4(0, 0, 0, 0);
      vv[u] = make_uint4(0, 0, 0, 0);
      if (j < s1 && j != pos) {          // slots >= ctx are never read (uninitialised memory may hold NaN bit patterns)
        // the KV cache is read exactly once per decode step: kv_nt streams it past the L2s (OPADPO_DEC_NT, A/B switch)
        typedef __attribute__((ext_vector_type(4))) unsigned u4n_t;
        if (kv_nt) {
          const u4n_t a = __builtin_nontemporal_load((const u4n_t*)(kb + (size_t)j * HD)), c = __builtin_nontemporal_load((const u4n_t*)(vb + (size_t)j * HD));
          kk[u] = make_uint4(a.x, a.y, a.z, a.w); vv[u] = make_uint4(c.x, c.y, c.z, c.w);
        } else {
          kk[u] = *(const uint4*)(kb + (size_t)j * HD);
          vv[u] = *(const uint4*)(vb + (size_t)j * HD);
        }
      }
    }
    float sc[U];
    float mn = m;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float kf[8];
      unpack8(kk[u], kf);
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) a += kf[e] * qf[e];
#pragma unroll
      for (int off = 1; off < LPK; off <<= 1) a += __shfl_xor(a, off, 64);
      sc[u] = ok[u] ? a : -1.0e30f;
      mn = fmaxf(mn, sc[u]);
    }
    const float alpha = fast_exp2(m - mn);
    m = mn;
    l *= alpha;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= alpha;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float pj = ok[u] ? fast_exp2(sc[u] - m) : 0.f;
      float vf[8];
      unpack8(vv[u], vf);
      l += pj;
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[e] += pj * vf[e];
    }
  }
  if constexpr (FUSE) {
    if (pos >= s0 && pos < s1 && wave == 0 && g == 0) {          // the new token's key: one lane group, from registers
      const size_t H = (size_t)nh * HD;
      const bf16_t* krow = q + (size_t)b * ldq + H + h * HD;
      float kf[8], vf[8];
      rope8(krow, kf);
      const uint4 vraw = *(const uint4*)(krow + H + d * 8);
      unpack8(vraw, vf);
      *(uint4*)(kc_w + (seq + pos) * HD + d * 8) = pack8(kf);
      *(uint4*)(vc_w + (seq + pos) * HD + d * 8) = vraw;
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) a += kf[e] * qf[e];
#pragma unroll
      for (int off = 1; off < LPK; off <<= 1) a += __shfl_xor(a, off, 64);
      if (!km || km[pos]) {
        const float mn = fmaxf(m, a);
        const float alpha = fast_exp2(m - mn), pj = fast_exp2(a - mn);
        m = mn;
        l = l * alpha + pj;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = acc[e] * alpha + pj * vf[e];
      }
    }
  }
  const int grp = wave * KPW + g;
  if (d == 0) { sm_m[grp] = m; sm_l[grp] = l; }
  *(float4*)&sm_acc[grp][d * 8] = make_float4(acc[0], acc[1], acc[2], acc[3]);
  *(float4*)&sm_acc[grp][d * 8 + 4] = make_float4(acc[4], acc[5], acc[6], acc[7]);
  __syncthreads();
  if (tid < HD) {
    float M = -1.0e30f, Lsum = 0.f, O = 0.f;
#pragma unroll 4
    for (int i = 0; i < NG; ++i) M = fmaxf(M, sm_m[i]);
#pragma unroll 4
    for (int i = 0; i < NG; ++i) {
      const float w = fast_exp2(sm_m[i] - M);
      Lsum += sm_l[i] * w;
      O += sm_acc[i][tid] * w;
    }
    const int bh = b * nh + h;
    if (splits == 1) {
      o[(size_t)bh * HD + tid] = f2bf(Lsum > 0.f ? O / Lsum : 0.f);
    } else {
      float* part = ws_part + ((size_t)bh * splits + split) * (HD + 2);
      part[2 + tid] = O;
      if (tid == 0) { part[0] = M; part[1] = Lsum; }
    }
  }
}

// one block per (sequence, head): out = sum_s w_s O_s / sum_s w_s L_s, w_s = 2^(M_s - max M)
__global__ void attn_decode_merge_kernel(const float* ws_part, bf16_t* o, int splits, int HD) {
  const int bh = blockIdx.x, tid = threadIdx.x;
  const float* p0 = ws_part + (size_t)bh * splits * (HD + 2);
  float MM = -1.0e30f;
  for (int i = 0; i < splits; ++i) MM = fmaxf(MM, p0[(size_t)i * (HD + 2)]);
  float LL = 0.f, OO = 0.f;
  for (int i = 0; i < splits; ++i) {
    const float* pi = p0 + (size_t)i * (HD + 2);
    const float w = fast_exp2(pi[0] - MM);
    LL += pi[1] * w;
    OO += pi[2 + tid] * w;
  }
  o[(size_t)bh * HD + tid] = f2bf(LL > 0.f ? OO / LL : 0.f);
}

// ---- decode-step RoPE + KV-cache append: q rotated in place, k rotated into k_cache[b,h,pos,:], v copied into
// v_cache[b,h,pos,:] (pos = pos_ptr[0], device-resident).  qkv rows are [q(H) | k(H) | v(H)].  One launch replaces
// rope + 2x(gather, scatter).
__global__ __launch_bounds__(256) void rope_kv_append_kernel(bf16_t* qkv, int ld, const float* cosb, const float* sinb, bf16_t* kc,
                                                              bf16_t* vc, int B, int nh, int hd, const int32_t* pos_ptr, int max_ctx) {
  const int half = hd / 2, cpr = half / 8;        // chunks (8 pairs) per head
  const int total = B * nh * cpr;
  const int pos = pos_ptr[0];
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int b = idx / (nh * cpr), rem = idx % (nh * cpr);
    const int h = rem / cpr, i0 = (rem % cpr) * 8;
    const float* cp = cosb + (size_t)pos * half + i0;
    const float* sp = sinb + (size_t)pos * half + i0;
    const size_t H = (size_t)nh * hd;
    bf16_t* qp = qkv + (size_t)b * ld + h * hd;
    const bf16_t* kp = qp + H;
    const bf16_t* vp = qp + 2 * H;
    const size_t crow = (((size_t)b * nh + h) * max_ctx + pos) * hd;
    float x1[8], x2[8], o1[8], o2[8];
    unpack8(*(const uint4*)(qp + i0), x1);
    unpack8(*(const uint4*)(qp + half + i0), x2);
#pragma unroll
    for (int j = 0; j < 8; ++j) { o1[j] = x1[j] * cp[j] - x2[j] * sp[j]; o2[j] = x2[j] * cp[j] + x1[j] * sp[j]; }
    *(uint4*)(qp + i0) = pack8(o1);
    *(uint4*)(qp + half + i0) = pack8(o2);
    unpack8(*(const uint4*)(kp + i0), x1);
    unpack8(*(const uint4*)(kp + half + i0), x2);
#pragma unroll
    for (int j = 0; j < 8; ++j) { o1[j] = x1[j] * cp[j] - x2[j] * sp[j]; o2[j] = x2[j] * cp[j] + x1[j] * sp[j]; }
    *(uint4*)(kc + crow + i0) = pack8(o1);
    *(uint4*)(kc + crow + half + i0) = pack8(o2);
    *(uint4*)(vc + crow + i0) = *(const uint4*)(vp + i0);
    *(uint4*)(vc + crow + half + i0) = *(const uint4*)(vp + half + i0);
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
__global__ __launch_bounds__(256) void sample_kernel_generic(const float* logits, int ldl, int V, float inv_temp, int top_k, float top_p,
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


// ---- fast sampler: the whole row lives in LDS ------------------------------------------------------------------------
// 1024 threads per row; the temperature-scaled row (V <= 32768 floats = 128 KiB of the CU's 160 KiB LDS) is read from
// memory ONCE, every later pass is a conflict-free stride-1 LDS sweep (the generic kernel makes ~66 passes over memory).
// Same filter semantics as the generic kernel:
//   top-k : 4-level radix select (8-bit digits of the monotone key) with an LDS histogram of counts ->
//           thr = key of the k-th largest logit (ties kept);
//   top-p : the same descent over a histogram of probability MASS -> largest t2 with mass(thr <= key < t2) <= (1-p)*Z.
//           Mass is fixed-point (2^-40 of the row maximum) in uint64, so LDS atomics are exact and order-independent:
//           the draw is deterministic for (seed, step, row);
//   draw  : block-wide exclusive scan of the per-thread kept mass, the owner thread walks its <= 32 values.
constexpr int SAMP_T = 1024, SAMP_W = SAMP_T / 64, SAMP_MAXV = 32768;
constexpr float SAMP_FIX = 1099511627776.0f;   // 2^40

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint64_t)__shfl_xor((long long)v, o, 64);
  return v;
}

// digit choice for one radix level, executed by wave 0: hist[256] (uint64) -> LDS result {digit, carried sum}.
//   DESC (top-k):  largest digit d with base + sum(hist[d..255]) >= need;   carry = base + sum(hist[d+1..255])
//   !DESC (top-p): largest digit d with base + sum(hist[0..d-1]) <= need;   carry = base + sum(hist[0..d-1])
template <bool DESC>
__device__ __forceinline__ void pick_digit(const uint64_t* hist, uint64_t base, uint64_t need, int lane, uint32_t* out_digit,
                                           uint64_t* out_carry) {
  uint64_t c[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = hist[lane * 4 + i];
  const uint64_t s = c[0] + c[1] + c[2] + c[3];
  uint64_t incl = s;                       // DESC: sum over lanes >= lane ; else: sum over lanes <= lane
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t t = (uint64_t)(DESC ? __shfl_down((long long)incl, off, 64) : __shfl_up((long long)incl, off, 64));
    if (DESC ? (lane + off < 64) : (lane >= off)) incl += t;
  }
  const uint64_t outside = base + incl - s;   // DESC: bins above this lane's 4 ; else: bins below this lane's 4
  uint64_t S[4];                              // DESC: base + sum(hist[bin..255]) ; else: base + sum(hist[0..bin-1])
  if (DESC) {
    S[3] = outside + c[3]; S[2] = S[3] + c[2]; S[1] = S[2] + c[1]; S[0] = S[1] + c[0];
  } else {
    S[0] = outside; S[1] = S[0] + c[0]; S[2] = S[1] + c[1]; S[3] = S[2] + c[2];
  }
  int cnt = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) cnt += DESC ? (S[i] >= need) : (S[i] <= need);
  int total = cnt;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
  const int dsel = max(total - 1, 0);          // satisfying bins form the prefix [0..dsel]
  if (lane == (dsel >> 2)) {
    const int i = dsel & 3;
    *out_digit = (uint32_t)dsel;
    *out_carry = DESC ? (S[i] - c[i]) : S[i];
  }
}

__global__ __launch_bounds__(SAMP_T) void sample_kernel_fast(const float* logits, int ldl, int V, float inv_temp, int top_k, float top_p,
                                                             uint64_t seed, uint64_t step_arg, const int32_t* step_ptr, uint8_t* finished,
                                                             int pad_id, int eos_id, int32_t* out, int32_t* history, int compact) {
  extern __shared__ __attribute__((aligned(16))) float zs[];     // [V] scaled logits
  const uint64_t step = step_ptr ? (uint64_t)step_ptr[0] : step_arg;
  __shared__ uint64_t hist[256];
  __shared__ uint64_t wsum[SAMP_W];
  __shared__ float wmax[SAMP_W];
  __shared__ uint32_t s_digit;
  __shared__ uint64_t s_carry;
  const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (finished && finished[row]) {
    if (tid == 0) {
      out[row] = pad_id;
      if (history) history[step * gridDim.x + row] = pad_id;
    }
    return;
  }
  const float4* z4 = (const float4*)(logits + (size_t)row * ldl);
  float mx = -3.0e38f;
  for (int gi = tid; gi < (V >> 2); gi += SAMP_T) {
    float4 t = z4[gi];
    t.x *= inv_temp; t.y *= inv_temp; t.z *= inv_temp; t.w *= inv_temp;
    *(float4*)&zs[gi * 4] = t;
    mx = fmaxf(fmaxf(mx, fmaxf(t.x, t.y)), fmaxf(t.z, t.w));
  }
  mx = wave_max(mx);
  if (lane == 0) wmax[wave] = mx;
  __syncthreads();
  mx = wmax[0];
#pragma unroll
  for (int i = 1; i < SAMP_W; ++i) mx = fmaxf(mx, wmax[i]);
  auto qmass = [&](float x) -> uint64_t { return (uint64_t)(__expf(x - mx) * SAMP_FIX); };   // exp(-inf) = 0
  auto block_sum = [&](uint64_t x) -> uint64_t {
    x = wave_sum_u64(x);
    __syncthreads();
    if (lane == 0) wsum[wave] = x;
    __syncthreads();
    uint64_t t = 0;
#pragma unroll
    for (int i = 0; i < SAMP_W; ++i) t += wsum[i];
    return t;
  };

  uint32_t thr = 0;   // keep tokens with key >= thr
  if (top_k > 0 && top_k < V) {
    uint32_t prefix = 0, pmask = 0;
    uint64_t above = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int base = 0; base < V; base += SAMP_T) {
        const int i = base + tid;
        const uint32_t k = i < V ? fkey(zs[i]) : 0u;
        bool mine = i < V && (k & pmask) == prefix;
        const uint32_t dg = (k >> shift) & 255u;
        // wave-aggregated counting for crowded bins (the top byte of a float key takes only a few values)
#pragma unroll 1
        for (int it = 0; it < 3; ++it) {
          const uint64_t act = __ballot(mine);
          if (!act) break;
          const int lead = __ffsll((long long)act) - 1;
          const uint32_t ldg = __shfl(dg, lead, 64);
          const uint64_t same = __ballot(mine && dg == ldg);
          if (lane == lead) atomicAdd((unsigned long long*)&hist[ldg], (unsigned long long)__popcll(same));
          if (dg == ldg) mine = false;
        }
        if (mine) atomicAdd((unsigned long long*)&hist[dg], 1ull);
      }
      __syncthreads();
      if (wave == 0) pick_digit<true>(hist, above, (uint64_t)top_k, lane, &s_digit, &s_carry);
      __syncthreads();
      prefix |= s_digit << shift;
      pmask |= 255u << shift;
      above = s_carry;
      __syncthreads();
    }
    thr = prefix;
    // COMPACT tail (round 4): after the top-k threshold the kept set is a few dozen tokens, yet the top-p descent, the mass sums and the draw
    // swept the whole vocabulary seven more times.  The kept tokens are gathered into an LDS list (one more sweep); when they are at most 64,
    // wave 0 finishes alone, one token per lane, with the SAME integer arithmetic: the masses are exact fixed-point integers, so their sums do not
    // depend on the order; the top-p threshold "largest t with mass(thr <= key < t) <= budget" is the smallest kept key whose mass of keys <= it
    // exceeds the budget (none: 2^32 - 1); the draw walks the kept tokens in the order (owner thread = index mod 1024, index) of the full-sweep
    // form.  Same token for every (seed, step, row) as the sweeps (tests/test_ops_gpu.py::test_sampler_compact_tail_is_exact); more than 64 kept
    // tokens (ties at the threshold) fall through to the sweeps.
    if (compact) {
      __shared__ int s_n;
      __shared__ int s_idx[64];
      if (tid == 0) s_n = 0;
      __syncthreads();
      for (int i = tid; i < V; i += SAMP_T)
        if (fkey(zs[i]) >= thr) {
          const int pos = atomicAdd(&s_n, 1);
          if (pos < 64) s_idx[pos] = i;
        }
      __syncthreads();
      const int n = s_n;
      if (n <= 64) {
        if (wave != 0) return;
        const bool valid = lane < n;
        const int idx = valid ? s_idx[lane] : 0x7fffffff;
        const float x = valid ? zs[idx] : 0.f;
        const uint32_t key = valid ? fkey(x) : 0u;
        const uint64_t qm = valid ? qmass(x) : 0ull;
        uint32_t thr2 = thr;
        if (top_p < 1.0f) {
          const uint64_t zk = wave_sum_u64(qm);
          const uint64_t budget = (uint64_t)((1.0 - (double)top_p) * (double)zk);
          uint64_t below_incl = 0;                         // mass of the kept keys <= this lane's key
          for (int j = 0; j < n; ++j) {
            const uint32_t kj = (uint32_t)__shfl((int)key, j, 64);
            const uint64_t qj = (uint64_t)__shfl((long long)qm, j, 64);
            if (kj <= key) below_incl += qj;
          }
          uint32_t t2 = (valid && below_incl > budget) ? key : 0xffffffffu;
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) t2 = min(t2, (uint32_t)__shfl_xor((int)t2, o, 64));
          thr2 = max(thr, t2);
        }
        const bool kept = valid && key >= thr2;
        const uint64_t qk = kept ? qm : 0ull;
        const uint64_t total = wave_sum_u64(qk);
        const uint32_t ord = ((uint32_t)(idx & (SAMP_T - 1)) << 20) | (uint32_t)(idx >> 10);      // (owner thread, then index): V <= 32768 -> index / 1024 < 32
        uint64_t before = 0;                               // kept mass ahead of this lane's token in the draw order
        for (int j = 0; j < n; ++j) {
          const uint32_t oj = (uint32_t)__shfl((int)ord, j, 64);
          const uint64_t qj = (uint64_t)__shfl((long long)qk, j, 64);
          if (oj < ord) before += qj;
        }
        const uint64_t r = mix64(mix64(seed ^ (step * 0xD1B54A32D192ED03ull)) + (uint64_t)row);
        const uint64_t r24 = r >> 40;
        const uint64_t target = (total >> 24) * r24 + (((total & 0xffffffull) * r24) >> 24);
        if (total == 0) {
          if (lane == 0) {
            out[row] = pad_id;
            if (history) history[step * gridDim.x + row] = pad_id;
          }
          return;
        }
        if (qk > 0 && before <= target && target < before + qk) {
          out[row] = idx;
          if (history) history[step * gridDim.x + row] = idx;
          if (finished && eos_id >= 0 && idx == eos_id) finished[row] = 1;
        }
        return;
      }
    }
  }
  if (top_p < 1.0f) {
    uint64_t zk = 0;
    for (int i = tid; i < V; i += SAMP_T) {
      const float x = zs[i];
      if (fkey(x) >= thr) zk += qmass(x);
    }
    zk = block_sum(zk);
    const uint64_t budget = (uint64_t)((1.0 - (double)top_p) * (double)zk);
    uint32_t prefix = 0, pmask = 0;
    uint64_t below = 0;
    for (int shift = 24; shift >= 0; shift -= 8) {
      if (tid < 256) hist[tid] = 0;
      __syncthreads();
      for (int i = tid; i < V; i += SAMP_T) {
        const float x = zs[i];
        const uint32_t k = fkey(x);
        if (k >= thr && (k & pmask) == prefix) {
          const uint64_t qm = qmass(x);
          if (qm) atomicAdd((unsigned long long*)&hist[(k >> shift) & 255u], (unsigned long long)qm);
        }
      }
      __syncthreads();
      if (wave == 0) pick_digit<false>(hist, below, budget, lane, &s_digit, &s_carry);
      __syncthreads();
      prefix |= s_digit << shift;
      pmask |= 255u << shift;
      below = s_carry;
      __syncthreads();
    }
    thr = max(thr, prefix);
  }
  // multinomial over the kept set; cumulative order = (thread, then index) — any fixed order is a valid draw
  uint64_t local = 0;
  for (int i = tid; i < V; i += SAMP_T) {
    const float x = zs[i];
    if (fkey(x) >= thr) local += qmass(x);
  }
  uint64_t incl = local;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint64_t t = (uint64_t)__shfl_up((long long)incl, off, 64);
    if (lane >= off) incl += t;
  }
  __syncthreads();
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  uint64_t wave_base = 0, total = 0;
#pragma unroll
  for (int i = 0; i < SAMP_W; ++i) {
    if (i < wave) wave_base += wsum[i];
    total += wsum[i];
  }
  const uint64_t r = mix64(mix64(seed ^ (step * 0xD1B54A32D192ED03ull)) + (uint64_t)row);
  const uint64_t r24 = r >> 40;                                   // u = r24 / 2^24 in [0, 1)
  const uint64_t target = (total >> 24) * r24 + (((total & 0xffffffull) * r24) >> 24);   // floor(u * total) < total
  const uint64_t start = wave_base + incl - local;
  if (total == 0) {
    if (tid == 0) {
      out[row] = pad_id;
      if (history) history[step * gridDim.x + row] = pad_id;
    }
    return;
  }
  if (local > 0 && start <= target && target < start + local) {
    uint64_t run = start;
    int pick = pad_id;
    for (int i = tid; i < V && run <= target; i += SAMP_T) {
      const float x = zs[i];
      if (fkey(x) >= thr) {
        const uint64_t qm = qmass(x);
        if (qm) pick = i;
        run += qm;
      }
    }
    out[row] = pick;
    if (history) history[step * gridDim.x + row] = pick;
    if (finished && eos_id >= 0 && pick == eos_id) finished[row] = 1;
  }
}

}  // namespace

static int attn_decode_splits(int B, int nh, int max_ctx) {
  static const int force = getenv("OPADPO_ATTN_DEC_SPLITS") ? atoi(getenv("OPADPO_ATTN_DEC_SPLITS")) : 0;      // diagnostics
  if (force > 0) return std::min(force, std::max(1, max_ctx / 256));
  // 257..767 (sequence, head) pairs (9..23 sequences at 32 heads): two key ranges per pair, run as 4-wave workgroups (>= 512 of them) - measured
  // against one 16-wave workgroup per pair: decode step B = 12 4.47 -> 4.27 ms, B = 16 4.58 -> 4.40; from 768 pairs on 4-wave workgroups without a
  // split (B = 24: 5.34 -> 5.12); up to 256 pairs the 16-wave form stays (B = 8: 3.80 against 4.07 with a split)
  if (B * nh >= 768) {
    // 4-wave workgroups, 1024 of them resident on the chip: a partly filled last round costs a whole one (attention per layer 65.6 us at 32
    // sequences, 106 at 40, 127.6 at 64), so two key ranges per pair are used where they cut the rounds by a fifth or more - 33..48 sequences at 32
    // heads: decode step B = 36 6.62 -> 6.18 ms, B = 40 6.76 -> 6.34, B = 48 7.09 -> 6.79; B = 56 / 64 stay unsplit (7.38 / 7.70 against 7.56 / 7.98)
    const int pairs = B * nh, r1 = (pairs + 1023) / 1024, r2x2 = (2 * pairs + 1023) / 1024;
    return (r2x2 * 5 <= r1 * 8 && max_ctx >= 512) ? 2 : 1;
  }
  if (B * nh > 256) return std::min(2, std::max(1, max_ctx / 256));
  // up to 256 pairs: as many key ranges as keep the 16-wave workgroups within ONE round of the 256 CUs (rounded up, 5..7 sequences ran 320-448
  // workgroups in two rounds: attention 27.4 + 4.9 us merge per layer at B = 5 against 23.2 at B = 8; profiles/r04k_rollout_b*_kernel_stats.csv)
  int splits = std::max(1, 256 / (B * nh));
  splits = std::min(splits, std::max(1, max_ctx / 256));
  return std::max(1, std::min(splits, 16));
}
static bool attn_decode_fat(int B, int nh, int splits) {
  static const int fat_lim = getenv("OPADPO_ATTN_DEC_FAT") ? atoi(getenv("OPADPO_ATTN_DEC_FAT")) : 512;      // diagnostics: 16-wave workgroups below this many of them
  return B * nh * splits < fat_lim;
}
// the KV cache is read exactly once per decode step: non-temporal loads (round 4; same-box B = 64: 9.24 -> 8.93 ms per step, B = 8 neutral; OPADPO_DEC_NT=0 switches back)
static int dec_nt() { static const int v = getenv("OPADPO_DEC_NT") ? atoi(getenv("OPADPO_DEC_NT")) : 1; return v; }
size_t attn_decode_workspace_bytes(int B, int nh, int hd, int max_ctx) {
  const int splits = attn_decode_splits(B, nh, max_ctx);
  return splits == 1 ? 0 : (size_t)B * nh * splits * (hd + 2) * 4;
}

hipError_t launch_attn_decode_fused(const bf16_t* qkv, int ld, const float* cosb, const float* sinb, bf16_t* kc, bf16_t* vc, bf16_t* o,
                                    const uint8_t* key_mask, int B, int nh, int hd, const int32_t* pos_ptr, int max_ctx, float scale,
                                    void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  if ((hd != 64 && hd != 128) || !pos_ptr || ld % 8) return hipErrorInvalidValue;
  int splits = attn_decode_splits(B, nh, max_ctx);
  if (splits > 1 && (!workspace || workspace_bytes < attn_decode_workspace_bytes(B, nh, hd, max_ctx))) splits = 1;
  float* part = (float*)workspace;
  const float sl2 = scale * 1.4426950408889634f;
  const bool fat = attn_decode_fat(B, nh, splits);
  const dim3 gr(nh, B, splits);
#define ADF(HD_, NW_) hipLaunchKernelGGL((attn_decode_kernel<HD_, NW_, true>), gr, dim3(NW_ * 64), 0, st, qkv, ld, kc, vc, o, key_mask, nh, 0, pos_ptr, max_ctx, sl2, part, cosb, sinb, kc, vc, dec_nt())
  if (hd == 128) { if (fat) ADF(128, 16); else ADF(128, 4); }
  else           { if (fat) ADF(64, 16); else ADF(64, 4); }
#undef ADF
  if (splits > 1) hipLaunchKernelGGL(attn_decode_merge_kernel, dim3(B * nh), dim3(hd), 0, st, part, o, splits, hd);
  return hipGetLastError();
}

hipError_t launch_attn_decode(const bf16_t* q, const bf16_t* kc, const bf16_t* vc, bf16_t* o, const uint8_t* key_mask,
                              int B, int nh, int hd, int ctx, const int32_t* ctx_ptr, int max_ctx, int ldq, float scale,
                              void* workspace, size_t workspace_bytes, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  if (hd != 64 && hd != 128) return hipErrorInvalidValue;
  int splits = attn_decode_splits(B, nh, max_ctx);
  if (splits > 1 && (!workspace || workspace_bytes < attn_decode_workspace_bytes(B, nh, hd, max_ctx))) splits = 1;
  float* part = (float*)workspace;
  const float sl2 = scale * 1.4426950408889634f;
  const bool fat = attn_decode_fat(B, nh, splits);       // few blocks: 16 waves each
  const dim3 gr(nh, B, splits);
#define AD(HD_, NW_) hipLaunchKernelGGL((attn_decode_kernel<HD_, NW_>), gr, dim3(NW_ * 64), 0, st, q, ldq, kc, vc, o, key_mask, nh, ctx, ctx_ptr, max_ctx, sl2, part, nullptr, nullptr, nullptr, nullptr, dec_nt())
  if (hd == 128) { if (fat) AD(128, 16); else AD(128, 4); }
  else           { if (fat) AD(64, 16); else AD(64, 4); }
#undef AD
  if (splits > 1) hipLaunchKernelGGL(attn_decode_merge_kernel, dim3(B * nh), dim3(hd), 0, st, part, o, splits, hd);
  return hipGetLastError();
}

hipError_t launch_rope_kv_append(bf16_t* qkv, int ld, const float* cosb, const float* sinb, bf16_t* kc, bf16_t* vc, int B, int nh,
                                 int hd, const int32_t* pos_ptr, int max_ctx, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  if (hd % 16 || ld % 8 || !pos_ptr) return hipErrorInvalidValue;
  const int total = B * nh * (hd / 16);
  hipLaunchKernelGGL(rope_kv_append_kernel, dim3((total + 255) / 256), dim3(256), 0, st, qkv, ld, cosb, sinb, kc, vc, B, nh, hd, pos_ptr, max_ctx);
  return hipGetLastError();
}

// 1 (default): one-wave tail on the kept tokens after the top-k threshold; 0: the full vocabulary sweeps (identical draws; the exactness test's yardstick)
static const int env_sample_compact_ = getenv("OPADPO_SAMPLE_COMPACT") ? atoi(getenv("OPADPO_SAMPLE_COMPACT")) : 1;
static int g_sample_compact = env_sample_compact_;
void opadpo_set_sample_compact(int on) { g_sample_compact = on < 0 ? env_sample_compact_ : (on ? 1 : 0); }      // -1: the environment decides (as for attn64)
hipError_t launch_sample(const float* logits, int ldl, int rows, int V, float temperature, int top_k, float top_p,
                         uint64_t seed, uint64_t step, const int32_t* step_ptr, uint8_t* finished, int pad_id, int eos_id,
                         int32_t* out, int32_t* history, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  const bool fast = V % 4 == 0 && ldl % 4 == 0 && V <= SAMP_MAXV && ((uintptr_t)logits & 15) == 0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)sample_kernel_fast, hipFuncAttributeMaxDynamicSharedMemorySize, SAMP_MAXV * 4);
    attr_set = true;
  }
  const int compact = g_sample_compact;      // process switch (opadpo_set_flags use_tr bit 9 / OPADPO_SAMPLE_COMPACT=0 at load): the same for eager and captured launches
  if (fast)
    hipLaunchKernelGGL(sample_kernel_fast, dim3(rows), dim3(SAMP_T), (size_t)V * 4, st, logits, ldl, V, 1.0f / temperature, top_k, top_p, seed,
                       step, step_ptr, finished, pad_id, eos_id, out, history, compact);
  else
    hipLaunchKernelGGL(sample_kernel_generic, dim3(rows), dim3(256), 0, st, logits, ldl, V, 1.0f / temperature, top_k, top_p, seed,
                       step, step_ptr, finished, pad_id, eos_id, out, history);
  return hipGetLastError();
}
