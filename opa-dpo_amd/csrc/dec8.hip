// Decode projections for <= 8 token rows with the preceding LlamaRMSNorm folded in (round 6).
//
// Replaces, per decoder layer and decode step of a rollout with <= 8 sequences per device (BASELINE.json configs[4]: batch 64 over 8 GPUs = 8 per
// device; online_generator.py:292-309 -> HF generate -> LlamaDecoderLayer.forward): rmsnorm_fwd + q|k|v projection, rmsnorm_fwd + gate|up projection
// (+ SwiGLU), and the o / down projections with their residual adds - seven launches become five, and every projection becomes ONE round of
// persistent workgroups whose weight stream starts before anything else happens.  What the r05 profile showed at B = 8 (profiles/r05j_b8_by_grid.txt):
// two 4.9-us rmsnorm launches per layer moving nothing (8 % of the step), an o-projection streaming at 2.2 TB/s because its launch is 15 us short.
//
// Geometry (one workgroup per CU, 8 waves):
//   * the weight rows [b * rpw, (b + 1) * rpw) belong to workgroup b (rpw = ceil(rows / 256): 48 for q|k|v at 7B, 43 gate/up PAIRS, 16 for o / down) -
//     every CU streams the same number of bytes;
//   * wave w owns the K-slice [w * PER, (w + 1) * PER) k-steps of 64 for EVERY row of the workgroup.  Its activation fragments (<= 8 tokens x that
//     slice) therefore never change: they are built ONCE, in registers (PER x 4 VGPRs) - no LDS traffic and no activation re-reads in the loop (the
//     per-CU vector-memory path is what bounds these kernels: profiles/r04_decode_stream.txt);
//   * the 16x16x32 MFMA runs as two 8x8x32 products (rows 0..7 = 8 weight rows with k-chunks 0..3 of a 64-deep k-step, rows 8..15 = the same rows with
//     chunks 4..7; tokens likewise): one wave load covers 8 weight rows x 128 contiguous bytes = whole cache lines (gemm_nt_skinny8_kernel's trick);
//   * weights: non-temporal 16-byte loads, double-buffered in registers one batch (<= 14 k-steps of one 8-row group) ahead of the MFMAs;
//   * a wave's partial 8x8 blocks go to LDS once per group; after ONE barrier the 8 K-slices of every output are added in slice order (fixed order:
//     deterministic, and a token's result does not depend on the other tokens of the batch).
// RMSNorm fold (norm_w != NULL): every wave reads its K-slice of the <= 8 residual rows (fp32 or bf16) while its first weight batches are in flight,
// the row sums of squares meet in LDS (8 slices added in slice order), n = bf16(x * rstd * w) - rmsnorm_fwd_kernel's formula - becomes the fragment.
// MODE 0: bf16 C[M, N]; MODE 1: fp32 C[M, N] = product + fp32 / bf16 residual R (o / down); MODE 2: OPADPO_ACT_SWIGLU_PAIR weight layout (rows per 128 =
// [64 gate | 64 up]) -> bf16 C[M, N / 2] = silu(gate) * up on the bf16-rounded sums (silu_mul_fwd_kernel's formula).
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

struct Dec8Args {
  const void* x; int x_f32; int ldx;       // activations [M, K]
  const bf16_t* norm_w; float eps;         // RMSNorm weight [K] (nullable)
  const bf16_t* W; int ldw;                // [N, K]
  void* C; int ldc;
  const void* R; int r_f32; int ldr;       // MODE 1 residual (nullable)
  int M, N, K, rpw;
};

constexpr int DEC8_MAXG = 16;              // 8-row groups per workgroup (LDS: groups x NR x 8 waves x 1 KiB)

template <int MODE, int PER, int NB, bool NORM>
__global__ __launch_bounds__(512) void gemm_nt_dec8_kernel(Dec8Args p) {
  constexpr int NR = MODE == 2 ? 2 : 1;
  constexpr int U = (PER + NB - 1) / NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const part_ss = (float*)smem;                               // [8 waves][8 tokens]
  f32x4_t* const red = (f32x4_t*)(smem + 256);                      // [group][NR][wave][64 lanes]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, c = lane >> 4;
  const int r8 = r & 7, kc = ((r >> 3) * 4 + c) * 8;                 // weight row / token of the lane within the 8, first element of its k-chunk inside a k-step
  const int rows_all = MODE == 2 ? p.N / 2 : p.N;                    // MODE 2: gate / up pairs
  const int n_lo = blockIdx.x * p.rpw, n_hi = min(n_lo + p.rpw, rows_all);
  const int G = (n_hi - n_lo + 7) >> 3;
  const int ns = p.K >> 6;
  const int ks0 = wave * PER;                                        // this wave's K-slice: k-steps [ks0, ks0 + PER) (those < ns exist)

  // weight row of (group g, lane): MODE 2 pair q -> gate row (q / 64) * 128 + q % 64, up row 64 further
  auto wrow = [&](int g, int i) -> long long {
    const int q = n_lo + g * 8 + r8;
    if (q >= n_hi) return -1;
    if (MODE == 2) return (long long)(q >> 6) * 128 + (q & 63) + i * 64;
    return q;
  };
  u32x4_t wb[2][U][NR];
  auto load_batch = [&](auto BUF, int g, auto H) {
    constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const long long row = wrow(g, i);
      const bf16_t* wp = p.W + (size_t)(row < 0 ? 0 : row) * p.ldw + kc;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int s = h * U + u;
        wb[buf][u][i] = u32x4_t{0, 0, 0, 0};
        if (s < PER && row >= 0 && ks0 + s < ns) wb[buf][u][i] = __builtin_nontemporal_load((const u32x4_t*)(wp + (size_t)(ks0 + s) * 64));
      }
    }
  };
  // the weight stream starts first: batch 0 (and 1) of this wave are in flight under the activation set-up
  if (G > 0) load_batch(std::integral_constant<int, 0>{}, 0, std::integral_constant<int, 0>{});

  // ---- activation fragments of this wave's K-slice, built once ----
  u32x4_t xa[PER];
  {
    const bool tok_ok = r8 < p.M;
    if constexpr (NORM) {
      float xf[PER][8];
      float ss = 0.f;
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        const bool live = tok_ok && ks0 + s < ns;
        const size_t off = (size_t)r8 * p.ldx + (size_t)(ks0 + s) * 64 + kc;
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[s][e] = 0.f;
        if (live) {
          if (p.x_f32) {
            const float4 a = *(const float4*)((const float*)p.x + off), b = *(const float4*)((const float*)p.x + off + 4);
            xf[s][0] = a.x; xf[s][1] = a.y; xf[s][2] = a.z; xf[s][3] = a.w; xf[s][4] = b.x; xf[s][5] = b.y; xf[s][6] = b.z; xf[s][7] = b.w;
          } else {
            unpack8(*(const uint4*)((const bf16_t*)p.x + off), xf[s]);
          }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += xf[s][e] * xf[s][e];
      }
      // the 8 lanes of a token (lane bits 3, 4, 5) -> the wave's slice sum; the 8 slices meet in LDS and are added in slice order
      ss += __shfl_xor(ss, 8, 64);
      ss += __shfl_xor(ss, 16, 64);
      ss += __shfl_xor(ss, 32, 64);
      if (lane < 8) part_ss[wave * 8 + lane] = ss;
      __syncthreads();
      float tot = part_ss[r8];
#pragma unroll
      for (int w2 = 1; w2 < 8; ++w2) tot += part_ss[w2 * 8 + r8];
      const float rs = rsqrtf(tot / (float)p.K + p.eps);
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        float g8[8];
        uint4 gw = make_uint4(0u, 0u, 0u, 0u);
        if (ks0 + s < ns) gw = *(const uint4*)(p.norm_w + (size_t)(ks0 + s) * 64 + kc);
        unpack8(gw, g8);
#pragma unroll
        for (int e = 0; e < 8; ++e) xf[s][e] = xf[s][e] * rs * g8[e];
        const uint4 o = pack8(xf[s]);
        xa[s] = u32x4_t{o.x, o.y, o.z, o.w};
      }
    } else {
#pragma unroll
      for (int s = 0; s < PER; ++s) {
        xa[s] = u32x4_t{0, 0, 0, 0};
        if (tok_ok && ks0 + s < ns) xa[s] = *(const u32x4_t*)((const bf16_t*)p.x + (size_t)r8 * p.ldx + (size_t)(ks0 + s) * 64 + kc);
      }
    }
  }

  // ---- the weight stream: batches of one group, double-buffered ----
  f32x4_t acc[NR];
  auto compute = [&](auto BUF, int g, auto H) {
    constexpr int buf = decltype(BUF)::value, h = decltype(H)::value;
    if (h == 0) {
#pragma unroll
      for (int i = 0; i < NR; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int s = h * U + u;
      if (s < PER) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&wb[buf][u][i], *(const bf16x8_t*)&xa[s < PER ? s : 0], acc[i], 0, 0, 0);
      }
    }
    if (h == NB - 1) {
#pragma unroll
      for (int i = 0; i < NR; ++i) red[((g * NR + i) * 8 + wave) * 64 + lane] = acc[i];
    }
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  if constexpr (NB == 1) {
    // buffers alternate between consecutive groups
    for (int g = 0; g < G; g += 2) {
      if (g + 1 < G) load_batch(I1{}, g + 1, I0{});
      compute(I0{}, g, I0{});
      if (g + 1 < G) {
        if (g + 2 < G) load_batch(I0{}, g + 2, I0{});
        compute(I1{}, g + 1, I0{});
      }
    }
  } else {
    // two batches per group: buffer 0 = first half of the slice, buffer 1 = second half
    for (int g = 0; g < G; ++g) {
      load_batch(I1{}, g, I1{});
      compute(I0{}, g, I0{});
      if (g + 1 < G) load_batch(I0{}, g + 1, I0{});
      compute(I1{}, g, I1{});
    }
  }
  __syncthreads();

  // ---- the 8 K-slices of every output, added in slice order; lane (token j < 8, half g2 < 2) and lane + 40 hold the low / high k-chunk halves ----
  const int slot = tid & 31, gi0 = tid >> 5;                        // 16 groups per pass
  const int j = slot & 7, g2 = slot >> 3 & 1;
  if (slot >= 16) return;
  const int L = g2 * 16 + j;
  for (int g = gi0; g < G; g += 16) {
    f32x4_t v[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const f32x4_t* rp = red + (size_t)(g * NR + i) * 8 * 64;
      v[i] = rp[L] + rp[L + 40];
#pragma unroll
      for (int w2 = 1; w2 < 8; ++w2) v[i] += rp[w2 * 64 + L] + rp[w2 * 64 + L + 40];
    }
    if (j >= p.M) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int n = n_lo + g * 8 + g2 * 4 + q;
      if (n >= n_hi) continue;
      if (MODE == 2) {
        const float gt = bf2f(f2bf(v[0][q])), up = bf2f(f2bf(v[NR - 1][q]));
        ((bf16_t*)p.C)[(size_t)j * p.ldc + n] = f2bf(gt / (1.0f + __expf(-gt)) * up);
      } else if (MODE == 1) {
        float o = v[0][q];
        if (p.R) o += p.r_f32 ? ((const float*)p.R)[(size_t)j * p.ldr + n] : bf2f(((const bf16_t*)p.R)[(size_t)j * p.ldr + n]);
        ((float*)p.C)[(size_t)j * p.ldc + n] = o;
      } else {
        ((bf16_t*)p.C)[(size_t)j * p.ldc + n] = f2bf(v[0][q]);
      }
    }
  }
}

template <int MODE, int PER, int NB, bool NORM>
hipError_t dec8_go(const Dec8Args& a, int grid, size_t lds, hipStream_t st) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_dec8_kernel<MODE, PER, NB, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, 256 + DEC8_MAXG * 2 * 8 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL((gemm_nt_dec8_kernel<MODE, PER, NB, NORM>), dim3(grid), dim3(512), lds, st, a);
  return hipGetLastError();
}

}  // namespace

// per = k-steps of 64 per wave; the instantiated slices: K = 4096 (8), 5120 (10), 11008 (22 = 2 x 11), 13824 (27 = 14 + 13)
static int dec8_per(int K) { return ((K >> 6) + 7) >> 3; }
bool gemm_nt_dec8_ok(int M, int N, int K, int mode, int with_norm) {
  if (M < 1 || M > 8 || K % 64 || N % 8 || (mode == 2 && N % 128)) return false;
  const int per = dec8_per(K);
  if (per != 8 && per != 10 && per != 22 && per != 27) return false;
  if (with_norm && per != 8 && per != 10) return false;      // the fold keeps the slice's fp32 rows in registers: hidden sizes only
  const int rows = mode == 2 ? N / 2 : N;
  const int rpw = (rows + 255) / 256;
  return (rpw + 7) / 8 <= DEC8_MAXG;
}

hipError_t launch_gemm_nt_dec8(const void* x, int x_f32, int ldx, const bf16_t* norm_w, float eps, const bf16_t* W, int ldw, void* C, int ldc,
                               const void* R, int r_f32, int ldr, int mode, int M, int N, int K, hipStream_t st) {
  if (mode < 0 || mode > 2 || !gemm_nt_dec8_ok(M, N, K, mode, norm_w != nullptr) || !x || !W || !C || (mode != 1 && R) || ldx % 8 || ldw % 8) return hipErrorInvalidValue;
  Dec8Args a;
  a.x = x; a.x_f32 = x_f32; a.ldx = ldx; a.norm_w = norm_w; a.eps = eps; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.R = R; a.r_f32 = r_f32; a.ldr = ldr;
  a.M = M; a.N = N; a.K = K;
  const int rows = mode == 2 ? N / 2 : N;
  static int cus = 0;
  if (!cus) { int dev = 0; hipDeviceProp_t pr; cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
  a.rpw = (rows + cus - 1) / cus;
  if ((a.rpw + 7) / 8 > DEC8_MAXG) a.rpw = (rows + 255) / 256;
  const int grid = (rows + a.rpw - 1) / a.rpw, G = (a.rpw + 7) / 8;
  const size_t lds = 256 + (size_t)G * (mode == 2 ? 2 : 1) * 8 * 1024;
  const int per = dec8_per(K);
#define DEC8_MODES(PER_, NB_, NORM_)                                              \
  do {                                                                            \
    if (mode == 0) return dec8_go<0, PER_, NB_, NORM_>(a, grid, lds, st);         \
    if (mode == 1) return dec8_go<1, PER_, NB_, NORM_>(a, grid, lds, st);         \
    return dec8_go<2, PER_, NB_, NORM_>(a, grid, lds, st);                        \
  } while (0)
  if (norm_w) {
    if (per == 8) DEC8_MODES(8, 1, true);
    if (per == 10) DEC8_MODES(10, 1, true);
  } else {
    if (per == 8) DEC8_MODES(8, 1, false);
    if (per == 10) DEC8_MODES(10, 1, false);
    if (per == 22) DEC8_MODES(22, 2, false);
    if (per == 27) DEC8_MODES(27, 2, false);
  }
#undef DEC8_MODES
  return hipErrorInvalidValue;
}
