// bf16 MFMA GEMMs for gfx950.
//
//  gemm_nt : C[M,N] = epi( alpha * ( A1[M,K1] . B1[N,K1]^T  +  A2[M,K2 @ group] . B2[N,K2]^T ) )
//            Both operands K-contiguous ("NT").  The second (A2,B2) pair is the LoRA tail:
//            y = [x | t] . [W | B]^T with t = s*(x A^T) — LoRA fused into the base GEMM by
//            K-concatenation, one fp32 MFMA accumulator, no second pass over C.  For fused
//            projections (q|k|v, gate|up) the tail's A-operand columns depend on the output
//            column group: A2 column offset = (n0 / a2_group_n) * a2_group_stride.
//            128x128 tile, BK=64, 4 waves (2x2) of 64x64, v_mfma_f32_16x16x32_bf16,
//            global_load_lds_dwordx4 staging (LDS image lane-linear, XOR swizzle applied on
//            the per-lane SOURCE address and on the ds_read_b128 address), double-buffered,
//            one barrier per K-step, XCD-aware grouped tile order.
//  gemm_tn : C[N1,N2] (fp32) += alpha * sum_m P[m,N1] . Q[m,N2 @ group]   (LoRA wgrad:
//            dB = dY^T t, dA = dT^T x).  Contraction runs along the rows of both operands, so
//            fragments come from LDS through ds_read_b64_tr_b16; split over M with fp32
//            atomics (gradients accumulate across micro-batches anyway).
#include <utility>
#include "common.h"
#include "kernels.h"
#include <type_traits>
#include <vector>
#include <array>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;       // 16 KiB per operand tile
constexpr int STAGE_BYTES = 2 * TILE_BYTES;   // A + B
constexpr int GROUP_M = 8;

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == OPADPO_ACT_QUICK_GELU) return v / (1.0f + __expf(-1.702f * v));
  if (act == OPADPO_ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
  return v;
}

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// Epilogue modes.  The bias / activation choice is made ONCE per kernel (epi_dispatch) and the unrolled epilogue is
// instantiated per mode: with the checks inside epilogue4, every one of its 16-64 inlined copies carried both activation
// bodies behind branches, the taken branches hopped through ~200 KiB of cold code, and the instruction-cache misses made the
// epilogue of a 256x256 block take 20-27 us (a fifth of the block's lifetime; measured with s_memtime stamps, variant 27).
//   0: alpha only   1: + bias   2: (+ bias) quick-GELU   3: (+ bias) GELU   4: nothing (alpha == 1)
template <class F>
__device__ __forceinline__ void epi_dispatch(const GemmNTArgs& p, F&& f) {
  if (p.act == 0) {
    if (!p.bias) {
      if (p.alpha == 1.0f) f(std::integral_constant<int, 4>{});
      else f(std::integral_constant<int, 0>{});
    } else {
      f(std::integral_constant<int, 1>{});
    }
  } else if (p.act == OPADPO_ACT_QUICK_GELU) {
    f(std::integral_constant<int, 2>{});
  } else {
    f(std::integral_constant<int, 3>{});
  }
}

// An accumulator that lives in an AGPR tuple (tied there by the inline-asm MFMAs) is copied out with explicit
// v_accvgpr_read: left to the register allocator, the 64 tuples of the w4 kernels were spilled to scratch around the epilogue.
__device__ __forceinline__ void acc_read4(const f32x4_t& a, float (&v)[4]) {
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[0]) : "a"(a[0]));
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[1]) : "a"(a[1]));
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[2]) : "a"(a[2]));
  asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(v[3]) : "a"(a[3]));
}

// 16-byte store of a result row piece that nothing in this kernel reads again: non-temporal (streaming) when `nt` - see launch_gemm_nt
typedef __attribute__((ext_vector_type(4))) unsigned u32x4n_t;
__device__ __forceinline__ void store16(void* dst, const uint4& v, int nt) {
  const u32x4n_t w = {v.x, v.y, v.z, v.w};
  if (nt) __builtin_nontemporal_store(w, (u32x4n_t*)dst);
  else *(u32x4n_t*)dst = w;
}

// the w4 kernels take only bias-free, activation-free problems (the launcher routes the others to the 8-wave kernel): two
// epilogue instantiations instead of five keep the hot kernel's code small
template <class F>
__device__ __forceinline__ void epi_dispatch_plain(const GemmNTArgs& p, F&& f) {
  if (p.alpha == 1.0f) f(std::integral_constant<int, 4>{});
  else f(std::integral_constant<int, 0>{});
}

// alpha, bias and activation of 4 consecutive output columns n..n+3
template <int MD>
__device__ __forceinline__ void epi_pre4(const GemmNTArgs& p, int n, float (&v)[4]) {
  if (MD == 4) return;
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] *= p.alpha;
  if (MD == 1 || (MD >= 2 && p.bias)) {
    const uint2 b = *(const uint2*)(p.bias + n);
    v[0] += __uint_as_float(b.x << 16); v[1] += __uint_as_float(b.x & 0xffff0000u);
    v[2] += __uint_as_float(b.y << 16); v[3] += __uint_as_float(b.y & 0xffff0000u);
  }
  if (MD >= 2) {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = act_apply(v[q], MD == 2 ? OPADPO_ACT_QUICK_GELU : OPADPO_ACT_GELU);
  }
}

// epilogue for 4 consecutive output columns of one row
template <int MD>
__device__ __forceinline__ void epilogue4(const GemmNTArgs& p, int m, int n, float v0, float v1, float v2, float v3) {
  float v[4] = {v0, v1, v2, v3};
  epi_pre4<MD>(p, n, v);
  if (p.R) {
    if (p.r_f32) {
      const float4 r = *(const float4*)((const float*)p.R + (size_t)m * p.ldr + n);
      v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
    } else {
      const uint2 r = *(const uint2*)((const bf16_t*)p.R + (size_t)m * p.ldr + n);
      v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
      v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
    }
  }
  if (p.out_f32) {
    *(float4*)((float*)p.C + (size_t)m * p.ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    uint2 o;
    o.x = pack_bf2(v[0], v[1]);
    o.y = pack_bf2(v[2], v[3]);
    *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
  }
}

// ------------------------------------------------------------------------------------------
// gemm_nt 128x128 kernel (small / skinny-N problems): 4 waves x 64x64, two LDS stages filled by global_load_lds, s_setprio around
// the MFMA cluster.  Instantiated with K-step 64, two blocks per CU (K-step 32 with 3-4 blocks, the register-staged and 32x32x16
// forms and a three-stage 128x256 ring were measured slower and are gone; DESIGN.md §6 keeps their numbers).
// ------------------------------------------------------------------------------------------
template <int BKX, bool PRIO, int MINW>
__global__ __launch_bounds__(256, MINW) void gemm_nt_kernel_x(GemmNTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWB = BKX * 2, CPR = BKX / 8, RPP = 1024 / ROWB, PIECES = 128 / RPP, PPW = PIECES / 4;
  constexpr int TB = 128 * ROWB, SB = 2 * TB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int m0, n0;
  if (p.quarter) {      // tail of a 256x256 launch: quarter (blockIdx & 3) of tile tile0 + (blockIdx >> 2) in THAT kernel's grouped order
    const int tiles_m = (p.M + 255) / 256, tiles_n = p.N / 256;
    const int swz = p.tile0 + (blockIdx.x >> 2), q = blockIdx.x & 3;
    const int width = p.group_m * tiles_n;
    const int first_m = (swz / width) * p.group_m;
    const int gsz = min(tiles_m - first_m, p.group_m);
    m0 = (first_m + (swz % width) % gsz) * 256 + (q >> 1) * 128;
    n0 = ((swz % width) / gsz) * 256 + (q & 1) * 128;
    if (m0 >= p.M) return;                       // block-uniform: the lower half of a ragged last row tile may be empty
  } else {
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
    const int swz = xcd_remap(blockIdx.x, gridDim.x);
    const int width = GROUP_M * tiles_n;
    const int group_id = swz / width;
    const int first_m = group_id * GROUP_M;
    const int gsz = min(tiles_m - first_m, GROUP_M);
    m0 = (first_m + (swz % width) % gsz) * BM;
    n0 = ((swz % width) / gsz) * BN;
  }
  const int nt1 = p.K1 / BKX, nt2 = p.K2 / BKX, nt = nt1 + nt2;
  const bf16_t* a2 = p.A2;
  if (p.a2_group_n > 0) a2 += (size_t)(n0 / p.a2_group_n) * p.a2_group_stride;
  if (p.a1_group_n > 0) p.A1 += (size_t)(n0 / p.a1_group_n) * p.a1_group_stride;
  const int srow = lane / CPR, spos = lane % CPR;
  static_assert(BKX == 64, "the 128-byte-row swizzle below is for K-step 64");
  auto fsw = [](int r) { return (r >> 1) & 7; };

  auto issue = [&](int buf, int t) {
    const bf16_t *Ab, *Bb;
    int lda, ldb, k0;
    if (t < nt1) { Ab = p.A1; lda = p.lda1; Bb = p.B1; ldb = p.ldb1; k0 = t * BKX; }
    else { Ab = a2; lda = p.lda2; Bb = p.B2; ldb = p.ldb2; k0 = (t - nt1) * BKX; }
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int piece = wave * PPW + i;
      const int r = piece * RPP + srow;
      const int c = spos ^ fsw(r);
      const int gr = min(m0 + r, p.M - 1);
      char* dA = smem + buf * SB + piece * 1024;
      __builtin_amdgcn_global_load_lds(GLB_PTR(Ab + (size_t)gr * lda + k0 + c * 8), LDS_PTR(void, dA), 16, 0, 0);
      // B tile: LDS row r holds the tile's row (r & 64) + 4 * (r & 15) + ((r >> 4) & 3), i.e. column slot s of B fragment j is row 4s + j
      // of the wave's 64: a lane ends up with 4 CONSECUTIVE output columns per (row block, r) and 16 lanes store 128 contiguous bytes of
      // a row (bf16) instead of 32-byte pieces of 16 rows; the permutation lives in the source address only
      const int rb = (r & 64) + 4 * (r & 15) + ((r >> 4) & 3);
      __builtin_amdgcn_global_load_lds(GLB_PTR(Bb + (size_t)(n0 + rb) * ldb + k0 + c * 8), LDS_PTR(void, dA + TB), 16, 0, 0);
    }
  };
  const int wm = wave >> 1, wn = wave & 1;
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int frow = lane & 15, fchk = lane >> 4;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    if (t + 1 < nt) issue(cur ^ 1, t + 1);
    const char* As = smem + cur * SB;
    const char* Bs = As + TB;
#pragma unroll
    for (int kk = 0; kk < BKX / 32; ++kk) {
      bf16x8_t af[4], bfr[4];
      const int c = kk * 4 + fchk;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = wm * 64 + i * 16 + frow;
        af[i] = *(const bf16x8_t*)(As + row * ROWB + ((c ^ fsw(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = wn * 64 + j * 16 + frow;
        bfr[j] = *(const bf16x8_t*)(Bs + row * ROWB + ((c ^ fsw(row)) << 4));
      }
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    cur ^= 1;
  }
  // acc[i][j][r] of lane (frow, fchk): row i*16 + 4*fchk + r, column 4*frow + j of the wave's 64x64 block
  epi_dispatch(p, [&](auto MD_) {
    constexpr int md = decltype(MD_)::value;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + i * 16 + 4 * fchk + r;
        if (m < p.M) epilogue4<md>(p, m, n0 + wn * 64 + 4 * frow, acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]);
      }
    }
  });
}

// ------------------------------------------------------------------------------------------
// gemm_nt tail kernel (round 4): SIXTEENTH tiles.  A partly filled last round of the 256x256 kernel (<= 64 tiles) used to run as quarter tiles
// on the 128x128 kernel - 4 blocks per tile, i.e. 64-256 blocks for a chip of 256 CUs: a 16-tile tail (6 % of a round's work) cost half a round.
// Here block b computes the 64x64 piece (b & 15) of tile tile0 + (b >> 4) of the 4-wave kernel's grouped order over the FULL K range, in the
// same k order (K-tiles in order, two 32-deep MFMA steps per K-tile): every element of C is bit-identical to what the 256x256 kernel and the
// quarter tiles write (tests/test_ops_gpu.py::test_gemm_nt_split_k_tail), so which tiles are tail tiles still cannot change a bit of the output.
// 16 blocks per tile fill the chip from a 16-tile tail on; 4 waves x (16 rows x 64 columns), a ring of NS 16-KiB LDS stages filled by
// global_load_lds with NS - 1 K-tiles in flight (counted vmcnt + raw barrier: a 16-tile tail is one block per CU, nothing else hides the latency), the
// 128x128 kernel's LDS image / B-row permutation (a lane owns 4 consecutive output columns).  Arithmetic intensity is low by construction
// (64 FLOP per operand byte from L2): the point is latency, not rate.
// ------------------------------------------------------------------------------------------
template <int NS>                                   // ring depth: NS - 1 K-tiles in flight per block.  8 (128 KiB: one block per CU) for tails that give every CU at most
                                                    // one block - nothing else hides the L2 / HBM latency there -, 4 (64 KiB, two blocks per CU) for longer tails
__global__ __launch_bounds__(256) void gemm_nt_tail64_kernel(GemmNTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ROWB = 128, TB = 64 * ROWB, SB = 2 * TB;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int m0, n0;
  {
    const int tiles_m = (p.M + 255) / 256, tiles_n = p.N / 256;
    const int swz = p.tile0 + (blockIdx.x >> 4), q = blockIdx.x & 15;
    const int width = p.group_m * tiles_n;
    const int first_m = (swz / width) * p.group_m;
    const int gsz = min(tiles_m - first_m, p.group_m);
    m0 = (first_m + (swz % width) % gsz) * 256 + (q >> 2) * 64;
    n0 = ((swz % width) / gsz) * 256 + (q & 3) * 64;
    if (m0 >= p.M) return;                       // block-uniform: the lower part of a ragged last row tile may be empty
  }
  const int nt1 = p.K1 / 64, nt2 = p.K2 / 64, nt = nt1 + nt2;
  const bf16_t* a2 = p.A2;
  if (p.a2_group_n > 0) a2 += (size_t)(n0 / p.a2_group_n) * p.a2_group_stride;
  if (p.a1_group_n > 0) p.A1 += (size_t)(n0 / p.a1_group_n) * p.a1_group_stride;
  const int srow = lane >> 3, spos = lane & 7;
  auto fsw = [](int r) { return (r >> 1) & 7; };
  auto issue = [&](int buf, int t) {
    const bf16_t *Ab, *Bb;
    int lda, ldb, k0;
    if (t < nt1) { Ab = p.A1; lda = p.lda1; Bb = p.B1; ldb = p.ldb1; k0 = t * 64; }
    else { Ab = a2; lda = p.lda2; Bb = p.B2; ldb = p.ldb2; k0 = (t - nt1) * 64; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int piece = wave * 2 + i;              // 8 pieces of 8 rows x 128 bytes per operand
      const int r = piece * 8 + srow;
      const int c = spos ^ fsw(r);
      const int gr = min(m0 + r, p.M - 1);
      char* dA = smem + buf * SB + piece * 1024;
      __builtin_amdgcn_global_load_lds(GLB_PTR(Ab + (size_t)gr * lda + k0 + c * 8), LDS_PTR(void, dA), 16, 0, 0);
      const int rb = 4 * (r & 15) + ((r >> 4) & 3);   // column slot s of B fragment j is row 4s + j of the 64 (gemm_nt_kernel_x)
      __builtin_amdgcn_global_load_lds(GLB_PTR(Bb + (size_t)(n0 + rb) * ldb + k0 + c * 8), LDS_PTR(void, dA + TB), 16, 0, 0);
    }
  };
  f32x4_t acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NS - 1; ++t)
    if (t < nt) issue(t, t);
  const int frow = lane & 15, fchk = lane >> 4;
  for (int t = 0; t < nt; ++t) {
    // K-tile t has landed for this wave's pieces when at most the pieces of the K-tiles behind it are outstanding (4 per K-tile and wave)
    const int behind = min(nt - 1 - t, NS - 2);
    if (NS > 4 && behind >= 6) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (NS > 4 && behind == 5) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (NS > 4 && behind == 4) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (NS > 4 && behind == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (behind >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (behind == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                   // ... for every wave's pieces; and everyone has finished reading the stage of K-tile t - 1
    if (t + NS - 1 < nt) issue((t + NS - 1) % NS, t + NS - 1);
    const char* As = smem + (t % NS) * SB;
    const char* Bs = As + TB;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + fchk;
      const int arow = wave * 16 + frow;
      const bf16x8_t af = *(const bf16x8_t*)(As + arow * ROWB + ((c ^ fsw(arow)) << 4));
      bf16x8_t bfr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = j * 16 + frow;
        bfr[j] = *(const bf16x8_t*)(Bs + row * ROWB + ((c ^ fsw(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc[j], 0, 0, 0);
    }
  }
  // acc[j][r] of lane (frow, fchk): row wave*16 + 4*fchk + r, column 4*frow + j of the block's 64x64 piece
  epi_dispatch(p, [&](auto MD_) {
    constexpr int md = decltype(MD_)::value;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wave * 16 + 4 * fchk + r;
      if (m < p.M) epilogue4<md>(p, m, n0 + 4 * frow, acc[0][r], acc[1][r], acc[2][r], acc[3][r]);
    }
  });
}

// ------------------------------------------------------------------------------------------
// gemm_nt "ping-pong" kernel: 256x256 block tile, 8 waves (2 x 4, each 128x64), BK = 64, two 64-KiB
// LDS stages (128 KiB, one block per CU).  The two wave groups G0 = waves 0-3 and G1 = waves 4-7
// (wave w and w+4 share a SIMD) run the SAME per-K-tile program
//        LOAD(t): 24 ds_read_b128 (all A/B fragments of K-tile t)    |  MFMA(t): 64 MFMAs from registers
// offset by one phase, separated by raw s_barriers: while one group streams fragments out of LDS the
// other keeps the SIMD's matrix pipe busy (s_setprio 1).  LDS-DMA for tile t+1 is issued two phases
// before its first reader and waited for with s_waitcnt vmcnt(0) one phase later, so it is in flight
// across a barrier; nothing waits for a load it just issued.
//   phase 2t   : G0 issue(t+1), LOAD(t)   | G1 issue(t+1), MFMA(t-1)
//   phase 2t+1 : G0 MFMA(t), vmcnt(0)     | G1 LOAD(t), vmcnt(0)
// ------------------------------------------------------------------------------------------
constexpr int P_BM = 256, P_BN = 256, P_BK = 64;
constexpr int P_TILE = P_BM * P_BK * 2;          // 32 KiB per operand tile
constexpr int P_STAGE = 2 * P_TILE;              // A + B

// ------------------------------------------------------------------------------------------
// gemm_nt "w4" kernel: 256x256 tile, BK = 64, FOUR waves (2x2), each owning a 128x128 block of C in 64 accumulator
// fragments (256 accumulator registers -> one wave per SIMD, 512-register budget), two 64-KiB LDS stages filled by
// buffer_load ... lds.  One wave per SIMD means nothing else can feed the matrix pipe, so every non-MFMA instruction of
// the K-loop sits INSIDE the MFMA stream and a wave reads each LDS fragment for 8 MFMAs (128 KiB of LDS reads per K-tile
// and block instead of 192 KiB in the 8-wave kernel).
// Round 5: the K-loop is ONE asm block with explicit registers, generated by csrc/w4_kloop_gen.py into csrc/w4_kloop.inc (the
// generator holds the register plan, the slot table and its hazard checks).  Rounds 1-4 wrote the same schedule idea in HIP (asm
// MFMAs / DMA halves pinned with sched_barrier): 562 quad-cycles per K-tile and wave against 522 for the bare MFMAs; the bisect against the
// vendor library's kernel of this geometry (tools/micro/kloop_bisect_gen.py, profiles/r05_kloop_bisect.txt) named the elements:
//   * the CU's vector-memory path takes one 1-KiB LDS-DMA piece per wave every 64 cycles: 13 pieces at a period of 4 MFMAs are free,
//     at a period of 3 they cost +17 quad-cycles per K-tile, at 2 +66 (bursts of 5 at period 3 are absorbed by the queue);
//   * the landing wait + barrier of tile t+1 belong at MFMA 92 (not 100) so that the 16 fragment reads of the next tile spread over
//     30 MFMA gaps with never two memory instructions in one gap;
//   * what the compiler adds around asm statements (lgkmcnt re-waits it cannot prove redundant, v_add3 address arithmetic per
//     fragment group, a taken branch in mid-tile, accumulators allocated out of order) is worth another 14 quad-cycles.
// Schedule per K-tile t (128 MFMAs): 16 reads of k-half 1 in gaps 1..31 | lgkmcnt(0) + barrier at 39/40: the stage of tile t is dead |
// 13 pieces of tile t+2 at gaps 41, 45, .. 89 (M0 written in the gap after the previous piece) | vmcnt(13) + barrier at 92/93: tile t+1
// has landed | 16 reads of k-half 0 of tile t+1 + the last 3 pieces over gaps 94..121 | lgkmcnt(0) at 127.  531 quad-cycles in the harness.
// ------------------------------------------------------------------------------------------
// ORDER_B: the 16 DMA pieces of a K-tile go B half first instead of A half first.  Nothing else changes (both halves are waited for
// together), yet same-box sustained runs (tools/ab_gemm.sh, 300 launches per shape) show a stable preference by shape:
// N <= 4096 (o / down projection, every dgrad into the hidden width, the r-wide LoRA products) is 1.5-4 % faster B first, the wide
// projections (q|k|v, gate|up, lm_head) 2.5-3 % faster A first.
// wave-uniform copy of a pointer (both halves through readfirstlane): a base the compiler cannot prove uniform turns buffer accesses into waterfall loops
__device__ __forceinline__ void* w4_uniform_ptr(const void* q) {
  const unsigned long long v = (unsigned long long)q;
  return (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}
#ifndef OPADPO_W4_RPD
#define OPADPO_W4_RPD 2       // residual row blocks in flight ahead of the read-out: one-tile-per-workgroup kernel
#endif
#ifndef OPADPO_W4S_RPD
#define OPADPO_W4S_RPD 1      // streaming kernel (its fragment registers stay live across the epilogue)
#endif
// residual operands the direct epilogue takes (everything else with a residual goes through the staged epilogue of gemm_nt_w4_kernel)
__host__ __device__ __forceinline__ bool w4_direct_resid_ok(const GemmNTArgs& p, bool ba = false) {
  return p.R && p.r_f32 && p.out_f32 && (ba || (!p.bias && p.act == 0)) && ((unsigned long long)p.M + 256ull) * (unsigned)p.ldr * 4ull < 0xffffffffull &&
         ((unsigned long long)p.M + 256ull) * (unsigned)p.ldc * 4ull < 0xffffffffull && p.ldr % 4 == 0;
}
// bias / activation problems the BA instantiations of the 256x256 kernels take (direct epilogue: alpha, bias of the lane's 8 columns, quick-GELU / GELU,
// then an fp32 residual - the order of epilogue4, so every kernel of the library gives the same bits)
__host__ __device__ __forceinline__ bool w4_direct_ba_ok(const GemmNTArgs& p) {
  return (p.bias || p.act == OPADPO_ACT_QUICK_GELU || p.act == OPADPO_ACT_GELU) && (p.act == 0 || p.act == OPADPO_ACT_QUICK_GELU || p.act == OPADPO_ACT_GELU) &&
         (!p.R || w4_direct_resid_ok(p, true)) && !p.rope_cos && !p.rope_pos;
}
// Direct epilogue of the 4-wave 256x256 kernels: accumulator layout acc[i][j][r] of lane (frow, fchk) = row i*16 + 4*fchk + r, column 8*frow + j of the
// wave's 128x128 block (the 8 fragments j of one (i, r) are 8 CONSECUTIVE columns).  No LDS, no barrier.
template <int PD = 1, bool BA = false, bool NORES = false>      // PD = row blocks of the residual operand requested ahead of the accumulator read-out (32 registers each); BA = bias / activation modes; NORES = no residual support compiled in
__device__ __forceinline__ void w4_direct_epilogue(const GemmNTArgs& p, f32x4_t (&acc)[8][8], int m0, int ncol0, int wr, int frow, int fchk) {
  // DIRECT epilogue (plain and alpha-scaled products, bf16 or fp32): 16-byte stores straight from the accumulators, one instruction =
  // 4 rows x 256 contiguous bytes (bf16) - no LDS round trip, no barrier.  (Round 1 staged every block through LDS because a fragment
  // then held 4 columns of 16 rows: 32-byte pieces, 27 us per block.)
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4s_t;
  const unsigned esz = p.out_f32 ? 4u : 2u;
  const bool small = ((unsigned long long)p.M + 256ull) * (unsigned)p.ldc * esz < 0xffffffffull;
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(w4_uniform_ptr(p.C), 0, small ? (int)((unsigned)p.M * (unsigned)p.ldc * esz) : 0, 0x00020000);
  const int mrow0 = m0 + wr * 128 + 4 * fchk, col = ncol0 + 8 * frow;
  // fp32 residual operand (w4_direct_resid_ok: fp32 R, fp32 C, 32-bit offsets): C = alpha * acc + R with the R rows of row block i + 1 requested
  // (non-temporal, 8 x 16 B per lane) before row block i is read out of the accumulators - the residual stream of the decoder layers is added here
  // instead of in the RMSNorm pass that follows (same fp32 operation on the same operands: same bits)
  const bool has_r = !NORES && p.R != nullptr;
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(w4_uniform_ptr(has_r ? p.R : p.C), 0, has_r ? (int)((unsigned)p.M * (unsigned)p.ldr * 4u) : 0, 0x00020000);
  u32x4s_t rq[PD + 1][4][2];
  auto r_issue = [&](int i, u32x4s_t (&dst)[4][2]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned vo = ((unsigned)(mrow0 + i * 16 + r) * (unsigned)p.ldr + (unsigned)col) * 4u;
      dst[r][0] = __builtin_amdgcn_raw_buffer_load_b128(rR, vo, 0, 2);
      dst[r][1] = __builtin_amdgcn_raw_buffer_load_b128(rR, vo + 16u, 0, 2);
    }
  };
  if (has_r) {
#pragma unroll
    for (int i = 0; i < PD; ++i) r_issue(i, rq[i]);
  }
  float bj[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // BA: the bias of the lane's 8 consecutive columns
  if constexpr (BA) { if (p.bias) unpack8(*(const uint4*)(p.bias + col), bj); }
  auto body = [&](auto MD_) {
    constexpr int md = decltype(MD_)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float v[8][4];
      if (has_r && i + PD < 8) r_issue(i + PD, rq[(i + PD) % (PD + 1)]);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        acc_read4(acc[i][j], v[j]);
        if constexpr (BA) {          // epi_pre4's arithmetic with the COLUMN's bias (v[j][0..3] are four rows of column j)
          if (md != 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float t = v[j][r] * p.alpha;
              if (md == 1 || (md >= 2 && p.bias)) t += bj[j];
              if (md >= 2) t = act_apply(t, md == 2 ? OPADPO_ACT_QUICK_GELU : OPADPO_ACT_GELU);
              v[j][r] = t;
            }
          }
        } else {
          epi_pre4<md>(p, 0, v[j]);
        }
      }
      if (has_r) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j][r] += __uint_as_float(rq[i % (PD + 1)][r][j >> 2][j & 3]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = mrow0 + i * 16 + r;
        if (p.out_f32) {
          u32x4s_t lo, hi;
          lo[0] = __float_as_uint(v[0][r]); lo[1] = __float_as_uint(v[1][r]); lo[2] = __float_as_uint(v[2][r]); lo[3] = __float_as_uint(v[3][r]);
          hi[0] = __float_as_uint(v[4][r]); hi[1] = __float_as_uint(v[5][r]); hi[2] = __float_as_uint(v[6][r]); hi[3] = __float_as_uint(v[7][r]);
          if (small) {
            const unsigned vo = ((unsigned)m * (unsigned)p.ldc + (unsigned)col) * 4u;
            if (p.store_nt) { __builtin_amdgcn_raw_buffer_store_b128(lo, rC, vo, 0, 2); __builtin_amdgcn_raw_buffer_store_b128(hi, rC, vo + 16u, 0, 2); }
            else { __builtin_amdgcn_raw_buffer_store_b128(lo, rC, vo, 0, 0); __builtin_amdgcn_raw_buffer_store_b128(hi, rC, vo + 16u, 0, 0); }
          } else if (m < p.M) {
            *(u32x4s_t*)((float*)p.C + (size_t)m * p.ldc + col) = lo;
            *(u32x4s_t*)((float*)p.C + (size_t)m * p.ldc + col + 4) = hi;
          }
        } else {
          u32x4s_t o;
          o[0] = pack_bf2(v[0][r], v[1][r]); o[1] = pack_bf2(v[2][r], v[3][r]); o[2] = pack_bf2(v[4][r], v[5][r]); o[3] = pack_bf2(v[6][r], v[7][r]);
          if (small) {
            if (p.store_nt) __builtin_amdgcn_raw_buffer_store_b128(o, rC, ((unsigned)m * (unsigned)p.ldc + (unsigned)col) * 2u, 0, 2);
            else __builtin_amdgcn_raw_buffer_store_b128(o, rC, ((unsigned)m * (unsigned)p.ldc + (unsigned)col) * 2u, 0, 0);
          }
          else if (m < p.M) *(u32x4s_t*)((bf16_t*)p.C + (size_t)m * p.ldc + col) = o;
        }
      }
      __builtin_amdgcn_sched_barrier(0);      // one row block at a time: 8, not 64, accumulator tuples live in VGPRs
    }
  };
  if constexpr (BA) {      // BA problems carry a bias or an activation: modes 1..3 only (three copies of the body, not five: the kernel must not spill -
    // its K-loop is an asm block that does not know about a scratch descriptor)
    if (p.act == OPADPO_ACT_QUICK_GELU) body(std::integral_constant<int, 2>{});
    else if (p.act == OPADPO_ACT_GELU) body(std::integral_constant<int, 3>{});
    else body(std::integral_constant<int, 1>{});
  } else {
    epi_dispatch_plain(p, body);
  }
}

// SwiGLU PAIR (merged no-grad pass; weight rows per 128 = [64 gate | 64 up]) in the DIRECT epilogue (round 5): in the accumulator layout the lanes frow 0..7 of a
// 16-lane row hold gate columns 8 frow + j, the lanes frow 8..15 the up columns of the SAME output columns - lane ^ 8 is the partner (DPP row_ror:8, no LDS).
// The gate lane finishes rows r = 0, 1 of a row block, the up lane rows 2, 3: each sends the partner the two values it does not finish itself (16 DPP moves
// per row block), forms silu(gate) * up on the bf16-ROUNDED values with silu_mul_fwd_kernel's formula (bit-identical to projection + silu_mul_fwd and to the
// staged epilogue) and stores 8 consecutive output columns of its two rows.  No LDS, no barrier: the product streams (gemm_nt_w4s_kernel).
__host__ __device__ __forceinline__ bool w4_direct_swiglu_pair_ok(const GemmNTArgs& p) {
  return p.act == OPADPO_ACT_SWIGLU_PAIR && !p.R && !p.bias && !p.out_f32 && p.alpha == 1.0f && p.ldc % 8 == 0 &&
         ((unsigned long long)p.M + 256ull) * (unsigned)p.ldc * 2ull < 0xffffffffull;
}
__device__ __forceinline__ void w4_direct_epilogue_swiglu_pair(const GemmNTArgs& p, f32x4_t (&acc)[8][8], int m0, int ncol0, int wr, int frow, int fchk) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4s_t;
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(w4_uniform_ptr(p.C), 0, (int)((unsigned)p.M * (unsigned)p.ldc * 2u), 0x00020000);
  const bool is_up = frow >= 8;
  const int mrow0 = m0 + wr * 128 + 4 * fchk + (is_up ? 2 : 0);           // this lane's two rows of every row block
  const int ocol = ncol0 / 2 + 8 * (frow & 7);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v[8][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc_read4(acc[i][j], v[j]);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      float gt[8], up[8], o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float mine = is_up ? v[j][2 + k] : v[j][k];                   // the value of MY row that I hold (gate lane: gate, up lane: up)
        const float send = is_up ? v[j][k] : v[j][2 + k];                   // the partner's row
        const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xf, 0xf, false));      // row_ror:8 = lane ^ 8 inside the 16-lane row
        gt[j] = bf2f(f2bf(is_up ? recv : mine));
        up[j] = bf2f(f2bf(is_up ? mine : recv));
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = gt[e] / (1.0f + __expf(-gt[e])) * up[e];
      const uint4 oa = pack8(o);
      const u32x4s_t st4 = {oa.x, oa.y, oa.z, oa.w};
      const unsigned vo = ((unsigned)(mrow0 + i * 16 + k) * (unsigned)p.ldc + (unsigned)ocol) * 2u;
      if (p.store_nt) __builtin_amdgcn_raw_buffer_store_b128(st4, rC, vo, 0, 2);
      else __builtin_amdgcn_raw_buffer_store_b128(st4, rC, vo, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Table-free rotary embedding (rope_pos) in the DIRECT epilogue (round 5, streaming kernel's EPI = 3 instantiation): a wave's 128 columns are one head, the
// rotation partners d and d + 64 sit in lanes l and l ^ 8 of a 16-lane row (as the SwiGLU pair's gate / up).  The low lane finishes rows r = 0, 1 of a row block
// for BOTH partner columns, the high lane rows 2, 3: 16 DPP moves per row block, one sin / cos pair per (row, frequency) - half the transcendental work of the
// staged form, where both partner lanes evaluate the same angle - and two 16-byte stores per row.  Arithmetic of the staged epilogue / rope_kernel on the
// bf16-ROUNDED projection: out[d] = x[d] cos - x[d+64] sin, out[d+64] = x[d+64] cos + x[d] sin, angle = fract(pos * theta^(-2i/128) / 2 pi) revolutions.
__device__ __forceinline__ void w4_direct_epilogue_rope_pos(const GemmNTArgs& p, f32x4_t (&acc)[8][8], int m0, int ncol0, int wr, int frow, int fchk) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4s_t;
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(w4_uniform_ptr(p.C), 0, (int)((unsigned)p.M * (unsigned)p.ldc * 2u), 0x00020000);
  const bool is_hi = frow >= 8;
  const int gh = frow & 7;
  const int mrow0 = m0 + wr * 128 + 4 * fchk + (is_hi ? 2 : 0);
  const int col_lo = ncol0 + 8 * gh;
  float frev[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) frev[e] = __builtin_amdgcn_exp2f(-(float)(2 * (gh * 8 + e)) * (1.0f / 128.0f) * p.rope_l2theta) * 0.15915494309189535f;
  int posn[2] = {p.rope_pos[min(mrow0, p.M - 1)], p.rope_pos[min(mrow0 + 1, p.M - 1)]};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v[8][4];
    const int pos2[2] = {posn[0], posn[1]};
    if (i + 1 < 8) { posn[0] = p.rope_pos[min(mrow0 + (i + 1) * 16, p.M - 1)]; posn[1] = p.rope_pos[min(mrow0 + (i + 1) * 16 + 1, p.M - 1)]; }      // next row block's positions
#pragma unroll
    for (int j = 0; j < 8; ++j) acc_read4(acc[i][j], v[j]);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const float posf = (float)pos2[k];
      float olo[8], ohi[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float mine = is_hi ? v[j][2 + k] : v[j][k];
        const float send = is_hi ? v[j][k] : v[j][2 + k];
        const float recv = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(send), 0x128, 0xf, 0xf, false));      // row_ror:8 = lane ^ 8
        const float xlo = bf2f(f2bf(is_hi ? recv : mine)), xhi = bf2f(f2bf(is_hi ? mine : recv));
        const float x = __builtin_amdgcn_fractf(posf * frev[j]);
        const float cs = __builtin_amdgcn_cosf(x), sn = __builtin_amdgcn_sinf(x);
        olo[j] = xlo * cs + -1.0f * (xhi * sn);
        ohi[j] = xhi * cs + 1.0f * (xlo * sn);
      }
      const uint4 a = pack8(olo), b = pack8(ohi);
      const u32x4s_t sa = {a.x, a.y, a.z, a.w}, sb = {b.x, b.y, b.z, b.w};
      const unsigned vo = ((unsigned)(mrow0 + i * 16 + k) * (unsigned)p.ldc + (unsigned)col_lo) * 2u;
      if (p.store_nt) { __builtin_amdgcn_raw_buffer_store_b128(sa, rC, vo, 0, 2); __builtin_amdgcn_raw_buffer_store_b128(sb, rC, vo + 128u, 0, 2); }
      else { __builtin_amdgcn_raw_buffer_store_b128(sa, rC, vo, 0, 0); __builtin_amdgcn_raw_buffer_store_b128(sb, rC, vo + 128u, 0, 0); }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
__host__ __device__ __forceinline__ bool w4_direct_rope_pos_ok(const GemmNTArgs& p) {
  return p.rope_pos && !p.rope_cos && !p.R && !p.bias && p.act == 0 && !p.out_f32 && p.alpha == 1.0f && p.rope_cols % 128 == 0 && p.ldc % 8 == 0 &&
         ((unsigned long long)p.M + 256ull) * (unsigned)p.ldc * 2ull < 0xffffffffull;
}

// SwiGLU backward in the DIRECT epilogue (round 5): the block is d_act[:, ncol0 ..] of the down projection's dgrad, R the stored pre-activations
// [gate | up] ([M, 2N] bf16), C receives [d_gate | d_up] - the arithmetic of silu_mul_bwd_kernel on the bf16-ROUNDED d_act, so the result is
// bit-identical to the two-kernel path without d_act ever reaching HBM.  A lane owns 8 consecutive columns of 4 rows per row block: its gate
// and up values are two 16-byte loads per row, requested one row block ahead of the accumulator read-out (the staged form of rounds 3-4
// loaded them inside the read-back loop, every load's latency exposed: slower than the separate kernel).
__host__ __device__ __forceinline__ bool w4_direct_swiglu_bwd_ok(const GemmNTArgs& p) {
  return p.act == OPADPO_ACT_SWIGLU_BWD && p.R && !p.r_f32 && !p.out_f32 && !p.bias && p.alpha == 1.0f && p.ldr % 8 == 0 && p.ldc % 8 == 0 &&
         ((unsigned long long)p.M + 256ull) * (unsigned)p.ldr * 2ull < 0xffffffffull && ((unsigned long long)p.M + 256ull) * (unsigned)p.ldc * 2ull < 0xffffffffull;
}
__device__ __forceinline__ void w4_direct_epilogue_swiglu_bwd(const GemmNTArgs& p, f32x4_t (&acc)[8][8], int m0, int ncol0, int wr, int frow, int fchk) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4s_t;
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(w4_uniform_ptr(p.C), 0, (int)((unsigned)p.M * (unsigned)p.ldc * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(w4_uniform_ptr(p.R), 0, (int)((unsigned)p.M * (unsigned)p.ldr * 2u), 0x00020000);
  const int mrow0 = m0 + wr * 128 + 4 * fchk, col = ncol0 + 8 * frow;
  u32x4s_t gq[2][4][2];      // [buffer][row][gate / up]
  auto g_issue = [&](int i, u32x4s_t (&dst)[4][2]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned vo = ((unsigned)(mrow0 + i * 16 + r) * (unsigned)p.ldr + (unsigned)col) * 2u;
      dst[r][0] = __builtin_amdgcn_raw_buffer_load_b128(rR, vo, 0, 2);
      dst[r][1] = __builtin_amdgcn_raw_buffer_load_b128(rR, vo + (unsigned)p.N * 2u, 0, 2);
    }
  };
  g_issue(0, gq[0]);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float v[8][4];
    if (i + 1 < 8) g_issue(i + 1, gq[(i + 1) & 1]);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc_read4(acc[i][j], v[j]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float dd[8], gt[8], up[8], dg[8], du[8];
      uint4 db;
      db.x = pack_bf2(v[0][r], v[1][r]); db.y = pack_bf2(v[2][r], v[3][r]); db.z = pack_bf2(v[4][r], v[5][r]); db.w = pack_bf2(v[6][r], v[7][r]);
      unpack8(db, dd);                                          // d_act as the two-kernel path stores it: rounded to bf16
      const u32x4s_t gw = gq[i & 1][r][0], uw = gq[i & 1][r][1];
      unpack8(make_uint4(gw[0], gw[1], gw[2], gw[3]), gt);
      unpack8(make_uint4(uw[0], uw[1], uw[2], uw[3]), up);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float sg = 1.0f / (1.0f + __expf(-gt[e]));
        const float silu = gt[e] * sg;
        du[e] = dd[e] * silu;
        dg[e] = dd[e] * up[e] * sg * (1.0f + gt[e] * (1.0f - sg));
      }
      const uint4 og = pack8(dg), ou = pack8(du);
      const u32x4s_t sg4 = {og.x, og.y, og.z, og.w}, su4 = {ou.x, ou.y, ou.z, ou.w};
      const unsigned vo = ((unsigned)(mrow0 + i * 16 + r) * (unsigned)p.ldc + (unsigned)col) * 2u;
      if (p.store_nt) { __builtin_amdgcn_raw_buffer_store_b128(sg4, rC, vo, 0, 2); __builtin_amdgcn_raw_buffer_store_b128(su4, rC, vo + (unsigned)p.N * 2u, 0, 2); }
      else { __builtin_amdgcn_raw_buffer_store_b128(sg4, rC, vo, 0, 0); __builtin_amdgcn_raw_buffer_store_b128(su4, rC, vo + (unsigned)p.N * 2u, 0, 0); }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Logical tile index -> (row tile, column tile) of the 256x256 4-wave kernels: row tiles in groups of group_m, a group walked column by column - 32 consecutive
// indices are group_m rows x 32 / group_m columns, one XCD's footprint of a round.  (Bundling groups into super-groups so that a round of the chip covers 16 or 32
// rows x 16 or 8 columns instead of 8 x 32 measured 0.5 - 2 % slower: profiles/r06z_ab_superh.txt.)
__device__ __forceinline__ void w4_tile_map(const GemmNTArgs& p, int swz, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int width = p.group_m * tiles_n;
  const int first_m = (swz / width) * p.group_m;
  const int gsz = min(tiles_m - first_m, p.group_m);
  tm = first_m + (swz % width) % gsz;
  tn = (swz % width) / gsz;
}

#ifndef W4K_INC
#define W4K_INC "w4_kloop.inc"      // experiment builds (tools/build_kloop_exp.sh) name another file emitted by the same generator
#endif
#include W4K_INC
// DEEP (round 6): the K-loop text with the vendor library's skeleton (W4K_TEXT_*_DEEP: third barrier, first operand's region released after ITS reads, pieces from MFMA 23
// on) - the same MFMA order, bit-identical results; +1.5-2.2 % on the products of >= 128 K-tiles, which is where the launcher picks it
template <bool ORDER_B, bool BA = false, bool DEEP = false>      // BA: the instantiation for bias / activation problems (a separate code object: the hot kernel's epilogue stays two modes small)
__global__ __launch_bounds__(256) void gemm_nt_w4_kernel(GemmNTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  const int tiles_m = (p.M + P_BM - 1) / P_BM, tiles_n = p.N / P_BN;
  const int swz = p.xcd_cyclic ? xcd_remap_cyclic(blockIdx.x, gridDim.x) : xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  w4_tile_map(p, swz, tiles_m, tiles_n, tm, tn);
  const int m0 = __builtin_amdgcn_readfirstlane(tm * P_BM), n0 = __builtin_amdgcn_readfirstlane(tn * P_BN);

  const int nt1 = p.K1 / P_BK, nt2 = p.K2 / P_BK, nt = nt1 + nt2;
  const bf16_t* a2 = p.A2;
  if (p.a2_group_n > 0) a2 += (size_t)(n0 / p.a2_group_n) * p.a2_group_stride;
  const bf16_t* a1 = p.A1;
  if (p.a1_group_n > 0) a1 += (size_t)(n0 / p.a1_group_n) * p.a1_group_stride;
  const int srow = lane >> 3, spos = lane & 7;

  // the resource descriptors must sit in SGPRs: a base the compiler cannot prove wave-uniform turns every buffer_load into
  // a readfirstlane "waterfall" loop (seen once: +45 % K-loop time), so uniformity is stated explicitly
  auto uni = [](const void* q) -> void* {
    const unsigned long long v = (unsigned long long)q;
    return (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)v));
  };
  const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc(uni(a1), 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB1 = __builtin_amdgcn_make_buffer_rsrc(uni(p.B1), 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(uni(nt2 ? a2 : a1), 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc(uni(nt2 ? p.B2 : p.B1), 0, (int)0xffffffffu, 0x00020000);
  const unsigned lrow = (unsigned)(wave * 64 + srow);                     // row of this wave's piece 0 inside the 256-row tile
  const unsigned csw[2] = {(unsigned)((spos ^ ((srow >> 1) & 7)) * 16), (unsigned)((spos ^ ((4 + (srow >> 1)) & 7)) * 16)};
  // B tile: LDS row P holds the tile's row L(P) = P with bits 0 and 3 exchanged, its 16-byte chunks XOR-ed with (P >> 4) & 7.  The MFMA
  // column slot s of B fragment f is the tile row 8s + f (so a lane ends up with 8 CONSECUTIVE output columns, one per fragment: 16-byte
  // stores straight from the accumulators, 256 contiguous bytes per row and instruction); with the exchange the 16 lanes of a fragment
  // read sit 2 KiB apart in pairs of adjacent LDS rows and the XOR spreads the pairs over the 8 chunks of a row: conflict-free, and
  // all 8 fragments of a lane are one base address + immediates.  The permutation costs nothing: it lives in the per-lane source offsets.
  const unsigned lrowB_lo = (unsigned)((srow & 1) * 8 + (srow & 6));     // + (pi & 1) + (pi >> 1) * 16 + wave * 64
  const unsigned m_last = (unsigned)(p.M - 1);
  // per-lane byte offsets of the wave's 16 pieces of a K-tile (8 rows x 128 B each): vo[0..7] the B pieces, vo[8..15] the A pieces (rows
  // clamped to M - 1).  No vector ALU work per piece in the loop: the K position is the scalar offset, the descriptor sits in SGPRs.
  auto calc_voff = [&](bool second, unsigned (&vo)[16]) {
    const unsigned lda = (unsigned)(second ? p.lda2 : p.lda1) * 2u, ldb = (unsigned)(second ? p.ldb2 : p.ldb1) * 2u;
    // K-folded problem (GemmNTArgs::b1_fold_n): column group g of the output reads rows 0 .. fold_n - 1 of B1 from column g * b1_fold_koff on
    const unsigned fold_g = (!second && p.b1_fold_n > 0) ? (unsigned)(n0 / p.b1_fold_n) : 0u;
    const unsigned bn0 = (unsigned)n0 - fold_g * (unsigned)(p.b1_fold_n > 0 ? p.b1_fold_n : 0), bkoff = fold_g * (unsigned)p.b1_fold_koff * 2u;
#pragma unroll
    for (int pi = 0; pi < 8; ++pi) {
      vo[8 + pi] = min((unsigned)m0 + lrow + pi * 8u, m_last) * lda + csw[pi & 1];
      vo[pi] = (bn0 + (unsigned)(wave * 64 + (pi >> 1) * 16 + (pi & 1)) + lrowB_lo) * ldb + (unsigned)((spos ^ ((wave * 4 + (pi >> 1)) & 7)) * 16) + bkoff;
    }
  };
  unsigned vo1[16], vo2[16];
  calc_voff(false, vo1);
  calc_voff(true, vo2);
  if (__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)LDS_PTR(void, smem)) != 0) __builtin_trap();      // the asm block addresses the dynamic LDS block from 0 (the kernel has no static LDS)
  // prologue: K-tiles 0 and 1 (q = 0..7: A pieces wave*8 + q, q = 8..15: B pieces)
  auto issue_piece = [&](int t, int q) {
    const bool second = t >= nt1;
    const int k0 = (second ? (t - nt1) : t) * P_BK;
    char* base = smem + (t & 1) * P_STAGE;
    const int piece = wave * 8 + (q & 7);
    const unsigned v = second ? vo2[q < 8 ? 8 + q : q - 8] : vo1[q < 8 ? 8 + q : q - 8];
    if (q < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rA2 : rA1, LDS_PTR(void, base + piece * 1024), 16, v, k0 * 2, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rB2 : rB1, LDS_PTR(void, base + P_TILE + piece * 1024), 16, v, k0 * 2, 0, 0);
  };
#pragma unroll
  for (int q = 0; q < 16; ++q) issue_piece(0, q);
  if (nt > 1) {
#pragma unroll
    for (int q = 0; q < 16; ++q) issue_piece(1, q);
  }
  // (the wait for tile 0 + barrier open the asm block: the accumulator clears and the block's operand set-up below run under the DMA latency)

  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fchk = lane >> 4;
  {
    // operands of the asm block (names fixed by w4_kloop.inc)
    const int fsw = (frow >> 1) & 7;                                       // swizzle term: same for every fragment of a lane
    const unsigned offA = (unsigned)((wr * 128 + frow) * 128), offB = (unsigned)(P_TILE + (wc * 128 + (frow >> 1) * 16 + (frow & 1)) * 128);
    const unsigned cb0 = (unsigned)((fchk ^ fsw) << 4), cb1 = (unsigned)(((4 + fchk) ^ fsw) << 4);
    unsigned w4k_rb[4] = {offB + cb0, offB + cb1, offA + cb0, offA + cb1};
    auto lo32 = [](const void* q) { return __builtin_amdgcn_readfirstlane((int)(unsigned long long)q); };
    auto hi16 = [](const void* q) { return __builtin_amdgcn_readfirstlane((int)(((unsigned long long)q >> 32) & 0xffffu)); };
    const void* pa2 = nt2 ? (const void*)a2 : (const void*)a1; const void* pb2 = nt2 ? (const void*)p.B2 : (const void*)p.B1;
    const int n_loop = max(nt - 2, 0);
    int w4k_na = __builtin_amdgcn_readfirstlane(min(max(nt1 - 2, 0), n_loop));
    int w4k_nb = __builtin_amdgcn_readfirstlane(n_loop - w4k_na);
    const bool first_pair = w4k_na > 0;                                    // the loop starts fetching from the first operand pair
    unsigned w4k_vo[16], w4k_vo2[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { w4k_vo[q] = first_pair ? vo1[q] : vo2[q]; w4k_vo2[q] = vo2[q]; }
    int w4k_dx[4] = {first_pair ? lo32(p.B1) : lo32(pb2), first_pair ? hi16(p.B1) : hi16(pb2), -1, 0x00020000};
    int w4k_dy[4] = {first_pair ? lo32(a1) : lo32(pa2), first_pair ? hi16(a1) : hi16(pa2), -1, 0x00020000};
    const int w4k_dx2[4] = {lo32(pb2), hi16(pb2), -1, 0x00020000};
    const int w4k_dy2[4] = {lo32(pa2), hi16(pa2), -1, 0x00020000};
    int w4k_stg = 0;
    // K byte offset of the tile fetched LAST (the loop adds 128 before its first piece): tile 1 of the pair in use
    int w4k_koff = __builtin_amdgcn_readfirstlane(first_pair ? 128 : (1 - nt1) * 128);
    const int w4k_koff2 = -128;
    const int w4k_has2 = __builtin_amdgcn_readfirstlane(nt >= 2 ? 1 : 0);
    int w4k_pcx[8], w4k_pcy[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      w4k_pcy[q] = __builtin_amdgcn_readfirstlane((wave * 8 + q) * 1024);
      w4k_pcx[q] = __builtin_amdgcn_readfirstlane(P_TILE + (wave * 8 + q) * 1024);
    }
    if constexpr (DEEP) {
      if constexpr (ORDER_B) W4K_RUN(W4K_TEXT_BFIRST_DEEP);
      else W4K_RUN(W4K_TEXT_AFIRST_DEEP);
    } else {
      if constexpr (ORDER_B) W4K_RUN(W4K_TEXT_BFIRST);
      else W4K_RUN(W4K_TEXT_AFIRST);
    }
  }
  __builtin_amdgcn_sched_barrier(0);                                             // no accumulator read may be scheduled above the block
  // Accumulator layout (B tile rows interleaved, see above): acc[i][j][r] of lane (frow, fchk) is row i*16 + 4*fchk + r, column 8*frow + j
  // of the wave's 128x128 block - the 8 fragments j of one (i, r) are 8 CONSECUTIVE columns.
  auto staged_epi = [&]() {
    char* stg = smem + wave * 32768;
    const int ncol0 = n0 + wc * 128;
    const bool special = p.act == OPADPO_ACT_SWIGLU_PAIR || p.act == OPADPO_ACT_SWIGLU_BWD || ((p.rope_pos || p.rope_cos) && n0 < p.rope_cols);
    if ((!p.R || w4_direct_resid_ok(p, BA)) && !special) {
      w4_direct_epilogue<BA ? 1 : OPADPO_W4_RPD, BA>(p, acc, m0, ncol0, wr, frow, fchk);
      return;
    }
    if (w4_direct_swiglu_bwd_ok(p) && !p.swiglu_bwd_staged) {
      w4_direct_epilogue_swiglu_bwd(p, acc, m0, ncol0, wr, frow, fchk);
      return;
    }
    if (w4_direct_swiglu_pair_ok(p) && !p.swiglu_bwd_staged) {      // (swiglu_bwd_staged doubles as "staged epilogues": cross-check of the direct forms)
      w4_direct_epilogue_swiglu_pair(p, acc, m0, ncol0, wr, frow, fchk);
      return;
    }
    // Staged epilogues (residual operand, SwiGLU pair / backward, rotary embedding): each wave passes its 128x128 block through its own
    // 32 KiB of the (now dead) stages and works on whole rows: 256 B (bf16) / 512 B (fp32) contiguous per row.
    __builtin_amdgcn_s_barrier();                       // every wave has read its last fragments out of the stages
    auto half = [&](auto HF, auto MD_) {
      constexpr int hf = decltype(HF)::value, md = decltype(MD_)::value;
#pragma unroll
      for (int i4 = 0; i4 < 4; ++i4) {
        float v[8][4];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          acc_read4(acc[hf * 4 + i4][j], v[j]);
          epi_pre4<md>(p, 0, v[j]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = i4 * 16 + 4 * fchk + r;
          *(float4*)(stg + row * 512 + (((2 * frow) ^ (row & 15)) << 4)) = make_float4(v[0][r], v[1][r], v[2][r], v[3][r]);
          *(float4*)(stg + row * 512 + (((2 * frow + 1) ^ (row & 15)) << 4)) = make_float4(v[4][r], v[5][r], v[6][r], v[7][r]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int mbase = m0 + wr * 128 + hf * 64;
      if (p.out_f32) {
#pragma unroll 4
        for (int ps = 0; ps < 32; ++ps) {
          const int row = ps * 2 + (lane >> 5), c = lane & 31, m = mbase + row, n = ncol0 + c * 4;
          float4 v = *(const float4*)(stg + row * 512 + ((c ^ (row & 15)) << 4));
          if (m < p.M) {
            if (p.R) {
              if (p.r_f32) {
                const float4 r = *(const float4*)((const float*)p.R + (size_t)m * p.ldr + n);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
              } else {
                const uint2 r = *(const uint2*)((const bf16_t*)p.R + (size_t)m * p.ldr + n);
                v.x += __uint_as_float(r.x << 16); v.y += __uint_as_float(r.x & 0xffff0000u);
                v.z += __uint_as_float(r.y << 16); v.w += __uint_as_float(r.y & 0xffff0000u);
              }
            }
            *(float4*)((float*)p.C + (size_t)m * p.ldc + n) = v;
          }
        }
      } else {
#pragma unroll 4
        for (int ps = 0; ps < 16; ++ps) {
          const int row = ps * 4 + (lane >> 4), g = lane & 15, m = mbase + row, n = ncol0 + g * 8;
          const float4 lo = *(const float4*)(stg + row * 512 + (((2 * g) ^ (row & 15)) << 4));
          const float4 hi = *(const float4*)(stg + row * 512 + (((2 * g + 1) ^ (row & 15)) << 4));
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          if (m < p.M) {
            if (p.R) {
              if (p.r_f32) {
                const float4 r0 = *(const float4*)((const float*)p.R + (size_t)m * p.ldr + n);
                const float4 r1 = *(const float4*)((const float*)p.R + (size_t)m * p.ldr + n + 4);
                v[0] += r0.x; v[1] += r0.y; v[2] += r0.z; v[3] += r0.w; v[4] += r1.x; v[5] += r1.y; v[6] += r1.z; v[7] += r1.w;
              } else {
                const uint4 r = *(const uint4*)((const bf16_t*)p.R + (size_t)m * p.ldr + n);
                v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
                v[4] += __uint_as_float(r.z << 16); v[5] += __uint_as_float(r.z & 0xffff0000u);
                v[6] += __uint_as_float(r.w << 16); v[7] += __uint_as_float(r.w & 0xffff0000u);
              }
            }
            uint4 o;
            o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]); o.z = pack_bf2(v[4], v[5]); o.w = pack_bf2(v[6], v[7]);
            *(uint4*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = o;
          }
        }
      }
      if (hf == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the second half overwrites the staging rows
    };
    if (!p.out_f32 && (!p.R || p.act == OPADPO_ACT_SWIGLU_BWD)) {
      // bf16 result without residual: the whole 128x128 block fits the wave's 32 KiB as bf16 -> one LDS round trip, half the bytes
      // (32-byte groups XOR-swizzled by row & 7; 16-byte writes of a lane's 8 columns, 16-byte reads along rows)
      epi_dispatch_plain(p, [&](auto MD_) {
        constexpr int md = decltype(MD_)::value;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float v[8][4];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            acc_read4(acc[i][j], v[j]);
            epi_pre4<md>(p, 0, v[j]);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = i * 16 + 4 * fchk + r;
            uint4 o;
            o.x = pack_bf2(v[0][r], v[1][r]); o.y = pack_bf2(v[2][r], v[3][r]); o.z = pack_bf2(v[4][r], v[5][r]); o.w = pack_bf2(v[6][r], v[7][r]);
            *(uint4*)(stg + row * 256 + (((frow >> 1) ^ (row & 7)) << 5) + (frow & 1) * 16) = o;
          }
          __builtin_amdgcn_sched_barrier(0);      // one row block at a time: keeps 8, not 64, accumulators live in VGPRs
        }
      });
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int mb = m0 + wr * 128;
      if (p.act == OPADPO_ACT_SWIGLU_PAIR) {
        // fused SwiGLU: the wave's 128 columns are [gate c0..c0+63 | up c0..c0+63] (weight rows permuted by the caller), the
        // output has N/2 columns.  silu(gate) * up on the bf16-ROUNDED staged values, same formula as silu_mul_fwd_kernel ->
        // bit-identical to the two-kernel path.  8 rows x 8 chunks per pass.
        const int row0 = lane >> 3, g = lane & 7;
        const int ocol = ncol0 / 2 + g * 8;
#pragma unroll 4
        for (int ps = 0; ps < 16; ++ps) {
          const int row = ps * 8 + row0, m = mb + row;
          const char* rp = stg + row * 256;
          float gt[8], up[8], o[8];
          unpack8(*(const uint4*)(rp + (((g >> 1) ^ (row & 7)) << 5) + (g & 1) * 16), gt);
          unpack8(*(const uint4*)(rp + ((((g + 8) >> 1) ^ (row & 7)) << 5) + (g & 1) * 16), up);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = gt[e] / (1.0f + __expf(-gt[e])) * up[e];
          if (m < p.M) store16((bf16_t*)p.C + (size_t)m * p.ldc + ocol, pack8(o), p.store_nt);
        }
        return;
      }
      if (p.act == OPADPO_ACT_SWIGLU_BWD) {
        // fused SwiGLU backward (dgrad of the down projection): the staged bf16-ROUNDED block is d_act[:, ncol0..+127]; R holds the
        // stored pre-activations [gate | up] ([M, 2N]) and C receives [d_gate | d_up] - the arithmetic of silu_mul_bwd_kernel on the
        // same rounded operands, so the result is bit-identical to the two-kernel path without d_act ever reaching HBM.
        const int row0 = lane >> 4, g = lane & 15;
        const bf16_t* gu = (const bf16_t*)p.R;
#pragma unroll 4
        for (int ps = 0; ps < 32; ++ps) {
          const int row = ps * 4 + row0, m = mb + row;
          if (m >= p.M) continue;
          const size_t n = (size_t)ncol0 + g * 8;
          float dd[8], gt[8], up[8], dg[8], du[8];
          unpack8(*(const uint4*)(stg + row * 256 + (((g >> 1) ^ (row & 7)) << 5) + (g & 1) * 16), dd);
          unpack8(*(const uint4*)(gu + (size_t)m * p.ldr + n), gt);
          unpack8(*(const uint4*)(gu + (size_t)m * p.ldr + p.N + n), up);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float sg = 1.0f / (1.0f + __expf(-gt[e]));
            const float silu = gt[e] * sg;
            du[e] = dd[e] * silu;
            dg[e] = dd[e] * up[e] * sg * (1.0f + gt[e] * (1.0f - sg));
          }
          *(uint4*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = pack8(dg);
          *(uint4*)((bf16_t*)p.C + (size_t)m * p.ldc + p.N + n) = pack8(du);
        }
        return;
      }
      if (p.rope_pos && ncol0 < p.rope_cols) {
        // fused rotary embedding WITHOUT table traffic (round 3; the table form below pays 64 KiB of fp32 cos / sin reads per 32 KiB of output
        // and a position computation per row: 7.7 us per block, as much as the in-place rope kernel costs).  Lane (row0 = lane >> 4, g =
        // lane & 15) rotates 8 columns of rows row0, row0 + 4, ...; the angles come from v_sin / v_cos of the fractional revolution
        // pos * inv_freq / 2pi, evaluated per (row, frequency).  (An angle-addition recurrence along the rows was 1.4 us per block cheaper but
        // made a row's last bit depend on its offset inside the 128-row block, i.e. on the batch around it.)
        // Same arithmetic on the bf16-ROUNDED staged values as rope_kernel; angles differ from the table's by <= 2e-4 rad.
        const int row0 = lane >> 4, g = lane & 15, gh = g & 7;
        const float sign = g < 8 ? -1.0f : 1.0f;
        float frev[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) frev[e] = __builtin_amdgcn_exp2f(-(float)(2 * (gh * 8 + e)) * (1.0f / 128.0f) * p.rope_l2theta) * 0.15915494309189535f;
        int pos = p.rope_pos[min(mb + row0, p.M - 1)];
#pragma unroll 2
        for (int ps = 0; ps < 32; ++ps) {
          const int row = ps * 4 + row0, m = mb + row;
          const int pn = p.rope_pos[min(m + 4, p.M - 1)];                    // next row of this lane: requested before the math of this one
          const char* rp = stg + row * 256;
          float xs[8], xp[8], o[8];
          unpack8(*(const uint4*)(rp + (((g >> 1) ^ (row & 7)) << 5) + (g & 1) * 16), xs);
          unpack8(*(const uint4*)(rp + ((((g ^ 8) >> 1) ^ (row & 7)) << 5) + (g & 1) * 16), xp);
          const float posf = (float)pos;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            // the angle of (position, frequency) is a function of those two numbers ONLY - no recurrence along the rows of the tile - so a
            // row's result does not depend on where the row sits in the batch (tests/test_fullsize_gpu.py P3: bit-equal)
            const float x = __builtin_amdgcn_fractf(posf * frev[e]);
            o[e] = xs[e] * __builtin_amdgcn_cosf(x) + sign * (xp[e] * __builtin_amdgcn_sinf(x));
          }
          if (m < p.M) store16((bf16_t*)p.C + (size_t)m * p.ldc + ncol0 + g * 8, pack8(o), p.store_nt);
          pos = pn;
        }
        return;
      }
      if (p.rope_cos && ncol0 < p.rope_cols) {
        // fused rotary embedding: the wave's 128 columns are one head; out[d] = x[d] cos - x[d+64] sin, out[d+64] = x[d+64] cos + x[d] sin
        // on the bf16-ROUNDED staged values (what rope_kernel reads back from HBM), position = row % L with the packed-response restart
        const int row0 = lane >> 4, g = lane & 15, gh = g & 7;
        const float sign = g < 8 ? -1.0f : 1.0f;
        int pos = (mb + row0) % p.rope_L;
        const int lim = p.rope_seg_prefix + p.rope_seg_len;
        int rem = (p.rope_seg_len > 0 && pos >= p.rope_seg_prefix) ? (pos - p.rope_seg_prefix) % p.rope_seg_len : 0;
#pragma unroll 4
        for (int ps = 0; ps < 32; ++ps) {
          const int row = ps * 4 + row0, m = mb + row;
          const int p2 = (p.rope_seg_len > 0 && pos >= lim) ? p.rope_seg_prefix + rem : pos;
          const char* rp = stg + row * 256;
          float xs[8], xp[8], o[8];
          unpack8(*(const uint4*)(rp + (((g >> 1) ^ (row & 7)) << 5) + (g & 1) * 16), xs);
          unpack8(*(const uint4*)(rp + ((((g ^ 8) >> 1) ^ (row & 7)) << 5) + (g & 1) * 16), xp);
          const float4 c0 = *(const float4*)(p.rope_cos + (size_t)p2 * 64 + gh * 8), c1 = *(const float4*)(p.rope_cos + (size_t)p2 * 64 + gh * 8 + 4);
          const float4 s0 = *(const float4*)(p.rope_sin + (size_t)p2 * 64 + gh * 8), s1 = *(const float4*)(p.rope_sin + (size_t)p2 * 64 + gh * 8 + 4);
          const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = xs[e] * cs[e] + sign * (xp[e] * sn[e]);
          if (m < p.M) *(uint4*)((bf16_t*)p.C + (size_t)m * p.ldc + ncol0 + g * 8) = pack8(o);
          pos += 4; rem += 4;
          if (pos >= p.rope_L) { pos -= p.rope_L; rem = (p.rope_seg_len > 0 && pos >= p.rope_seg_prefix) ? (pos - p.rope_seg_prefix) % p.rope_seg_len : 0; }
          else if (p.rope_seg_len > 0) {
            if (pos >= p.rope_seg_prefix && pos - 4 < p.rope_seg_prefix) rem = pos - p.rope_seg_prefix;
            while (rem >= p.rope_seg_len) rem -= p.rope_seg_len;
          }
        }
        return;
      }
      if ((unsigned long long)p.M * (unsigned)p.ldc * 2ull < 0xffffffffull) {
        // rows through a descriptor that ends after row M-1: 32-bit offsets (one add per row group), no predicate, 8 reads then 8 stores
        typedef __attribute__((ext_vector_type(4))) unsigned u32x4s_t;
        const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(uni(p.C), 0, (int)((unsigned)p.M * (unsigned)p.ldc * 2u), 0x00020000);
        const int row0 = lane >> 4, g = lane & 15;
        const unsigned step = 4u * (unsigned)p.ldc * 2u;
        unsigned vo = (unsigned)(mb + row0) * (unsigned)p.ldc * 2u + (unsigned)(ncol0 + g * 8) * 2u;
        const char* src = stg + row0 * 256 + (g & 1) * 16;
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) {
          u32x4s_t o[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int ps = pg * 8 + u;                  // row = ps*4 + row0: row & 7 = (ps & 1)*4 + row0
            o[u] = *(const u32x4s_t*)(src + ps * 1024 + (((g >> 1) ^ (((ps & 1) << 2) | row0)) << 5));
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (p.store_nt) __builtin_amdgcn_raw_buffer_store_b128(o[u], rC, vo, 0, 2);
            else __builtin_amdgcn_raw_buffer_store_b128(o[u], rC, vo, 0, 0);
            vo += step;
          }
        }
      } else {
#pragma unroll 8
        for (int ps = 0; ps < 32; ++ps) {
          const int row = ps * 4 + (lane >> 4), g = lane & 15, m = mb + row;
          const uint4 o = *(const uint4*)(stg + row * 256 + (((g >> 1) ^ (row & 7)) << 5) + (g & 1) * 16);
          if (m < p.M) *(uint4*)((bf16_t*)p.C + (size_t)m * p.ldc + ncol0 + g * 8) = o;
        }
      }
      return;
    }
    epi_dispatch_plain(p, [&](auto MD_) {
      half(std::integral_constant<int, 0>{}, MD_);
      half(std::integral_constant<int, 1>{}, MD_);
    });
  };
  staged_epi();
}

// ------------------------------------------------------------------------------------------
// gemm_nt "w4s" kernel (round 5): the 4-wave 256x256 kernel as a STREAMING persistent kernel for the plain products (alpha-only, bf16 or fp32
// out, no residual / activation / rotary epilogue - those need the LDS stages for staging).  One workgroup per CU walks the tiles b, b + grid,
// b + 2 grid, .. of the same XCD-aware tile order; the K-tile stream never drains at an output-tile boundary: the last two K-tiles of a tile fetch
// K-tiles 0 and 1 of the NEXT tile, the last one reads the next tile's first fragments, and the (direct, LDS-free) epilogue stores the finished
// tile while those pieces fly.  What a tile of the one-tile-per-workgroup kernel pays besides its K-loop - workgroup dispatch, address set-up,
// the 2-K-tile pipeline fill with nothing to compute, the drain - is paid once per workgroup instead of once per tile.  The stores are issued
// AFTER the next tile's first pieces, so the in-order vmcnt waits of the next tile's loop never sit behind a store that has not been issued yet (what
// sank the round-4 persistent kernel).  Bit-identical to gemm_nt_w4_kernel (same k order; tests/test_ops_gpu.py).  Needs K1 >= 3 K-tiles.
// Where a tile's cycles go (OPADPO_W4S_DIAG build + tools/w4s_diag.py, profiles/r05c_w4s_tile_anatomy.txt; K = 4096, wave 0): K-loop 139 k
// (2.18 k per K-tile), epilogue 7.7 k for a bf16 tile and 13-21 k for an fp32 tile, per-tile set-up 1.4 k.  The epilogue is the CU's STORE rate -
// 128 KiB per tile at ~17 B per cycle and CU (the ~35 GB/s per-CU vector-memory figure of the decode kernels) - not a chip-wide write burst:
// starting the workgroups of an XCD in 2-8 phases up to a burst apart changed nothing for bf16 tiles (7.74 k in every variant; built, measured,
// removed).  Hiding it needs the finished tile parked somewhere for ~4 K-tiles; registers (256 accumulators + 128 fragment registers) and LDS
// (2 x 64 KiB stages) are both full.
// ------------------------------------------------------------------------------------------
#ifndef OPADPO_W4S_DIAG
#define OPADPO_W4S_DIAG 0
#endif
#if OPADPO_W4S_DIAG
// diagnostics build (tools/build_diag.sh w4sdiag:"-DOPADPO_W4S_DIAG=1"): cycles of wave 0 of every workgroup spent in [0] the K-loop block, [1] the epilogue,
// [2] the per-tile set-up between them, [3] tiles - summed over the launch; read with opadpo_debug_w4s_read
__device__ unsigned long long g_w4s_diag[4];
extern "C" int opadpo_debug_w4s_read(unsigned long long* out4, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_w4s_diag), 32);
  if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_w4s_diag), z, 32); }
  return (int)e;
}
#endif
template <bool ORDER_B, int EPI = 0, bool DEEP = false>      // DEEP: the K-loop's DEEP text (products of >= 128 K-tiles; see gemm_nt_w4_kernel).  EPI 0: plain / fp32 residual / SwiGLU backward epilogues (the hot instantiation, at its register limit); 2: SwiGLU pair only; 3: table-free rotary embedding (own code objects)
__global__ __launch_bounds__(256) void gemm_nt_w4s_kernel(GemmNTArgs p, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int tiles_m = (p.M + P_BM - 1) / P_BM, tiles_n = p.N / P_BN;
  auto tile_of = [&](int b, int& m0, int& n0) {
    const int swz = p.xcd_cyclic ? xcd_remap_cyclic(b, n_tiles) : xcd_remap(b, n_tiles);
    int tm, tn;
    w4_tile_map(p, swz, tiles_m, tiles_n, tm, tn);
    m0 = __builtin_amdgcn_readfirstlane(tm * P_BM);
    n0 = __builtin_amdgcn_readfirstlane(tn * P_BN);
  };
  const int nt1 = p.K1 / P_BK, nt2 = p.K2 / P_BK;
  const int srow = lane >> 3, spos = lane & 7;
  const unsigned lrow = (unsigned)(wave * 64 + srow);
  const unsigned csw[2] = {(unsigned)((spos ^ ((srow >> 1) & 7)) * 16), (unsigned)((spos ^ ((4 + (srow >> 1)) & 7)) * 16)};
  const unsigned lrowB_lo = (unsigned)((srow & 1) * 8 + (srow & 6));
  const unsigned m_last = (unsigned)(p.M - 1);
  auto calc_voff = [&](bool second, int m0, int n0, unsigned (&vo)[16]) {      // as in gemm_nt_w4_kernel: vo[0..7] B pieces, vo[8..15] A pieces
    const unsigned lda = (unsigned)(second ? p.lda2 : p.lda1) * 2u, ldb = (unsigned)(second ? p.ldb2 : p.ldb1) * 2u;
#pragma unroll
    for (int pi = 0; pi < 8; ++pi) {
      vo[8 + pi] = min((unsigned)m0 + lrow + pi * 8u, m_last) * lda + csw[pi & 1];
      vo[pi] = ((unsigned)n0 + (unsigned)(wave * 64 + (pi >> 1) * 16 + (pi & 1)) + lrowB_lo) * ldb + (unsigned)((spos ^ ((wave * 4 + (pi >> 1)) & 7)) * 16);
    }
  };
  auto lo32 = [](const void* q) { return __builtin_amdgcn_readfirstlane((int)(unsigned long long)q); };
  auto hi16 = [](const void* q) { return __builtin_amdgcn_readfirstlane((int)(((unsigned long long)q >> 32) & 0xffffu)); };
  auto a1_of = [&](int n0) -> const bf16_t* { return p.a1_group_n > 0 ? p.A1 + (size_t)(n0 / p.a1_group_n) * p.a1_group_stride : p.A1; };
  auto a2_of = [&](int n0) -> const bf16_t* { if (!nt2) return a1_of(n0); return p.a2_group_n > 0 ? p.A2 + (size_t)(n0 / p.a2_group_n) * p.a2_group_stride : p.A2; };
  if (__builtin_amdgcn_readfirstlane((int)(unsigned)(size_t)LDS_PTR(void, smem)) != 0) __builtin_trap();

  int b = blockIdx.x, m0, n0;
  tile_of(b, m0, n0);
  unsigned w4k_vo[16];
  calc_voff(false, m0, n0, w4k_vo);
  {   // pipeline fill, once per workgroup: K-tiles 0 and 1 of the first tile (K1 >= 3 K-tiles: both from the first operand pair)
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(w4_uniform_ptr(a1_of(n0)), 0, (int)0xffffffffu, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(w4_uniform_ptr(p.B1), 0, (int)0xffffffffu, 0x00020000);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        char* base = smem + t * P_STAGE;
        const int piece = wave * 8 + (q & 7);
        if (q < 8) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(void, base + piece * 1024), 16, w4k_vo[8 + q], t * P_BK * 2, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, LDS_PTR(void, base + P_TILE + piece * 1024), 16, w4k_vo[q - 8], t * P_BK * 2, 0, 0);
      }
  }
  f32x4_t acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t w4k_fr[16];
#pragma unroll
  for (int f = 0; f < 16; ++f) w4k_fr[f] = bf16x8_t{};
  const int frow = lane & 15, fchk = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  const unsigned offA = (unsigned)((wr * 128 + frow) * 128), offB = (unsigned)(P_TILE + (wc * 128 + (frow >> 1) * 16 + (frow & 1)) * 128);
  const unsigned cb0 = (unsigned)((fchk ^ fsw) << 4), cb1 = (unsigned)(((4 + fchk) ^ fsw) << 4);
  unsigned w4k_rb[4] = {offB + cb0, offB + cb1, offA + cb0, offA + cb1};
  int w4k_dx[4] = {lo32(p.B1), hi16(p.B1), -1, 0x00020000};
  int w4k_dy[4] = {lo32(a1_of(n0)), hi16(a1_of(n0)), -1, 0x00020000};
  int w4k_stg = 0;
  int w4k_koff = 128;                                          // K byte offset of the tile fetched last (the loop adds 128 before its first piece)
  int w4k_pcx[8], w4k_pcy[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    w4k_pcy[q] = __builtin_amdgcn_readfirstlane((wave * 8 + q) * 1024);
    w4k_pcx[q] = __builtin_amdgcn_readfirstlane(P_TILE + (wave * 8 + q) * 1024);
  }
  int w4k_first = 1;
#if OPADPO_W4S_DIAG
  unsigned long long dg_asm = 0, dg_epi = 0, dg_glue = 0, dg_n = 0, dg_t = __builtin_readcyclecounter();
#endif
  for (;;) {
    const int nb = b + (int)gridDim.x;
    // 1 = there is a next tile (s37 of the streaming text).  Computed on the scalar unit by hand: hipcc lowers the select of this uniform compare
    // through a vector register, which the scalar asm operand below cannot take ("illegal VGPR to SGPR copy")
    int w4k_has2;
    asm volatile("s_cmp_lt_i32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(w4k_has2) : "s"(nb), "s"(n_tiles) : "scc");
    int m0n = m0, n0n = n0;
    if (w4k_has2) tile_of(nb, m0n, n0n);
    unsigned w4k_vo2[16], w4k_vo3[16];
    calc_voff(true, m0, n0, w4k_vo2);                          // second operand pair of THIS tile (the K-concatenated LoRA tail)
    calc_voff(false, m0n, n0n, w4k_vo3);                       // first operand pair of the NEXT tile
    const void* pb2 = nt2 ? (const void*)p.B2 : (const void*)p.B1;
    const int w4k_dx2[4] = {lo32(pb2), hi16(pb2), -1, 0x00020000};
    const int w4k_dy2[4] = {lo32(a2_of(n0)), hi16(a2_of(n0)), -1, 0x00020000};
    const int w4k_dx3[4] = {lo32(p.B1), hi16(p.B1), -1, 0x00020000};
    const int w4k_dy3[4] = {lo32(a1_of(n0n)), hi16(a1_of(n0n)), -1, 0x00020000};
    int w4k_na = __builtin_amdgcn_readfirstlane(nt1 - 3), w4k_nb = __builtin_amdgcn_readfirstlane(nt2);
    // (loop-carried scalars: stated wave-uniform once more, or the PHIs of the tile loop end up in vector registers)
    w4k_stg = __builtin_amdgcn_readfirstlane(w4k_stg); w4k_koff = __builtin_amdgcn_readfirstlane(w4k_koff);
    w4k_first = __builtin_amdgcn_readfirstlane(w4k_first);
#pragma unroll
    for (int q = 0; q < 4; ++q) { w4k_dx[q] = __builtin_amdgcn_readfirstlane(w4k_dx[q]); w4k_dy[q] = __builtin_amdgcn_readfirstlane(w4k_dy[q]); }
#if OPADPO_W4S_DIAG
    { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); dg_glue += t_ - dg_t; dg_t = t_; __builtin_amdgcn_sched_barrier(0); }
#endif
    if constexpr (DEEP) {
      if constexpr (ORDER_B) W4S_RUN(W4S_TEXT_BFIRST_DEEP);
      else W4S_RUN(W4S_TEXT_AFIRST_DEEP);
    } else {
      if constexpr (ORDER_B) W4S_RUN(W4S_TEXT_BFIRST);
      else W4S_RUN(W4S_TEXT_AFIRST);
    }
    __builtin_amdgcn_sched_barrier(0);
#if OPADPO_W4S_DIAG
    { const unsigned long long t_ = __builtin_readcyclecounter(); dg_asm += t_ - dg_t; dg_t = t_; ++dg_n; __builtin_amdgcn_sched_barrier(0); }
#endif
    if constexpr (EPI == 2) w4_direct_epilogue_swiglu_pair(p, acc, m0, n0 + wc * 128, wr, frow, fchk);
    else if constexpr (EPI == 3) {
      if (n0 + wc * 128 < p.rope_cols) w4_direct_epilogue_rope_pos(p, acc, m0, n0 + wc * 128, wr, frow, fchk);
      else w4_direct_epilogue<1, false, true>(p, acc, m0, n0 + wc * 128, wr, frow, fchk);
    }
    else if (p.act == OPADPO_ACT_SWIGLU_BWD) w4_direct_epilogue_swiglu_bwd(p, acc, m0, n0 + wc * 128, wr, frow, fchk);
    else w4_direct_epilogue<OPADPO_W4S_RPD, false>(p, acc, m0, n0 + wc * 128, wr, frow, fchk);
    __builtin_amdgcn_sched_barrier(0);
#if OPADPO_W4S_DIAG
    { const unsigned long long t_ = __builtin_readcyclecounter(); dg_epi += t_ - dg_t; dg_t = t_; __builtin_amdgcn_sched_barrier(0); }
#endif
    if (!w4k_has2) break;
    b = nb; m0 = m0n; n0 = n0n;
    w4k_first = 0;
  }
#if OPADPO_W4S_DIAG
  if (tid == 0) { atomicAdd(&g_w4s_diag[0], dg_asm); atomicAdd(&g_w4s_diag[1], dg_epi); atomicAdd(&g_w4s_diag[2], dg_glue); atomicAdd(&g_w4s_diag[3], dg_n); }
#endif
}

// ------------------------------------------------------------------------------------------
// gemm_nt "p8" kernel: the 256x256 / BK 64 / 8-wave (2x4, 128x64 per wave) geometry of the ping-pong kernel with FOUR
// phases per K-tile instead of one.  A phase = read block (the ds_read_b128 of one 64x32 quadrant of the wave's C block,
// TWO LDS-DMA pieces, a counted s_waitcnt vmcnt) | barrier | 16 MFMAs under s_setprio 1 | barrier.  The wave groups
// G0 = waves 0-3 (rows 0..127) and G1 = waves 4-7 run the same program one barrier apart, so on every SIMD one wave is in
// its MFMA block while its sibling is in its read block: a ~60-cycle LDS-DMA issue never sits between two MFMAs of the
// only wave that could feed the matrix pipe, and read block (2 DMA + 4..12 reads) and MFMA block (16 x 17 cycles) are the
// same length.  Quadrant order (a,b) = (0,0) (0,1) (1,1) (1,0): reads 12 / 4 / 8 / 0 fragments (both B halves stay in
// registers), so every region of a stage is dead after the third phase and is re-staged for tile t+2 at least 3 slots
// (a slot = half a phase) after its last reader and at least 4 of the issuing wave's read blocks before its first
// reader: ONE rule, s_waitcnt vmcnt(8) (2 pieces x 4 read blocks stay in flight) in every read block.
//   stage regions (8 KiB = 8 DMA pieces each): A rows 0-63 / 64-127 (read by G0 only), 128-191 / 192-255 (G1 only),
//   B rows of the even 32-row blocks (b = 0) split in two, B rows of the odd blocks (b = 1) split in two.
//   issue schedule (tile T = the tile being computed):     p0            p1            p2            p3
//       G0 (2 pieces per wave and phase)                Bb1.h2(T+1)   A1.a1(T+1)    A1.a0(T+2)    Bb0.h2(T+2)
//       G1                                              A0.a1(T+1)    A0.a0(T+2)    Bb0.h1(T+2)   Bb1.h1(T+2)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void gemm_nt_p8_kernel(GemmNTArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  const int tiles_m = (p.M + P_BM - 1) / P_BM, tiles_n = p.N / P_BN;
  const int swz = xcd_remap(blockIdx.x, gridDim.x);
  const int width = GROUP_M * tiles_n;
  const int group_id = swz / width;
  const int first_m = group_id * GROUP_M;
  const int gsz = min(tiles_m - first_m, GROUP_M);
  const int tm = first_m + (swz % width) % gsz;
  const int tn = (swz % width) / gsz;
  const int m0 = tm * P_BM, n0 = tn * P_BN;

  const int nt1 = p.K1 / P_BK, nt2 = p.K2 / P_BK, nt = nt1 + nt2;
  const bf16_t* a2 = p.A2;
  if (p.a2_group_n > 0) a2 += (size_t)(n0 / p.a2_group_n) * p.a2_group_stride;
  if (p.a1_group_n > 0) p.A1 += (size_t)(n0 / p.a1_group_n) * p.a1_group_stride;
  const int srow = lane >> 3, spos = lane & 7;

  const __amdgpu_buffer_rsrc_t rA1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.A1, 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.B1, 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc((void*)(nt2 ? a2 : p.A1), 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB2 = __builtin_amdgcn_make_buffer_rsrc((void*)(nt2 ? p.B2 : p.B1), 0, (int)0xffffffffu, 0x00020000);
  const unsigned csw[2] = {(unsigned)((spos ^ ((srow >> 1) & 7)) * 16), (unsigned)((spos ^ ((4 + (srow >> 1)) & 7)) * 16)};
  const unsigned m_last = (unsigned)(p.M - 1);
  // one 1-KiB piece (8 rows x 128 B) of K-tile t: A piece pc = rows pc*8.. of the 256-row A tile, B piece likewise
  auto issue_A = [&](int t, int pc) {
    if (t >= nt) return;
    const bool second = t >= nt1;
    const int k0 = (second ? (t - nt1) : t) * P_BK;
    const unsigned ld2 = (unsigned)(second ? p.lda2 : p.lda1) * 2u;
    const unsigned row = min((unsigned)m0 + pc * 8u + srow, m_last);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rA2 : rA1, LDS_PTR(void, smem + (t & 1) * P_STAGE + pc * 1024), 16,
                                             row * ld2 + csw[pc & 1], k0 * 2, 0, 0);
  };
  auto issue_B = [&](int t, int pc) {
    if (t >= nt) return;
    const bool second = t >= nt1;
    const int k0 = (second ? (t - nt1) : t) * P_BK;
    const unsigned ld2 = (unsigned)(second ? p.ldb2 : p.ldb1) * 2u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(second ? rB2 : rB1, LDS_PTR(void, smem + (t & 1) * P_STAGE + P_TILE + pc * 1024), 16,
                                             (unsigned)srow * ld2 + csw[pc & 1], ((unsigned)n0 + pc * 8u) * ld2 + k0 * 2, 0, 0);
  };
  const int wl = wave & 3;
  // region -> this wave's two pieces (e = wl*2 + j, j = 0,1)
  auto piece_A = [&](int half, int a, int j) { return half * 16 + a * 8 + wl * 2 + j; };                 // A{half}.a{a}
  auto piece_B = [&](int b, int h, int j) {                                                              // Bb{b}.h{h+1}
    const int e = wl * 2 + j;                       // 0..7 inside the 8-piece half-region
    return h * 16 + (e >> 2) * 8 + b * 4 + (e & 3);
  };

  f32x4_t acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t af[2][4], bfr[2][4];
  const int frow = lane & 15, fchk = lane >> 4;
  const int fsw = (frow >> 1) & 7;
  const int offA = (wr * 128 + frow) * 128, offB = P_TILE + (wc * 64 + frow) * 128;
  auto read_A = [&](int t, int a) {      // 8 fragments: rows a*64 + i*16 of the wave's 128
    const char* st = smem + (t & 1) * P_STAGE + offA + a * 64 * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i) af[kk][i] = *(const bf16x8_t*)(st + i * 2048 + (((kk * 4 + fchk) ^ fsw) << 4));
  };
  auto read_B = [&](int t, int b) {      // 4 fragments: cols b*32 + j*16 of the wave's 64
    const char* st = smem + (t & 1) * P_STAGE + offB + b * 32 * 128;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[kk][b * 2 + j] = *(const bf16x8_t*)(st + j * 2048 + (((kk * 4 + fchk) ^ fsw) << 4));
  };
  auto mfma_q = [&](int a, int b) {
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[a * 4 + i][b * 2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][b * 2 + j], af[kk][i], acc[a * 4 + i][b * 2 + j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
  };
#define P8_BAR()                                   \
  do {                                             \
    __builtin_amdgcn_sched_barrier(0);             \
    __builtin_amdgcn_s_barrier();                  \
    __builtin_amdgcn_sched_barrier(0);             \
  } while (0)
  // phase tail: counted wait for this wave's older DMA pieces, barrier, fragments landed, MFMAs, barrier
#define P8_COMPUTE(a, b, steady)                                                        \
  do {                                                                                  \
    if (steady) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                      \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             \
    P8_BAR();                                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                \
    __builtin_amdgcn_sched_barrier(0);                                                  \
    mfma_q(a, b);                                                                       \
    P8_BAR();                                                                           \
  } while (0)

  // prologue: all of tile 0, and the regions of tile 1 that the in-loop schedule does not deliver during tile 0
  // (A0.a0, A1.a0, Bb0.h1, Bb0.h2, Bb1.h1); in-loop: A0.a1(1), A1.a1(1), Bb1.h2(1)
#pragma unroll
  for (int q = 0; q < 4; ++q) { issue_A(0, wave * 4 + q); issue_B(0, wave * 4 + q); }
  if (wr == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j) { issue_A(1, piece_A(0, 0, j)); issue_B(1, piece_B(0, 0, j)); issue_B(1, piece_B(1, 0, j)); }
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) { issue_A(1, piece_A(1, 0, j)); issue_B(1, piece_B(0, 1, j)); }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  P8_BAR();

  if (wr == 0) {
    for (int t = 0; t < nt; ++t) {
      const bool steady = t + 2 < nt;
      read_A(t, 0); read_B(t, 0);
      __builtin_amdgcn_sched_barrier(0);
      issue_B(t + 1, piece_B(1, 1, 0)); issue_B(t + 1, piece_B(1, 1, 1));
      P8_COMPUTE(0, 0, steady);
      read_B(t, 1);
      __builtin_amdgcn_sched_barrier(0);
      issue_A(t + 1, piece_A(1, 1, 0)); issue_A(t + 1, piece_A(1, 1, 1));
      P8_COMPUTE(0, 1, steady);
      read_A(t, 1);
      __builtin_amdgcn_sched_barrier(0);
      issue_A(t + 2, piece_A(1, 0, 0)); issue_A(t + 2, piece_A(1, 0, 1));
      P8_COMPUTE(1, 1, steady);
      issue_B(t + 2, piece_B(0, 1, 0)); issue_B(t + 2, piece_B(0, 1, 1));
      P8_COMPUTE(1, 0, steady);
    }
    P8_BAR();
  } else {
    P8_BAR();
    for (int t = 0; t < nt; ++t) {
      const bool steady = t + 2 < nt;
      read_A(t, 0); read_B(t, 0);
      __builtin_amdgcn_sched_barrier(0);
      issue_A(t + 1, piece_A(0, 1, 0)); issue_A(t + 1, piece_A(0, 1, 1));
      P8_COMPUTE(0, 0, steady);
      read_B(t, 1);
      __builtin_amdgcn_sched_barrier(0);
      issue_A(t + 2, piece_A(0, 0, 0)); issue_A(t + 2, piece_A(0, 0, 1));
      P8_COMPUTE(0, 1, steady);
      read_A(t, 1);
      __builtin_amdgcn_sched_barrier(0);
      issue_B(t + 2, piece_B(0, 0, 0)); issue_B(t + 2, piece_B(0, 0, 1));
      P8_COMPUTE(1, 1, steady);
      issue_B(t + 2, piece_B(1, 0, 0)); issue_B(t + 2, piece_B(1, 0, 1));
      P8_COMPUTE(1, 0, steady);
    }
  }
#undef P8_COMPUTE
#undef P8_BAR
  epi_dispatch(p, [&](auto MD_) {
    constexpr int md = decltype(MD_)::value;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int m = m0 + wr * 128 + i * 16 + frow;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        epilogue4<md>(p, m, n0 + wc * 64 + j * 16 + fchk * 4, acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    }
  });
}

// ------------------------------------------------------------------------------------------
// TN (wgrad) GEMM
// ------------------------------------------------------------------------------------------
constexpr int TK = 64;  // rows (m) per LDS stage

// ---------------------------------------------------------------------------------------------------
// gemm_nt "skinny" kernel (M <= 64: KV-cache decode, one token per sequence).  HBM-bound: the job is to stream
// the [N,K] weight exactly once at full bandwidth with enough bytes in flight, not to feed the MFMA pipe.
//   * one workgroup (8 waves) = NR*16 weight rows x the WHOLE K, so no cross-workgroup reduction / atomics and the
//     full epilogue (alpha, bias, act, residual, fp32 out) stays fused; N = 4096 -> 256 workgroups = one per CU;
//   * inside the workgroup the K range is split over the 8 waves (contiguous slices, so every weight row is read as
//     8 sequential streams); each lane loads its MFMA fragment (row = lane&15, 8 bf16 at k-chunk lane>>4) straight
//     from global memory with 16-byte non-temporal loads, U k-steps issued back to back before the MFMAs;
//   * the weight fragment is the first MFMA operand, so a lane ends up with 4 consecutive output columns of one
//     activation row -> vector stores; activations (a few hundred KB, L2-resident) are the second operand, MF
//     fragments of 16 rows, rows >= M are zero and never loaded;
//   * the 8 partial accumulators meet in LDS (NR*MF KiB per wave), one barrier.
// The LoRA tail ([x|t].[W|B]^T) is a second K segment, split over the waves the same way.
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
// SW (fused SwiGLU, OPADPO_ACT_SWIGLU_PAIR): the weight rows come in groups of 128 = [64 gate | 64 up]; a workgroup takes 16
// gate rows and the 16 up rows 64 further (NR = 2), so after the reduction a lane holds gate and up of the same 4 output
// columns: act = silu(gate) * up on the bf16-rounded values (same formula as silu_mul_fwd_kernel), N/2 output columns.
template <int MF, int NR, int U, bool SW = false>
__global__ __launch_bounds__(512) void gemm_nt_skinny_kernel(GemmNTArgs p) {
  static_assert(!SW || NR == 2, "SwiGLU pairs: one gate + one up fragment per workgroup");
  __shared__ __attribute__((aligned(16))) float red[8][NR * MF][64][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = SW ? (blockIdx.x >> 2) * 128 + (blockIdx.x & 3) * 16 : blockIdx.x * (16 * NR);
  constexpr int RSTEP = SW ? 64 : 16;          // weight rows between the fragments of a workgroup
  const int r = lane & 15, c = lane >> 4;

  const bf16_t* a1 = p.A1;
  const bf16_t* a2 = p.A2;
  if (p.a1_group_n > 0) a1 += (size_t)(n0 / p.a1_group_n) * p.a1_group_stride;
  if (p.a2_group_n > 0) a2 += (size_t)(n0 / p.a2_group_n) * p.a2_group_stride;

  f32x4_t acc[NR][MF];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int f = 0; f < MF; ++f) acc[i][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto segment = [&](const bf16_t* A, int lda, const bf16_t* B, int ldb, int K) {
    const int ns = K >> 5;                       // 32-element k-steps
    const int per = (ns + 7) >> 3;
    const int sb = wave * per, se = min(ns, sb + per);
    const bf16_t* wp[NR];
    const bf16_t* xp[MF];
#pragma unroll
    for (int i = 0; i < NR; ++i) wp[i] = B + (size_t)(n0 + i * RSTEP + r) * ldb + c * 8;
#pragma unroll
    for (int f = 0; f < MF; ++f) xp[f] = A + (size_t)min(f * 16 + r, p.M - 1) * lda + c * 8;
    auto load = [&](u32x4_t (&w)[U][NR], u32x4_t (&x)[U][MF], int s) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool live = s + u < se;
        const int k = (s + u) << 5;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          w[u][i] = u32x4_t{0, 0, 0, 0};
          if (live) w[u][i] = __builtin_nontemporal_load((const u32x4_t*)(wp[i] + k));
        }
#pragma unroll
        for (int f = 0; f < MF; ++f) {
          x[u][f] = u32x4_t{0, 0, 0, 0};
          if (live && f * 16 + r < p.M) x[u][f] = *(const u32x4_t*)(xp[f] + k);
        }
      }
    };
    auto compute = [&](const u32x4_t (&w)[U][NR], const u32x4_t (&x)[U][MF]) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
          for (int f = 0; f < MF; ++f)
            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&w[u][i], *(const bf16x8_t*)&x[u][f], acc[i][f], 0, 0, 0);
    };
    for (int s = sb; s < se; s += U) {
      u32x4_t w[U][NR], x[U][MF];
      load(w, x, s);
      compute(w, x);
    }
  };
  segment(a1, p.lda1, p.B1, p.ldb1, p.K1);
  if (p.K2 > 0) segment(a2, p.lda2, p.B2, p.ldb2, p.K2);

#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int f = 0; f < MF; ++f) *(f32x4_t*)red[wave][i * MF + f][lane] = acc[i][f];
  __syncthreads();
  if constexpr (SW) {
    for (int f = wave; f < MF; f += 8) {
      f32x4_t g = *(const f32x4_t*)red[0][f][lane], u = *(const f32x4_t*)red[0][MF + f][lane];
#pragma unroll
      for (int w2 = 1; w2 < 8; ++w2) { g += *(const f32x4_t*)red[w2][f][lane]; u += *(const f32x4_t*)red[w2][MF + f][lane]; }
      const int m = f * 16 + r;
      if (m >= p.M) continue;
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gt = bf2f(f2bf(g[e])), up = bf2f(f2bf(u[e]));
        o[e] = gt / (1.0f + __expf(-gt)) * up;
      }
      uint2 st;
      st.x = pack_bf2(o[0], o[1]);
      st.y = pack_bf2(o[2], o[3]);
      *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + (blockIdx.x >> 2) * 64 + (blockIdx.x & 3) * 16 + c * 4) = st;
    }
    return;
  }
  for (int idx = wave; idx < NR * MF; idx += 8) {
    f32x4_t v = *(const f32x4_t*)red[0][idx][lane];
#pragma unroll
    for (int w2 = 1; w2 < 8; ++w2) v += *(const f32x4_t*)red[w2][idx][lane];
    const int i = idx / MF, f = idx % MF;
    const int m = f * 16 + r;
    if (m < p.M) epi_dispatch(p, [&](auto MD_) { epilogue4<decltype(MD_)::value>(p, m, n0 + i * 16 + c * 4, v[0], v[1], v[2], v[3]); });
  }
}


// M <= 16 form of the weight-streaming kernel (the shipped rollout runs 4 sequences per device): the 16x16x32 MFMA is used as TWO
// 8x8x32 products so that one wave load covers 8 weight rows x 128 CONTIGUOUS bytes (whole cache lines) instead of 16 rows x 64 B
// (measured at M = 8: q|k|v 3.9 -> 4.9 TB/s, gate|up 3.4 -> 4.2, down 3.5 -> 4.5, lm_head 3.7 -> 4.6).
// Operand rows 0..7 = weight rows with the k-chunks 0..3 of a 64-element k-step, rows 8..15 = the same weight rows with the
// chunks 4..7; the activation operand is split the same way (columns 0..7 = 8 tokens with chunks 0..3, 8..15 = the same tokens
// with chunks 4..7; MF8 operands of 8 tokens).  The diagonal 8x8 blocks of the 16x16 result are the two partial sums (lanes l
// and l + 40), the off-diagonal blocks are ignored - the matrix pipe is idle in this kernel anyway.  Everything else as in
// gemm_nt_skinny_kernel: K split over the 8 waves, partials meet in LDS, fused epilogue (SW: SwiGLU pairs, 8 gate + 8 up rows).
template <int MF8, int NR, int U, bool SW = false>
__global__ __launch_bounds__(512) void gemm_nt_skinny8_kernel(GemmNTArgs p) {
  static_assert(!SW || NR == 2, "SwiGLU pairs: one gate + one up fragment per workgroup");
  __shared__ __attribute__((aligned(16))) float red[8][NR * MF8][64][4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = SW ? (blockIdx.x >> 3) * 128 + (blockIdx.x & 7) * 8 : blockIdx.x * (8 * NR);
  constexpr int RSTEP = SW ? 64 : 8;
  const int r = lane & 15, c = lane >> 4;
  const int r8 = r & 7, kc = ((r >> 3) * 4 + c) * 8;        // row / token within the 8, first element of the lane's k-chunk

  const bf16_t* a1 = p.A1;
  const bf16_t* a2 = p.A2;
  if (p.a1_group_n > 0) a1 += (size_t)(n0 / p.a1_group_n) * p.a1_group_stride;
  if (p.a2_group_n > 0) a2 += (size_t)(n0 / p.a2_group_n) * p.a2_group_stride;

  f32x4_t acc[NR][MF8];
#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int f = 0; f < MF8; ++f) acc[i][f] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  auto segment = [&](const bf16_t* A, int lda, const bf16_t* B, int ldb, int K) {
    const int ns = K >> 6;                       // 64-element k-steps
    const int per = (ns + 7) >> 3;
    const int sb = wave * per, se = min(ns, sb + per);
    const bf16_t* wp[NR];
    const bf16_t* xp[MF8];
#pragma unroll
    for (int i = 0; i < NR; ++i) wp[i] = B + (size_t)(n0 + i * RSTEP + r8) * ldb + kc;
#pragma unroll
    for (int f = 0; f < MF8; ++f) xp[f] = A + (size_t)min(f * 8 + r8, p.M - 1) * lda + kc;
    for (int s = sb; s < se; s += U) {
      u32x4_t w[U][NR], x[U][MF8];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool live = s + u < se;
        const int k = (s + u) << 6;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          w[u][i] = u32x4_t{0, 0, 0, 0};
          if (live) w[u][i] = __builtin_nontemporal_load((const u32x4_t*)(wp[i] + k));
        }
#pragma unroll
        for (int f = 0; f < MF8; ++f) {
          x[u][f] = u32x4_t{0, 0, 0, 0};
          if (live && f * 8 + r8 < p.M) x[u][f] = *(const u32x4_t*)(xp[f] + k);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
          for (int f = 0; f < MF8; ++f)
            acc[i][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&w[u][i], *(const bf16x8_t*)&x[u][f], acc[i][f], 0, 0, 0);
    }
  };
  segment(a1, p.lda1, p.B1, p.ldb1, p.K1);
  if (p.K2 > 0) segment(a2, p.lda2, p.B2, p.ldb2, p.K2);

#pragma unroll
  for (int i = 0; i < NR; ++i)
#pragma unroll
    for (int f = 0; f < MF8; ++f) *(f32x4_t*)red[wave][i * MF8 + f][lane] = acc[i][f];
  __syncthreads();
  // lane (r < 8, c < 2): token f*8 + r, weight rows c*4 .. c*4+3 of the fragment = low-half partial; lane + 40 holds the high half
  if (c >= 2 || r >= 8) return;
  auto total = [&](int idx) {
    f32x4_t v = *(const f32x4_t*)red[0][idx][lane] + *(const f32x4_t*)red[0][idx][lane + 40];
#pragma unroll
    for (int w2 = 1; w2 < 8; ++w2) v += *(const f32x4_t*)red[w2][idx][lane] + *(const f32x4_t*)red[w2][idx][lane + 40];
    return v;
  };
  if constexpr (SW) {
    for (int f = wave; f < MF8; f += 8) {
      const int m = f * 8 + r;
      if (m >= p.M) continue;
      const f32x4_t g = total(f), u = total(MF8 + f);
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float gt = bf2f(f2bf(g[e])), up = bf2f(f2bf(u[e]));
        o[e] = gt / (1.0f + __expf(-gt)) * up;
      }
      uint2 st;
      st.x = pack_bf2(o[0], o[1]);
      st.y = pack_bf2(o[2], o[3]);
      *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + (blockIdx.x >> 3) * 64 + (blockIdx.x & 7) * 8 + c * 4) = st;
    }
  } else {
    for (int idx = wave; idx < NR * MF8; idx += 8) {
      const int i = idx / MF8, f = idx % MF8;
      const int m = f * 8 + r;
      if (m >= p.M) continue;
      const f32x4_t v = total(idx);
      epi_dispatch(p, [&](auto MD_) { epilogue4<decltype(MD_)::value>(p, m, n0 + i * 8 + c * 4, v[0], v[1], v[2], v[3]); });
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// gemm_nt "dec64" kernel: the decode GEMM for 33..64 tokens (rollout at 64 sequences per device, BASELINE.json configs[4]).
// The 16/32-row streaming kernels above keep one token fragment per MFMA in flight and re-read the whole activation K-range
// per 16 weight rows: at 64 tokens that is 4 B of L2 activation traffic per weight byte and 1-2 KB of weights in flight per
// wave - 1.4-2.2 TB/s.  Here a workgroup (4 waves) owns 64 weight rows x 64 tokens x ONE K-slice:
//   * BOTH operands go through LDS by direct-to-LDS DMA (buffer_load ... lds), whole 128-byte lines per row, into a ring of
//     16-KiB stages (8 KiB weights + 8 KiB activations per 64-deep k-tile): 4 stages = 48 KiB in flight per workgroup with two
//     workgroups per CU, or 8 stages = 112 KiB with one per CU when the grid has at most one workgroup per CU anyway (q|k|v:
//     192); no registers spent on staging; 1 byte of activations per weight byte;
//   * wave w multiplies its 16 weight rows with all MF token fragments (2 x MF MFMAs per k-tile; the matrix pipe idles, the
//     kernel's job is the HBM stream);
//   * K is split over gridDim.y workgroups where N alone gives too few (o / down projection: N = 4096 -> 64 column tiles x 4
//     K-slices); a slice writes its fp32 partial tile, the consumer (rmsnorm_sum_fwd: the residual add + RMSNorm that follows
//     every such projection) adds the slices - deterministic, no atomics, no extra launch.
// MODE 0: bf16 C[M,N]; 1: fp32 C[split][M,N] (ldc = N of one slice; logits when gridDim.y == 1); 2: SwiGLU pair - weight rows
// per 128 are [64 gate | 64 up] (OPADPO_ACT_SWIGLU_PAIR), a workgroup takes 32 gate rows and the 32 up rows 64 further:
// bf16 C[M, N/2] = silu(gate) * up on the bf16-rounded values (same formula as silu_mul_fwd_kernel).
// ---------------------------------------------------------------------------------------------------
constexpr int D_BN = 64, D_BK = 64, D_HALF = 8192, D_STAGE = 16384;

// KW = 2 (MODE 0 only): 32 weight rows per workgroup, waves 0,1 take the first 32 k of every tile and waves 2,3 the second (partials
// meet in LDS once, at the end) - twice the workgroups for a projection whose 64-row tiles do not fill the chip (q|k|v: 192 -> 384).
// OPADPO_DEC64_DIAG (compile-time, diagnostics only - results WRONG; tools/build_diag.sh): 1 no activation DMA, 2 no weight DMA, 4 no fragment
// reads / MFMAs, 8 no barrier in the k-loop
#ifndef OPADPO_DEC64_DIAG
#define OPADPO_DEC64_DIAG 0
#endif
template <int MF, int MODE, int D_NS, int KW = 1>      // D_NS = ring depth: 4 (two workgroups per CU) or 8 (one per CU)
__global__ __launch_bounds__(256, 2) void gemm_nt_dec64_kernel(GemmNTArgs p) {
  constexpr int DDG = OPADPO_DEC64_DIAG;
  static_assert(KW == 1 || MODE == 0, "K-split inside the workgroup: bf16 output only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bt = blockIdx.x, split = blockIdx.y, splits = gridDim.y;
  const int nt = p.K1 / D_BK;
  const int per = (nt + splits - 1) / splits;
  const int t0 = split * per, t1 = min(nt, t0 + per);
  // weight rows of this workgroup: 64 consecutive rows, or (MODE 2) 32 gate rows + the 32 up rows 64 further
  const int n0 = MODE == 2 ? (bt >> 1) * 128 + (bt & 1) * 32 : bt * (D_BN / KW);
  auto wrow = [&](int r) { return MODE == 2 ? n0 + (r < 32 ? r : 32 + r) : n0 + r; };      // r >= 32 -> n0 + 64 + (r - 32)
  // DMA pieces: piece q (0..7) of an operand = rows q*8 .. q*8+7 x 128 B; wave w issues pieces 2w, 2w+1 of both operands.
  // LDS image: row r at r*128, 16-byte chunk c stored at position c ^ ((r >> 1) & 7) (conflict-free ds_read_b128 fragments);
  // the DMA writes lane-linear, so the swizzle is applied on the SOURCE side: lane (row q*8 + (l >> 3), position l & 7)
  // fetches global chunk (l & 7) ^ ((row >> 1) & 7).
  const int prow = lane >> 3, ppos = lane & 7;
  constexpr int WP = 2 / KW;                     // weight pieces per wave and stage (KW = 2: wave w issues piece w = rows 8w..8w+7)
  unsigned voffW[2], voffA[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = (wave * 2 + j) * 8 + prow;
    const int ch = ppos ^ ((r >> 1) & 7);
    voffA[j] = (unsigned)min(r, p.M - 1) * (unsigned)p.lda1 * 2u + ch * 16u;
    const int rw = (wave * WP + j) * 8 + prow;
    voffW[j] = (unsigned)wrow(rw) * (unsigned)p.ldb1 * 2u + (ppos ^ ((rw >> 1) & 7)) * 16u;
  }
  auto uni = [](const void* q) -> void* {
    const unsigned long long v = (unsigned long long)q;
    return (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)v));
  };
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(uni(p.B1), 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(uni(p.A1), 0, (int)0xffffffffu, 0x00020000);
  auto issue = [&](int t) {
    char* st = smem + ((t - t0) % D_NS) * D_STAGE;
    const int k2 = t * D_BK * 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int piece = wave * 2 + j;
      if (j < WP && !(DDG & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, LDS_PTR(void, st + (wave * WP + j) * 1024), 16, voffW[j], k2, 0, 2);      // aux 2 = nt: streamed once
      if (!(DDG & 1)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(void, st + D_HALF + piece * 1024), 16, voffA[j], k2, 0, 0);
    }
  };
  f32x4_t acc[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fc = lane >> 4;
  const int wr = (KW == 2 ? (wave & 1) : wave) * 16 + fr;
  const int wsw = (wr >> 1) & 7, asw = (fr >> 1) & 7;            // token rows f*16 + fr: (row >> 1) & 7 == (fr >> 1) & 7
#pragma unroll
  for (int s = 0; s < D_NS - 1; ++s)
    if (t0 + s < t1) issue(t0 + s);
  for (int t = t0; t < t1; ++t) {
    // stage t landed: this wave's pieces of stages t+1 .. t+NS-2 may stay in flight (2 + WP pieces each)
    constexpr int PPS = ((DDG & 1) ? 0 : 2) + ((DDG & 2) ? 0 : WP);      // pieces per wave and stage
    if (t1 - t - 1 >= D_NS - 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D_NS - 2) * PPS) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!(DDG & 8)) __syncthreads();                  // everyone's pieces of stage t are in; everyone is done reading stage t-1
    if (t + D_NS - 1 < t1) issue(t + D_NS - 1);          // into the slot stage t-1 occupied
    const char* st = smem + ((t - t0) % D_NS) * D_STAGE;
    if (DDG & 4) continue;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (KW == 2 && kk != (wave >> 1)) continue;
      const bf16x8_t wf = *(const bf16x8_t*)(st + wr * 128 + (((kk * 4 + fc) ^ wsw) << 4));
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const bf16x8_t af = *(const bf16x8_t*)(st + D_HALF + (f * 16 + fr) * 128 + (((kk * 4 + fc) ^ asw) << 4));
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, af, acc[f], 0, 0, 0);
      }
    }
  }
  // lane holds C[token f*16 + fr][4 consecutive columns fc*4 .. +3 of the wave's 16 weight rows]
  if constexpr (MODE == 2) {
    __syncthreads();                                           // ring is dead: reuse it for the gate / up exchange
    float* ex = (float*)smem;                                  // [2 up waves][MF][64 lanes][4]
    if (wave >= 2) {
#pragma unroll
      for (int f = 0; f < MF; ++f) *(f32x4_t*)(ex + (((wave - 2) * MF + f) * 64 + lane) * 4) = acc[f];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const int m = f * 16 + fr;
        if (m >= p.M) continue;
        const f32x4_t u = *(const f32x4_t*)(ex + ((wave * MF + f) * 64 + lane) * 4);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gt = bf2f(f2bf(acc[f][e])), up = bf2f(f2bf(u[e]));
          o[e] = gt / (1.0f + __expf(-gt)) * up;
        }
        uint2 stv;
        stv.x = pack_bf2(o[0], o[1]);
        stv.y = pack_bf2(o[2], o[3]);
        *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + (bt >> 1) * 64 + (bt & 1) * 32 + wave * 16 + fc * 4) = stv;
      }
    }
    return;
  }
  if constexpr (KW == 2) {                                     // the two k-halves of a 16-row group meet in LDS
    __syncthreads();
    float* ex = (float*)smem;
    if (wave >= 2) {
#pragma unroll
      for (int f = 0; f < MF; ++f) *(f32x4_t*)(ex + (((wave - 2) * MF + f) * 64 + lane) * 4) = acc[f];
    }
    __syncthreads();
    if (wave >= 2) return;
#pragma unroll
    for (int f = 0; f < MF; ++f) acc[f] += *(const f32x4_t*)(ex + ((wave * MF + f) * 64 + lane) * 4);
  }
#pragma unroll
  for (int f = 0; f < MF; ++f) {
    const int m = f * 16 + fr;
    if (m >= p.M) continue;
    const int n = n0 + (KW == 2 ? (wave & 1) : wave) * 16 + fc * 4;
    if constexpr (MODE == 1) {
      *(float4*)((float*)p.C + ((size_t)split * p.M + m) * p.ldc + n) = make_float4(acc[f][0], acc[f][1], acc[f][2], acc[f][3]);
    } else {
      uint2 stv;
      stv.x = pack_bf2(acc[f][0], acc[f][1]);
      stv.y = pack_bf2(acc[f][2], acc[f][3]);
      *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = stv;
    }
  }
}


// [TK][128] bf16 tiles (256-byte rows) with the 16-byte-chunk swizzle chunk ^= (row & 7) << 1, which is
// conflict-free for ds_read_b64_tr_b16 (the 32 lanes of a half-wave touch 8 rows x 32 B = all 64 banks).
__device__ __forceinline__ const char* tn_at(const char* tile, int row, int col) {
  const int b = col * 2;
  return tile + row * 256 + ((((b >> 4) ^ ((row & 7) << 1))) << 4) + (b & 15);
}
template <bool TR>
__device__ __forceinline__ bf16x8_t tn_frag(const char* tile, int k0, int c0, int lane) {
  union { bf16x8_t v; s16x4_t h[2]; uint16_t s[8]; } u;
  const int g = lane >> 4, c = lane & 15;
  if constexpr (TR) {
    const int col = c0 + (c & 3) * 4;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, tn_at(tile, k0 + g * 8 + (c >> 2), col)));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, tn_at(tile, k0 + g * 8 + 4 + (c >> 2), col)));
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) u.s[j] = *(const uint16_t*)tn_at(tile, k0 + g * 8 + j, c0 + c);
  }
  return u.v;
}

template <bool TR>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTNArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * TK * 128 * 2];  // P tile, Q tile: [TK][128] bf16
  char* Ps = smem;
  char* Qs = smem + TK * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tiles_n2 = p.N2 / 128;
  const int t1 = blockIdx.x / tiles_n2, t2 = blockIdx.x % tiles_n2;
  const int n1_0 = t1 * 128, n2_0 = t2 * 128;
  const int chunk = (((p.M + gridDim.y - 1) / gridDim.y) + TK - 1) / TK * TK;
  const int m_begin = blockIdx.y * chunk;
  const int m_end = min(p.M, m_begin + chunk);
  if (m_begin >= m_end) return;

  const bf16_t* Q = p.Q;
  if (p.q_group_n1 > 0) Q += (size_t)(n1_0 / p.q_group_n1) * p.q_group_stride;

  const int wm = wave >> 1, wn = wave & 1;
  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  uint4 pv[4], qv[4];
  auto fetch = [&](int mb) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;          // 1024 x 16 B per tile
      const int row = idx >> 4, c16 = idx & 15;
      const int m = mb + row;
      pv[i] = make_uint4(0, 0, 0, 0);
      qv[i] = make_uint4(0, 0, 0, 0);
      if (m < m_end) {
        pv[i] = *(const uint4*)(p.P + (size_t)m * p.ldp + n1_0 + c16 * 8);
        qv[i] = *(const uint4*)(Q + (size_t)m * p.ldq + n2_0 + c16 * 8);
      }
    }
  };
  fetch(m_begin);
  for (int mb = m_begin; mb < m_end; mb += TK) {
    // commit the prefetched rows to LDS (zero rows past m_end), then prefetch the next stage
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      const int row = idx >> 4, c16 = idx & 15;
      const int off = row * 256 + ((c16 ^ ((row & 7) << 1)) << 4);
      *(uint4*)(Ps + off) = pv[i];
      *(uint4*)(Qs + off) = qv[i];
    }
    __syncthreads();
    if (mb + TK < m_end) fetch(mb + TK);
#pragma unroll
    for (int kk = 0; kk < TK / 32; ++kk) {
      bf16x8_t pf[4], qf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pf[i] = tn_frag<TR>(Ps, kk * 32, wm * 64 + i * 16, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) qf[j] = tn_frag<TR>(Qs, kk * 32, wn * 64 + j * 16, lane);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[i], qf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // D[i = n1][j = n2]: lane holds col n2 = lane&15, rows n1 = (lane>>4)*4 + reg
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n1 = n1_0 + wm * 64 + i * 16 + g * 4 + q;
        const int n2 = n2_0 + wn * 64 + j * 16 + c;
        atomicAdd(p.C + (size_t)n1 * p.ldc + n2, acc[i][j][q] * p.alpha);
      }
}


// ---- gemm_tn, wide tiles ------------------------------------------------------------------------------------------
// The LoRA wgrads are tall-skinny reductions over M: dB = dY^T t has N2 = r = 256, dA = dT^T x has N1 = r, 2r or 3r.  With the
// 128x128 tile the BIG operand (dY, resp. x) is fetched once per 128-wide slice of the small one — twice for dB, up to six
// times for dA(q|k|v) — and the kernel runs at HBM speed on re-reads (dB(gate|up): 2 x 1.43 GB in 0.71 ms).  Here the tile
// spans 256 of the small dimension (BN1 x BN2 = 128 x 256 for dB, 256 x 128 for dA; 4 waves of 64x128 / 128x64), rows past the
// block's M-slice come back as zeros from a per-block buffer descriptor (no predicates, scalar row offsets), and each
// transposed LDS fragment feeds 4 or 8 MFMAs instead of 4.
// MEASURED (M = 32362, one decoder layer's 8 wgrads): 2.60 ms vs 2.37 ms with the 128x128 kernel — the re-reads hit the L2 /
// Infinity Cache (the slices of one row block run concurrently), and at ~250 VGPRs the wide kernel hides less latency:
// dA(down) 0.32 vs 0.38 ms, but dB(*) 0.18-0.78 vs 0.13-0.71 ms.  Kept behind set_flags(use_tr bit 2), off by default.
}  // namespace

namespace {
// 512-byte rows, 16-byte chunks: chunk ^= ((row & 7) << 1) ^ (((row >> 3) & 1) << 3).  A ds_read_b64_tr_b16 half-wave reads rows
// r..r+3 and r+8..r+11 (two 16-lane transposition groups): the second term moves the second group to the other 32 banks.
__device__ __forceinline__ int tn3_mask(int row) { return ((row & 7) << 1) ^ (((row >> 3) & 1) << 3); }
__device__ __forceinline__ int tn3_off(int row, int col) {
  const int b = col * 2;
  return row * 512 + ((((b >> 4) ^ tn3_mask(row))) << 4) + (b & 15);
}

// ------------------------------------------------------------------------------------------
// gemm_tn "w4" kernel: C[N1,N2] (fp32, atomics) += alpha * P[M,N1]^T Q[M,N2] with the geometry of gemm_nt_w4_kernel: 256x256
// tile, FOUR waves x 128x128 (256 AGPR accumulators tied in place), K-step = 64 rows of M, two 64-KiB LDS stages filled by
// buffer_load ... lds.  The 128x128 kernel re-reads both operands once per 128-wide slice of the other (dB(gate|up): 5.7 GB
// through the L2 for 1.45 GB of operands); this tile halves that and feeds each fragment to 8 MFMAs.
//   LDS image: [64 m][256 cols] per operand (512-byte rows, chunk swizzle tn3_mask on the per-lane SOURCE address); MFMA
//     fragments (16 cols x 32 m) come out of it with two ds_read_b64_tr_b16 each (64 per K-step and wave).
//   Rows past M: the buffer descriptors end at M rows, out-of-range pieces land as zeros.
//   Work: linear order (K-chunk s of p.splits, tile, K-step inside the chunk) cut into gridDim.x equal contiguous runs (one block
//     per CU, one round); a run that crosses a tile boundary flushes its accumulators and starts the next tile.  Consecutive runs
//     are neighbouring tiles over the SAME rows of M, and the block -> run map hands each XCD (blocks are dealt round-robin over
//     the 8 XCDs) a contiguous range of runs.  Every partial result is an fp32 atomic add.
//   The LDS-DMA is inline asm: behind the builtin the compiler cannot tell the transposing LDS reads (an intrinsic without alias
//     information) from the DMA's LDS writes and put s_waitcnt vmcnt(0) in front of them - an HBM round trip per fragment group
//     (measured 6.9 k cycles per K-step instead of 3.3 k).  The waits for DMA data are the explicit ones of the schedule.
//   Schedule of one K-step (MFMA index m = 0..127).  The 2 x 32 transposing reads are spread over 40 and 48 MFMAs (32 and 28 in
//     gemm_nt_w4), the "K-step t+1 has landed" barrier sits at MFMA 80:
//       m 0..39  : 16 fragments of set 1 (k 32..63 of this step)          | lgkmcnt(0), barrier at 48: the stage is dead
//       m 48..78 : DMA pieces 0..7 of K-step t+2, one per 4 MFMAs          | vmcnt(8), barrier at 80: K-step t+1 has landed
//       m 80..127: 16 fragments of set 0 of K-step t+1 (one per 3 MFMAs), DMA pieces 8..15
//   Measured (M = 32362, one decoder layer's 8 LoRA wgrads): 1.63 ms vs 2.35 ms with the 128x128 kernel; 1.1 PF/s on a 4096^2
//   output.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_tn_w4_kernel(GemmTNGroup G) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  // a GROUP of problems sharing M (the 8 LoRA wgrads of a decoder layer): their 256x256 tiles form one list, so the runs of a
  // launch are long (whole layer: 305 tiles x 506 K-steps over 256 blocks) and a block flushes 2-3 tiles of atomics per ~600
  // K-steps instead of one per 32 (r-wide outputs launched alone: 16 tiles -> 16 K-chunks, the flush as long as the K-loop)
  const int tiles = G.tile_end[G.n - 1];
  const int ksteps = (G.g[0].M + 63) / 64;
  const int chunk_len = (ksteps + G.splits - 1) / G.splits;
  const long long total = (long long)G.splits * tiles * chunk_len;
  const long long per = (total + gridDim.x - 1) / gridDim.x;
  const int lrun = (gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
  long long run_s = (long long)lrun * per;
  const long long run_e = min(total, run_s + per);
  // per-lane pieces: piece = wave*8 + q holds rows 2*piece, 2*piece + 1 (lane >> 5) of the K-step, 32 chunks of 16 B each
  const int prow = lane >> 5, pchk = lane & 31;
  // fragment read offsets: [frag][half] for the P (n1) and Q (n2) side; row = g*8 + (c >> 2) (+4), col = c0 + (c & 3)*4
  const int fg = lane >> 4, fc = lane & 15;
  int offP[8][2], offQ[8][2];
#pragma unroll
  for (int f = 0; f < 8; ++f)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = fg * 8 + h * 4 + (fc >> 2);
      offP[f][h] = tn3_off(row, wr * 128 + f * 16 + (fc & 3) * 4);
      offQ[f][h] = 2 * P_TILE + tn3_off(row, wc * 128 + f * 16 + (fc & 3) * 4);
    }
  const int c = lane & 15, g = lane >> 4;
  const unsigned lds0 = (unsigned)(size_t)LDS_PTR(void, smem);

  int seg_no = -1;                     // ordinal of the segment inside this run (empty segments count: the reduce kernel enumerates them the same way)
  while (run_s < run_e) {
    ++seg_no;
    const long long cidx = run_s / chunk_len;
    const int kin = (int)(run_s % chunk_len), sidx = (int)(cidx / tiles), gtile = (int)(cidx % tiles);
    int pi = 0;
    while (pi + 1 < G.n && gtile >= G.tile_end[pi]) ++pi;
    pi = __builtin_amdgcn_readfirstlane(pi);
    const GemmTNArgs& p = G.g[pi];
    const int tile = gtile - (pi ? G.tile_end[pi - 1] : 0);
    const int tiles_n2 = p.N2 / 256;
    const int seg = (int)min((long long)(chunk_len - kin), run_e - run_s);      // K-steps of this segment inside the padded chunk
    run_s += seg;
    const int ks0 = sidx * chunk_len + kin;
    const int nt = min(seg, ksteps - ks0);                                        // ... that exist (the last chunk is shorter)
    if (nt <= 0) continue;
    const int n1_0 = __builtin_amdgcn_readfirstlane((tile / tiles_n2) * 256), n2_0 = __builtin_amdgcn_readfirstlane((tile % tiles_n2) * 256);
    const int qshift = p.q_group_n1 > 0 ? (n1_0 / p.q_group_n1) * p.q_group_stride : 0;
    typedef __attribute__((ext_vector_type(4))) int i32x4_t;
    auto mk_rsrc = [&](const void* base, unsigned nrec) {
      const unsigned long long v = (unsigned long long)base;
      i32x4_t r;
      r[0] = __builtin_amdgcn_readfirstlane((int)v); r[1] = __builtin_amdgcn_readfirstlane((int)((v >> 32) & 0xffffu));
      r[2] = __builtin_amdgcn_readfirstlane((int)nrec); r[3] = 0x00020000;
      return r;
    };
    const i32x4_t rP = mk_rsrc(p.P, (unsigned)p.M * (unsigned)p.ldp * 2u);
    const i32x4_t rQ = mk_rsrc(p.Q + qshift, (unsigned)p.M * (unsigned)p.ldq * 2u - (unsigned)qshift * 2u);
    unsigned voff[16];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = (wave * 8 + q) * 2 + prow;                         // row of the K-step
      const int sc = pchk ^ tn3_mask(r);                               // source chunk that lands at LDS chunk pchk
      voff[q] = (unsigned)r * (unsigned)p.ldp * 2u + (unsigned)(n1_0 + sc * 8) * 2u;
      voff[8 + q] = (unsigned)r * (unsigned)p.ldq * 2u + (unsigned)(n2_0 + sc * 8) * 2u;
    }
    const unsigned stepP = 64u * (unsigned)p.ldp * 2u, stepQ = 64u * (unsigned)p.ldq * 2u;
    // K-steps past the end of the segment are issued as EMPTY pieces (descriptor of zero records: no memory traffic, zeros land in a dead stage): the loop body
    // needs no "has a next / next-but-one K-step" variants - every K-step issues 16 pieces and waits with the same counts (round 6: with the two-K-steps-per-trip
    // loop the six body variants of the peeled form spilled)
    i32x4_t rPz = rP, rQz = rQ;
    rPz[2] = 0; rQz[2] = 0;
    auto issue_piece = [&](int t, int q) {      // K-step t of this run; q = 0..7: P pieces, 8..15: Q pieces
      const bool real = t < nt;
      const i32x4_t dP = real ? rP : rPz, dQ = real ? rQ : rQz;
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (t & 1) * P_TILE + (q < 8 ? 0 : 2 * P_TILE) + (wave * 8 + (q & 7)) * 1024));
      const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(ks0 + t) * (q < 8 ? stepP : stepQ)));
      if (q < 8) asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff[q]), "s"(dP), "s"(soff), "s"(dst) : "memory");
      else asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(voff[q]), "s"(dQ), "s"(soff), "s"(dst) : "memory");
    };

    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t fa[2][8], fb[2][8];
    // fragment r of set kk (m 32*kk..) of K-step t: r = 0..7 -> Q fragments, 8..15 -> P fragments.  The two 64-bit halves are
    // joined as a register sequence (a union would be assembled with moves right behind the loads).
    // LDS image (round 6): [P stage 0 | P stage 1 | Q stage 0 | Q stage 1], 32 KiB each - the stage and the k-half of a fragment read are COMPILE-TIME terms
    // (PAR = parity of the K-step, the loop below runs two K-steps per trip) that fit the 16-bit offset field of ds_read_b64_tr_b16 next to the lane's one base
    // register per (fragment, half).  Rounds 4-5 kept [P | Q] per stage 64 KiB apart and chose the stage at run time: one v_add_u32 in front of every one of the
    // 64 reads of a K-step - 0.75 VALU instructions per MFMA in a loop whose NT sibling issues 0.08 (profiles/r06p_pmc_tn.txt).
    auto read_frag = [&](auto PAR, int kk, int r) {
      const char* st = smem + decltype(PAR)::value * P_TILE + kk * (32 * 512);
      typedef __attribute__((ext_vector_type(8))) short s16x8_t;
      const int o0 = r < 8 ? offQ[r][0] : offP[r - 8][0], o1 = r < 8 ? offQ[r][1] : offP[r - 8][1];
      const s16x4_t h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, st + o0));
      const s16x4_t h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, st + o1));
      const bf16x8_t v = __builtin_bit_cast(bf16x8_t, (s16x8_t)__builtin_shufflevector(h0, h1, 0, 1, 2, 3, 4, 5, 6, 7));
      if (r < 8) fb[kk][r] = v;
      else fa[kk][r - 8] = v;
    };
#define W4_PIN() __builtin_amdgcn_sched_barrier(0)
    auto mfma_run = [&](int kk, int idx0, int n) {      // idx = i*8 + j
#pragma unroll
      for (int e = 0; e < n; ++e) {
        const int idx = idx0 + e, i = idx >> 3, j = (i & 1) ? 7 - (idx & 7) : (idx & 7);      // the second operand's fragments walked back and forth: the MFMA at a row change keeps it (as in the gemm_nt K-loop; same sums per accumulator)
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fa[kk][i]), "v"(fb[kk][j]));
      }
    };
#ifndef OPADPO_TN_SCHED
#define OPADPO_TN_SCHED 1      // 1 (round 6): early stage release + all 16 pieces at a period of 4 + late landing wait; 0: the schedule of rounds 4-5 (A/B builds)
#endif
#if OPADPO_TN_SCHED == 1
    // Schedule of one K-step, round 6 (MFMA index m = 0..127).  The grouped launch is bound by its two-stage ring against the loaded HBM latency (a K-step cannot be
    // requested before its stage is free: profiles/r06q_tn_static_offsets.txt), so what counts is the window between a piece's issue and the landing wait:
    //   m 0..30  : the 16 fragments of set 1 (rows 32..63 of this K-step), one per 2 MFMAs            | lgkmcnt(0) + barrier at 32: the stage is dead
    //   m 34..94 : ALL 16 DMA pieces of K-step t+2, one per 4 MFMAs
    //   m 99     : vmcnt(16) + barrier: K-step t+1 has landed for everyone (the 16 pieces above stay in flight)
    //   m 100..127: the 16 fragments of set 0 of K-step t+1
    // window of a piece: 134..194 MFMA slots (rounds 4-5: stage released at 48, pieces at 50..125, landing wait at 79: 82..157 slots).
    auto tile_body = [&](int t, auto PAR) {
      using NPAR = std::integral_constant<int, 1 - decltype(PAR)::value>;
      // (small fully unrolled loops: one 128-trip loop over m is not unrolled by hipcc, and fa / fb indexed by a run-time m go to scratch)
#pragma unroll
      for (int g2 = 0; g2 < 16; ++g2) {             // m 0..31
        mfma_run(0, 2 * g2, 1);
        W4_PIN(); read_frag(PAR, 1, g2); W4_PIN();
        mfma_run(0, 2 * g2 + 1, 1);
      }
      mfma_run(0, 32, 1);
      W4_PIN();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // the stage of K-step t is dead: it takes K-step t+2
      W4_PIN();
      mfma_run(0, 33, 1);
#pragma unroll
      for (int q = 0; q < 16; ++q) {                // m 34..97: piece q behind MFMA 34 + 4 q
        const int m = 34 + 4 * q;
        mfma_run(m >> 6, m & 63, 1);
        W4_PIN(); issue_piece(t + 2, q); W4_PIN();
#pragma unroll
        for (int e = 1; e < 4; ++e) mfma_run((m + e) >> 6, (m + e) & 63, 1);
      }
      mfma_run(1, 34, 2);                           // m 98, 99
      W4_PIN();
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      __builtin_amdgcn_s_barrier();                // K-step t+1 has landed for everyone
      W4_PIN();
#pragma unroll
      for (int k = 0; k < 16; ++k) {                // m 100..127: fragment k behind MFMA 100 + (7 k) / 4  (100, 101, 103, 105, 107, 108, ...)
        const int m0 = 100 + (7 * k) / 4, m1 = k < 15 ? 100 + (7 * (k + 1)) / 4 : 128;
        mfma_run(1, m0 - 64, 1);
        W4_PIN(); read_frag(NPAR{}, 0, k); W4_PIN();
#pragma unroll
        for (int m = m0 + 1; m < m1; ++m) mfma_run(1, m - 64, 1);
      }
      W4_PIN();
    };
#else
    auto tile_body = [&](int t, auto PAR) {
      using NPAR = std::integral_constant<int, 1 - decltype(PAR)::value>;
      // m 0..39: the 16 fragments of set 1, one per 2 / 3 MFMAs
#pragma unroll
      for (int g2 = 0; g2 < 16; ++g2) {
        mfma_run(0, (g2 * 5) / 2, ((g2 + 1) * 5) / 2 - (g2 * 5) / 2);
        W4_PIN();
        read_frag(PAR, 1, g2);
        W4_PIN();
      }
      mfma_run(0, 40, 7);
      W4_PIN();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      W4_PIN(); mfma_run(0, 47, 1); W4_PIN();
      __builtin_amdgcn_s_barrier();            // the stage of K-step t is dead: it takes K-step t+2
      W4_PIN();
      // m 48..78: DMA pieces 0..7 of K-step t+2, one per 4 MFMAs (after m = 50, 54, ... 78)
#pragma unroll
      for (int m = 48; m < 79; ++m) {
        mfma_run(m >> 6, m & 63, 1);
        if ((m - 48) % 4 == 2) {
          W4_PIN();
          issue_piece(t + 2, (m - 48) / 4);
          W4_PIN();
        }
      }
      W4_PIN();
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // the 8 pieces above stay in flight
      W4_PIN(); mfma_run(1, 15, 1); W4_PIN();
      __builtin_amdgcn_s_barrier();            // K-step t+1 has landed for everyone
      W4_PIN();
      // m 80..127: the 16 fragments of set 0 of K-step t+1 (one per 3 MFMAs; stale bytes after the last K-step: never used), DMA pieces 8..15 (one per 6)
#pragma unroll
      for (int g2 = 0; g2 < 16; ++g2) {
        mfma_run(1, 16 + g2 * 3, 3);
        W4_PIN();
        read_frag(NPAR{}, 0, g2);
        if (g2 & 1) { W4_PIN(); issue_piece(t + 2, 8 + (g2 >> 1)); }
        W4_PIN();
      }
    };

#endif
#pragma unroll
    for (int q = 0; q < 16; ++q) issue_piece(0, q);
#pragma unroll
    for (int q = 0; q < 16; ++q) issue_piece(1, q);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    W4_PIN();
    using P0_ = std::integral_constant<int, 0>; using P1_ = std::integral_constant<int, 1>;
#pragma unroll
    for (int r = 0; r < 16; ++r) read_frag(P0_{}, 0, r);
    W4_PIN();
    for (int t = 0; t < nt; t += 2) {              // even K-steps live in stage 0, odd ones in stage 1
      tile_body(t, P0_{});
      if (t + 1 < nt) tile_body(t + 1, P1_{});
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // the empty pieces behind the last K-step
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    W4_PIN();
#undef W4_PIN
    // D[i = n1][j = n2]: lane holds col n2 = lane & 15, rows n1 = (lane >> 4)*4 + reg
    if (G.ws) {
      // deterministic flush: the raw partial tile of this (run, segment) as 64 one-KiB stores per wave, [wave][i][j][lane] float4 -
      // gemm_tn_reduce_kernel adds a tile's segments in a fixed order (no atomics: LoRA gradients bit-reproducible run to run)
      float4* slot = (float4*)(G.ws + ((size_t)lrun * G.smax + seg_no) * 65536) + (size_t)wave * 4096 + lane;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v[4];
          acc_read4(acc[i][j], v);
          slot[(i * 8 + j) * 64] = make_float4(v[0], v[1], v[2], v[3]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v[4];
          acc_read4(acc[i][j], v);
#pragma unroll
          for (int q = 0; q < 4; ++q)
            atomicAdd(p.C + (size_t)(n1_0 + wr * 128 + i * 16 + g * 4 + q) * p.ldc + n2_0 + wc * 128 + j * 16 + c, v[q] * p.alpha);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_s_barrier();      // the next run segment re-uses the stages
  }
}
// Deterministic flush of gemm_tn_w4_kernel, part 2: C[tile] += alpha * (sum of the tile's segments, K-chunk by K-chunk and run by run in
// ascending order).  The segment enumeration is recomputed from the launch geometry exactly as the kernel walks it (run = `per` consecutive
// K-steps of the (K-chunk, tile, K-step) order; a run's segment s is the part of unit unit0 + s it covers).  Grid (tiles, 4): a block adds 4096
// float4 (one wave's 128x128 quadrant) of one tile; every element of C is owned by one thread (plain read-modify-write, no atomics).
__global__ __launch_bounds__(256) void gemm_tn_reduce_kernel(GemmTNGroup G, int n_runs) {
  const int tiles = G.tile_end[G.n - 1];
  const int ksteps = (G.g[0].M + 63) / 64;
  const int chunk_len = (ksteps + G.splits - 1) / G.splits;
  const long long total = (long long)G.splits * tiles * chunk_len;
  const long long per = (total + n_runs - 1) / n_runs;
  const int gtile = blockIdx.x;
  int pi = 0;
  while (pi + 1 < G.n && gtile >= G.tile_end[pi]) ++pi;
  const GemmTNArgs& p = G.g[pi];
  const int tile = gtile - (pi ? G.tile_end[pi - 1] : 0);
  const int tiles_n2 = p.N2 / 256;
  const int n1_0 = (tile / tiles_n2) * 256, n2_0 = (tile % tiles_n2) * 256;
  float4 sum[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) sum[e] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int sidx = 0; sidx < G.splits; ++sidx) {
    const long long u = (long long)sidx * tiles + gtile, lo = u * chunk_len, hi = lo + chunk_len;
    for (long long r = lo / per; r <= (hi - 1) / per && r < n_runs; ++r) {
      const long long start = max(lo, r * per), end = min(min(hi, (r + 1) * per), total);
      const int kin = (int)(start - lo), seg = (int)(end - start);
      const int nt = min(seg, ksteps - (sidx * chunk_len + kin));
      if (nt <= 0) continue;                                      // the padded tail of the last K-chunk: the kernel wrote nothing
      const int seg_no = (int)(u - (r * per) / chunk_len);
      const float4* slot = (const float4*)(G.ws + ((size_t)r * G.smax + seg_no) * 65536) + (size_t)blockIdx.y * 4096 + threadIdx.x;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float4 v = slot[e * 256];
        sum[e].x += v.x; sum[e].y += v.y; sum[e].z += v.z; sum[e].w += v.w;
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int idx = blockIdx.y * 4096 + e * 256 + threadIdx.x;      // ((wave*8 + i)*8 + j)*64 + lane
    const int lane = idx & 63, j = (idx >> 6) & 7, i = (idx >> 9) & 7, wave = idx >> 12;
    const int row = n1_0 + (wave >> 1) * 128 + i * 16 + (lane >> 4) * 4, col = n2_0 + (wave & 1) * 128 + j * 16 + (lane & 15);
    float* cp = p.C + (size_t)row * p.ldc + col;
    cp[0] += sum[e].x * p.alpha;
    cp[(size_t)p.ldc] += sum[e].y * p.alpha;
    cp[2 * (size_t)p.ldc] += sum[e].z * p.alpha;
    cp[3 * (size_t)p.ldc] += sum[e].w * p.alpha;
  }
}
}  // namespace

static bool g_skinny8 = true;       // M <= 16 decode GEMMs: whole-cache-line form of the streaming kernel
static int g_dec64x_nw = 0;
static int g_dec64_variant = getenv("OPADPO_DEC64_V") ? atoi(getenv("OPADPO_DEC64_V")) : 0;      // 0 / 3 the whole-line streaming kernel (dec64x), 1 the LDS-ring kernel; use_tr bits 5-6
static int g_gemm_variant = 10;   // 10 (default): auto; 4: 128x128 kernel; 17: 8-wave 256x256 kernel (p8); 31: 4-wave 256x256 kernel (w4) forced; 15: M <= 64 streaming
static bool g_use_tr = true;
static int g_tn_w4 = 1;        // use_tr bit 3 CLEARS it: 256x256 gemm_tn_w4_kernel (default) vs the 128x128 kernel
static bool g_w4s_few = false;      // opadpo_set_flags use_tr bit 10 (tests): 8 workgroups walk the tile list, so small problems exercise long walks
static bool g_w4_chunk_deal = false; // opadpo_set_flags use_tr bit 12 (tests, A/B): every XCD walks a contiguous chunk of the tile order (rounds 1-5) instead of the block-cyclic deal
static bool g_w4_nodeep = false;    // opadpo_set_flags use_tr bit 11 (tests, A/B): the products of >= 128 K-tiles keep the default K-loop text (round 6: they run the DEEP text)
void opadpo_set_flags_impl(int use_glds, int use_tr) {
  g_gemm_variant = use_glds;
  g_use_tr = (use_tr & 1) != 0;
  opadpo_set_attn_dma((use_tr & 2) != 0);
  g_tn_w4 = (use_tr & 8) == 0;
  g_skinny8 = (use_tr & 16) == 0;
  if ((use_tr >> 5) & 3) g_dec64_variant = (use_tr >> 5) & 3; else if (!getenv("OPADPO_DEC64_V")) g_dec64_variant = 0;
  opadpo_set_sample_compact((use_tr & 512) ? 0 : -1);      // bit 9 forces the diagnostic sampler; otherwise OPADPO_SAMPLE_COMPACT decides
  g_w4s_few = (use_tr & 1024) != 0;
  g_w4_nodeep = (use_tr & 2048) != 0;
  g_w4_chunk_deal = (use_tr & 4096) != 0;
  g_dec64x_nw = (use_tr >> 7) & 3;      // bits 7-8: rows per workgroup of the dec64x kernel (0 = by shape, 1 / 2 / 3 = 48 / 64 / 128; tests)
}
bool opadpo_flag_tr() { return g_use_tr; }

// piece order of the 4-wave 256x256 kernel by shape (see gemm_nt_w4_kernel): B half first for N <= 16 column tiles
static int g_w4_order = -1;      // OPADPO_W4_ORDER=0 / 1 forces A first / B first (experiments)
// Streaming form (gemm_nt_w4s_kernel, round 5) for the plain products whenever every workgroup gets at least two tiles: one workgroup per CU walks
// the tile list and keeps its K-tile pipeline full across output tiles.  OPADPO_W4S=0 / variant 31: one tile per workgroup (A/B, cross-check).
static int g_w4s = -1, g_w4s_cus = 0, g_w4s_maxnt = 128;
static const int env_rope_direct_ = getenv("OPADPO_ROPE_DIRECT") ? atoi(getenv("OPADPO_ROPE_DIRECT")) : 1;      // 0: table-free rotary embedding through the staged epilogue (rounds 3-4; A/B)
// piece order of the 256x256 kernels: the B pieces of a K-tile first for <= 16 column tiles (o / down / the dgrads) and - under the block-cyclic deal, where the A panels
// of a round are shared chip-wide - for the shallow products of up to 64 column tiles as well (q|k|v: +0.8-1.8 % at 7B, +0.9 % at 13B; the widest shapes lose 5-7 % with it:
// profiles/r06zz_ab_order_cyclic.txt, r06zz_ab_order_other_shapes.txt); otherwise the A pieces first.  A placement choice: the results do not depend on it.
static bool w4_b_first(const GemmNTArgs& a) {
  const int tiles_n = a.N / P_BN, nt = (a.K1 + a.K2) / P_BK;
  return tiles_n <= 16 || (tiles_n <= 64 && nt < 128);
}
static const int env_w4_deep_ = getenv("OPADPO_W4_DEEP") ? atoi(getenv("OPADPO_W4_DEEP")) : 2;      // 2 (default): >= 128 K-tiles run the DEEP text, streaming where eligible; 1: DEEP text, one tile per workgroup; 0: default text (A/B)
#define W4_LAUNCH(GRID_)                                                                                                     \
  do {                                                                                                                       \
    const bool ob_ = g_w4_order >= 0 ? g_w4_order != 0 : w4_b_first(a);                                                       \
    const int grid_ = (GRID_), cus_ = g_w4s_few ? 8 : g_w4s_cus;                                                             \
    const bool stream_ = g_w4s > 0 && g_gemm_variant != 31 && !a.bias && !a.rope_cos && !a.rope_pos && a.b1_fold_n == 0 &&                               \
                         ((a.act == 0 && (!a.R || w4_direct_resid_ok(a))) || ((w4_direct_swiglu_bwd_ok(a) || w4_direct_swiglu_pair_ok(a)) && !a.swiglu_bwd_staged)) && \
                         a.K1 / P_BK >= 3 && (a.K1 + a.K2) / P_BK <= g_w4s_maxnt && grid_ >= 2 * cus_;                              \
    const bool deep_stream_ = env_w4_deep_ >= 2 && !g_w4_nodeep && g_w4s > 0 && g_gemm_variant != 31 && !a.bias && !a.rope_cos && !a.rope_pos && a.b1_fold_n == 0 && \
                              a.act == 0 && (!a.R || w4_direct_resid_ok(a)) && a.K1 / P_BK >= 3 && (a.K1 + a.K2) / P_BK >= 128 && (a.K1 + a.K2) / P_BK > g_w4s_maxnt && \
                              grid_ >= 2 * cus_;                                                                                \
    if (g_w4s > 0 && g_gemm_variant != 31 && env_rope_direct_ && w4_direct_rope_pos_ok(a) && a.K1 / P_BK >= 3 && (a.K1 + a.K2) / P_BK <= g_w4s_maxnt && grid_ >= 2 * cus_) { \
      if (ob_) hipLaunchKernelGGL((gemm_nt_w4s_kernel<true, 3>), dim3(cus_), dim3(256), 2 * P_STAGE, st, a, grid_);           \
      else hipLaunchKernelGGL((gemm_nt_w4s_kernel<false, 3>), dim3(cus_), dim3(256), 2 * P_STAGE, st, a, grid_);              \
    } else if (stream_ && a.act == OPADPO_ACT_SWIGLU_PAIR) {                                                                 \
      if (ob_) hipLaunchKernelGGL((gemm_nt_w4s_kernel<true, 2>), dim3(cus_), dim3(256), 2 * P_STAGE, st, a, grid_);           \
      else hipLaunchKernelGGL((gemm_nt_w4s_kernel<false, 2>), dim3(cus_), dim3(256), 2 * P_STAGE, st, a, grid_);              \
    } else if (stream_) {                                                                                                    \
      if (ob_) hipLaunchKernelGGL(gemm_nt_w4s_kernel<true>, dim3(cus_), dim3(256), 2 * P_STAGE, st, a, grid_);                \
      else hipLaunchKernelGGL(gemm_nt_w4s_kernel<false>, dim3(cus_), dim3(256), 2 * P_STAGE, st, a, grid_);                   \
    } else if (deep_stream_) {      /* >= 128 K-tiles: streaming pays with the DEEP text only (default text: -4.7 %) */      \
      if (ob_) hipLaunchKernelGGL((gemm_nt_w4s_kernel<true, 0, true>), dim3(cus_), dim3(256), 2 * P_STAGE, st, a, grid_);     \
      else hipLaunchKernelGGL((gemm_nt_w4s_kernel<false, 0, true>), dim3(cus_), dim3(256), 2 * P_STAGE, st, a, grid_);        \
    }                                                                                                                        \
    else if (env_w4_deep_ && !g_w4_nodeep && (a.K1 + a.K2) / P_BK >= 128) {      /* the deep-K products (down, the dgrads): the DEEP text */         \
      if (ob_) hipLaunchKernelGGL((gemm_nt_w4_kernel<true, false, true>), dim3(grid_), dim3(256), 2 * P_STAGE, st, a);       \
      else hipLaunchKernelGGL((gemm_nt_w4_kernel<false, false, true>), dim3(grid_), dim3(256), 2 * P_STAGE, st, a);          \
    }                                                                                                                        \
    else if (ob_) hipLaunchKernelGGL(gemm_nt_w4_kernel<true>, dim3(grid_), dim3(256), 2 * P_STAGE, st, a);                   \
    else hipLaunchKernelGGL(gemm_nt_w4_kernel<false>, dim3(grid_), dim3(256), 2 * P_STAGE, st, a);                            \
  } while (0)

// the bias / activation instantiations (w4_direct_ba_ok problems: the vision tower and the projector): one tile per workgroup only - the streaming
// kernel has no registers left for the bias / activation epilogue (it spills, and a kernel whose K-loop is an asm block must not touch scratch)
#define W4_LAUNCH_BA(GRID_)                                                                                                  \
  do {                                                                                                                       \
    const bool ob_ = g_w4_order >= 0 ? g_w4_order != 0 : a.N / P_BN <= 16;                                                    \
    if (ob_) hipLaunchKernelGGL((gemm_nt_w4_kernel<true, true>), dim3(GRID_), dim3(256), 2 * P_STAGE, st, a);                \
    else hipLaunchKernelGGL((gemm_nt_w4_kernel<false, true>), dim3(GRID_), dim3(256), 2 * P_STAGE, st, a);                   \
  } while (0)

// ---- K-folded r-wide products (round 6) ---------------------------------------------------------------------------------------------
// The N = lora_r = 256 products of the LoRA path (t = s x A^T for the o / down projections, dT = s dY B for their dgrads: four per layer and pass) are ONE
// column of 256x256 tiles: 96 workgroups on 256 CUs at the bench row count (the 128x128 kernel ran them at 0.4-0.55 PF/s, the quarter-tile form of the
// K = 11008 one at 0.55: profiles/r05zz_step_by_grid.txt - 17 ms of the step at 3 x their HBM floor).  Here the K range is cut in TWO halves that run as
// two column groups of one launch of the 256x256 kernel (192 workgroups, fp32 partial products side by side in a workspace; GemmNTArgs::b1_fold_n), and a
// small pass adds the halves in slice order, scales and rounds: C = bf16(alpha (P0 + P1)).  The rule depends on N and K only, never on M, and every row is
// summed the same way whatever the batch around it (P3: bit-equal), but it IS a different fp32 association than one pass over K (half sums, then their sum).
namespace {
__global__ __launch_bounds__(256) void gemm_fold2_reduce_kernel(const float* __restrict__ ws, bf16_t* __restrict__ C, int ldc, int M, int N, float alpha) {
  const int per_row = N >> 3;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)M * per_row) return;
  const int m = (int)(idx / per_row), c = (int)(idx % per_row) << 3;
  const float* p0 = ws + (size_t)m * 2 * N + c;
  const float4 a0 = *(const float4*)p0, a1 = *(const float4*)(p0 + 4), b0 = *(const float4*)(p0 + N), b1 = *(const float4*)(p0 + N + 4);
  const float v[8] = {(a0.x + b0.x) * alpha, (a0.y + b0.y) * alpha, (a0.z + b0.z) * alpha, (a0.w + b0.w) * alpha,
                      (a1.x + b1.x) * alpha, (a1.y + b1.y) * alpha, (a1.z + b1.z) * alpha, (a1.w + b1.w) * alpha};
  *(uint4*)(C + (size_t)m * ldc + c) = pack8(v);
}
}  // namespace
static float* g_fold_ws = nullptr;
static size_t g_fold_ws_bytes = 0;      // one buffer per process = per GPU, used in stream order (grown on demand: a device-synchronising hipFree + hipMalloc, first steps only)
static bool gemm_nt_kfold_ok(const GemmNTArgs& a, int variant) {
  static const int env_fold = getenv("OPADPO_KFOLD") ? atoi(getenv("OPADPO_KFOLD")) : 1;      // 0: the r-wide products in one pass over K (rounds 1-5; A/B)
  return env_fold && (variant == 10 || variant == 31) && !(a.act & OPADPO_GEMM_STREAM) && (a.act & 0xff) == 0 && a.N == 256 && a.K2 == 0 && !a.bias && !a.R &&
         !a.out_f32 && !a.rope_cos && !a.rope_pos && a.a1_group_n <= 0 && a.b1_fold_n == 0 && !a.quarter && a.K1 % 128 == 0 && a.K1 >= 1024 && a.ldc % 8 == 0 &&
         a.lda1 % 8 == 0 && a.ldb1 % 8 == 0 && (double)a.M * a.lda1 * 2 < 4.0e9 && (double)a.N * a.ldb1 * 2 < 4.0e9 && (double)(a.M + 256) * 512 * 4 < 4.0e9;
}

hipError_t launch_gemm_nt(const GemmNTArgs& a_in, hipStream_t st) {
  if (a_in.M <= 0) return hipSuccess;
  if (gemm_nt_kfold_ok(a_in, a_in.variant >= 0 ? a_in.variant : ::g_gemm_variant)) {
    const size_t need = (size_t)a_in.M * 512 * sizeof(float);
    if (need > g_fold_ws_bytes) {
      if (g_fold_ws) (void)hipFree(g_fold_ws);
      g_fold_ws = nullptr; g_fold_ws_bytes = 0;
      const size_t want = need + need / 4;
      if (hipMalloc(&g_fold_ws, want) != hipSuccess) { (void)hipGetLastError(); return hipErrorOutOfMemory; }
      g_fold_ws_bytes = want;
    }
    GemmNTArgs f = a_in;
    f.N = 512; f.K1 = a_in.K1 / 2; f.a1_group_n = 256; f.a1_group_stride = a_in.K1 / 2; f.b1_fold_n = 256; f.b1_fold_koff = a_in.K1 / 2;
    f.C = g_fold_ws; f.ldc = 512; f.out_f32 = 1; f.alpha = 1.0f;
    hipError_t e = launch_gemm_nt(f, st);
    if (e != hipSuccess) return e;
    const size_t n = (size_t)a_in.M * (256 / 8);
    hipLaunchKernelGGL(gemm_fold2_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g_fold_ws, (bf16_t*)a_in.C, a_in.ldc, a_in.M, 256, a_in.alpha);
    return hipGetLastError();
  }
  GemmNTArgs a = a_in;
  if (g_w4_order == -1) { const char* v = getenv("OPADPO_W4_ORDER"); g_w4_order = v ? atoi(v) : -2; }
  // C leaves the 4-wave 256x256 kernel's direct epilogue as NON-TEMPORAL stores (round 4): at the end of a round all 256 workgroups write
  // their 128-256 KiB of results at once - a 32-64 MB burst that drains at the fabric's write rate (the no-store diagnostic of the
  // persistent-loop experiment put it at 4-5 us of a ~100 us tile) while nothing computes; streaming stores do not fight the operand
  // panels for the 4-MiB L2s.  Same-box sustained A/B at M = 24576 (tools/ab_env.sh, profiles/r04_ab_nt_stores.txt): q|k|v +0.9 %, o
  // (fp32 out) +2.5 %, gate|up +0.6 %, down (fp32 out) +1.0 %, N = 768 +1.2 %.  OPADPO_W4_NT=0 keeps write-back stores (A/B).
  static const int env_nt = getenv("OPADPO_W4_NT") ? atoi(getenv("OPADPO_W4_NT")) : 1;
  a.store_nt = env_nt;
  // grouped tile order of the 256x256 4-wave kernel: 8 row tiles per group; 4 when the problem is at most 16 column tiles wide
  // (N <= 4096: o / down and three of the four dgrads) - measured at M = 32362: down 1.363 -> 1.397 PF/s, o 1.394 -> 1.401,
  // the wide projections lose 0.4-2 % with 4 or 6 and 6 % with 12.  OPADPO_W4_GM overrides (diagnostics).
  static const int env_gm = getenv("OPADPO_W4_GM") ? atoi(getenv("OPADPO_W4_GM")) : 0;
  a.group_m = env_gm > 0 ? env_gm : (a.N / P_BN <= 16 ? 4 : 8);
  // 32-tile blocks dealt to the XCDs block-cyclically (round 6; common.h xcd_remap_cyclic): the eight XCDs of a round share one row group's A panels through the
  // Infinity Cache and B stays within its reach between passes - down +2.0-5.3 %, gate|up +2.2-2.5 %, dgrad_gu +0.6-2.5 %, q|k|v +0.8-1.3 %, o +-0.2 %, step -1.0 %
  // (profiles/r06z_ab_cyclic.txt).  OPADPO_XCD_CYCLIC=0 / opadpo_set_flags use_tr bit 12: the contiguous-chunk deal of rounds 1-5 (same results, bit for bit).
  static const int env_cyc = getenv("OPADPO_XCD_CYCLIC") ? atoi(getenv("OPADPO_XCD_CYCLIC")) : 1;
  a.xcd_cyclic = env_cyc && !g_w4_chunk_deal;
  const bool stream_hint = (a.act & OPADPO_GEMM_STREAM) != 0;
  a.act &= 0xff;
  const int g_gemm_variant = a.variant >= 0 ? a.variant : ::g_gemm_variant;      // per-call override (opadpo_ctx_set_flags)
  if (a.N % BN || a.K1 % BK || a.K2 % BK || a.K1 + a.K2 <= 0) return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)gemm_nt_kernel_x<64, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4s_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4s_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4s_kernel<false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4s_kernel<true, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4s_kernel<false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4s_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4s_kernel<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4s_kernel<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    (void)hipFuncSetAttribute((const void*)gemm_nt_w4_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    {
      int dev_ = 0; hipDeviceProp_t pr_;
      g_w4s_cus = (hipGetDevice(&dev_) == hipSuccess && hipGetDeviceProperties(&pr_, dev_) == hipSuccess) ? pr_.multiProcessorCount : 256;
      g_w4s = getenv("OPADPO_W4S") ? atoi(getenv("OPADPO_W4S")) : 1;
      if (getenv("OPADPO_W4S_MAXNT")) g_w4s_maxnt = atoi(getenv("OPADPO_W4S_MAXNT"));
    }
    (void)hipFuncSetAttribute((const void*)gemm_nt_tail64_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    (void)hipFuncSetAttribute((const void*)gemm_nt_tail64_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
    (void)hipFuncSetAttribute((const void*)gemm_nt_p8_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    attr_set = true;
  }
  if (a.b1_fold_n > 0) {      // K-folded problem (built above): every tile on gemm_nt_w4_kernel, the only kernel that reads the folded B1
    if (a.N % P_BN || a.K1 % P_BK || a.K2) return hipErrorInvalidValue;
    const int tiles = ((a.M + P_BM - 1) / P_BM) * (a.N / P_BN);
    W4_LAUNCH(tiles);
    return hipGetLastError();
  }
  // Decode-sized weight streams with nothing fused (no LoRA tail, bias, residual or scale; plain or SwiGLU-pair epilogue) go to the whole-line
  // streaming kernel behind opadpo_gemm_nt_decode from 8 token rows up: the rollout's q|k|v, gate|up and lm_head at 8 sequences (decode step
  // 3.79 -> 3.71 ms) and the lm_head at 9-64 (B=16: 4.42 -> 4.30 ms).  Below 8 rows the 8-row kernels are ahead in the step (B=5: 3.73 vs 3.86 ms,
  // B=4: 3.41 vs 3.52; profiles/r04_decode_stream.txt section 16), and the projections with a residual epilogue (o, down) stay on them too.
  // OPADPO_STREAM_DEC64X = fewest token rows routed (default 8; 0 = never).
  static const int route_x = getenv("OPADPO_STREAM_DEC64X") ? atoi(getenv("OPADPO_STREAM_DEC64X")) : 8;
  if (route_x > 0 && a.M >= route_x && stream_hint && g_gemm_variant == 10 && g_dec64_variant != 1 && a.M <= 64 && a.K2 == 0 && !a.bias && !a.R && a.alpha == 1.0f &&
      a.a1_group_n <= 0 && !a.rope_cos && !a.rope_pos && a.K1 % 64 == 0 && a.N % 128 == 0 && a.lda1 % 8 == 0 && a.ldb1 % 8 == 0 && a.ldc % 4 == 0 &&
      (a.act == 0 || (a.act == OPADPO_ACT_SWIGLU_PAIR && !a.out_f32)) && (double)a.M * a.lda1 * 2 < 4.0e9 && (double)a.N * a.ldb1 * 2 < 4.0e9) {
    GemmNTArgs d = a;
    const int mode = a.act == OPADPO_ACT_SWIGLU_PAIR ? 2 : a.out_f32 ? 1 : 0;
    d.act = 0;
    return launch_gemm_nt_dec64(d, mode, 1, st);
  }
  if (a.rope_cos || a.rope_pos) {      // fused rotary embedding: only the 4-wave 256x256 kernel implements it (bf16 out, alpha-only epilogue)
    const bool ok32 = (double)a.M * a.lda1 * 2 < 4.0e9 && (double)a.N * a.ldb1 * 2 < 4.0e9 &&
                      (a.K2 == 0 || ((double)a.M * a.lda2 * 2 < 4.0e9 && (double)a.N * a.ldb2 * 2 < 4.0e9));
    if (a.bias || a.act || a.R || a.out_f32 || a.N % P_BN || !ok32 || a.rope_cols % 128) return hipErrorInvalidValue;
    if (a.rope_cos && (a.rope_L < 4 || (a.rope_seg_len > 0 && a.rope_seg_len < 4))) return hipErrorInvalidValue;
    if (a.rope_pos && !(a.rope_l2theta > 0.f)) return hipErrorInvalidValue;
    const int tiles = ((a.M + P_BM - 1) / P_BM) * (a.N / P_BN);
    W4_LAUNCH(tiles);
    return hipGetLastError();
  }
  if (a.act == OPADPO_ACT_SWIGLU_BWD) {       // fused SwiGLU backward epilogue (R = stored [gate | up], C = [d_gate | d_up]): the 4-wave 256x256 kernel only
    const bool ok32 = (double)a.M * a.lda1 * 2 < 4.0e9 && (double)a.N * a.ldb1 * 2 < 4.0e9 &&
                      (a.K2 == 0 || ((double)a.M * a.lda2 * 2 < 4.0e9 && (double)a.N * a.ldb2 * 2 < 4.0e9));
    if (a.bias || !a.R || a.r_f32 || a.out_f32 || a.alpha != 1.0f || a.N % P_BN || !ok32 || a.ldr % 8 || a.ldc % 8 || a.rope_cos) return hipErrorInvalidValue;
    const int tiles = ((a.M + P_BM - 1) / P_BM) * (a.N / P_BN);
    W4_LAUNCH(tiles);
    return hipGetLastError();
  }
  if (a.act == OPADPO_ACT_SWIGLU_PAIR) {      // fused SwiGLU epilogue: the 4-wave 256x256 kernel, or the weight-streaming kernel for decode
    const bool ok32 = (double)a.M * a.lda1 * 2 < 4.0e9 && (double)a.N * a.ldb1 * 2 < 4.0e9;
    if (a.bias || a.R || a.out_f32 || a.alpha != 1.0f || a.K2 != 0 || a.N % P_BN || !ok32) return hipErrorInvalidValue;
    if (a.M <= 64 && ((stream_hint && g_gemm_variant == 10) || g_gemm_variant == 15) && a.a1_group_n <= 0 && a.K1 % 32 == 0) {
      const int mf = (a.M + 15) / 16;
      const dim3 bl(512), gr(a.N / 32);
      if (g_skinny8 && (a.M <= 16 || (a.M <= 32 && a.N / 32 < 512))) {       // same kernel family as the plain projection of this size
        if (a.M <= 8) hipLaunchKernelGGL((gemm_nt_skinny8_kernel<1, 2, 8, true>), dim3(a.N / 16), bl, 0, st, a);
        else if (a.M <= 16) hipLaunchKernelGGL((gemm_nt_skinny8_kernel<2, 2, 4, true>), dim3(a.N / 16), bl, 0, st, a);
        else if (a.M <= 24) hipLaunchKernelGGL((gemm_nt_skinny8_kernel<3, 2, 2, true>), dim3(a.N / 16), bl, 0, st, a);
        else hipLaunchKernelGGL((gemm_nt_skinny8_kernel<4, 2, 2, true>), dim3(a.N / 16), bl, 0, st, a);
        return hipGetLastError();
      }
      if (mf == 1) hipLaunchKernelGGL((gemm_nt_skinny_kernel<1, 2, 8, true>), gr, bl, 0, st, a);
      else if (mf == 2) hipLaunchKernelGGL((gemm_nt_skinny_kernel<2, 2, 4, true>), gr, bl, 0, st, a);
      else if (mf == 3) hipLaunchKernelGGL((gemm_nt_skinny_kernel<3, 2, 2, true>), gr, bl, 0, st, a);
      else hipLaunchKernelGGL((gemm_nt_skinny_kernel<4, 2, 2, true>), gr, bl, 0, st, a);
      return hipGetLastError();
    }
    const int tiles = ((a.M + P_BM - 1) / P_BM) * (a.N / P_BN);
    W4_LAUNCH(tiles);
    return hipGetLastError();
  }
  // decode-sized problems (M <= 64): weight-streaming kernel, one workgroup per 16 (or 32) weight rows
  if (a.M <= 64 && ((stream_hint && g_gemm_variant == 10) || g_gemm_variant == 15)) {
    const int mf = (a.M + 15) / 16;
    // 32 weight rows per workgroup halve the activation re-reads from L2 (they bound the kernel from M ~ 32 on) but also the
    // number of blocks: measured better only while >= 384 blocks remain (N = 4096 at 128 blocks: 29 vs 24 us at M = 64)
    const bool wide = a.N % 32 == 0 && a.N / 32 >= (mf >= 3 ? 384 : 512) && (a.a1_group_n <= 0 || a.a1_group_n % 32 == 0) &&
                      (a.a2_group_n <= 0 || a.a2_group_n % 32 == 0);
    const dim3 bl(512), gr(wide ? a.N / 32 : a.N / 16);
#define SK(MF_, NR_, U_) hipLaunchKernelGGL((gemm_nt_skinny_kernel<MF_, NR_, U_>), gr, bl, 0, st, a)
    if (g_skinny8 && a.M > 16 && a.M <= 32 && !wide && (a.a1_group_n <= 0 || a.a1_group_n % 16 == 0) && (a.a2_group_n <= 0 || a.a2_group_n % 16 == 0)) {
      // 17..32 tokens, narrow N (q|k|v, o, down at 7B): the whole-cache-line form still wins (M = 24: 37 -> 31, 15 -> 13, 37 -> 28 us);
      // the 32-row form above keeps the wide projections (activation re-reads from L2 dominate there)
      if (a.M <= 24) hipLaunchKernelGGL((gemm_nt_skinny8_kernel<3, 2, 2>), dim3(a.N / 16), bl, 0, st, a);
      else hipLaunchKernelGGL((gemm_nt_skinny8_kernel<4, 2, 2>), dim3(a.N / 16), bl, 0, st, a);
      return hipGetLastError();
    }
    if (g_skinny8 && a.M <= 16 && (a.a1_group_n <= 0 || a.a1_group_n % 32 == 0) && (a.a2_group_n <= 0 || a.a2_group_n % 32 == 0)) {
      const bool wide8 = a.N % 32 == 0 && a.N / 32 >= 512;          // 32 weight rows per workgroup while >= 2 rounds of blocks remain
#define SK8(MF8_, NR_, U_) hipLaunchKernelGGL((gemm_nt_skinny8_kernel<MF8_, NR_, U_>), dim3(a.N / (8 * NR_)), bl, 0, st, a)
      if (a.M <= 8) { if (wide8) SK8(1, 4, 4); else SK8(1, 2, 8); }
      else          { if (wide8) SK8(2, 4, 2); else SK8(2, 2, 4); }
#undef SK8
      return hipGetLastError();
    }
    if (wide) { if (mf == 1) SK(1, 2, 8); else if (mf == 2) SK(2, 2, 4); else if (mf == 3) SK(3, 2, 2); else SK(4, 2, 2); }
    else      { if (mf == 1) SK(1, 1, 8); else if (mf == 2) SK(2, 1, 4); else if (mf == 3) SK(3, 1, 4); else SK(4, 1, 4); }
#undef SK
    return hipGetLastError();
  }
  // variant 10 (default, "auto"): a 256x256 kernel when it yields at least ~1.25 rounds of blocks on the 256 CUs (or fills its
  // last round), the 128x128 kernel otherwise (skinny LoRA GEMMs, N not a multiple of 256).
  const bool off32 = (double)a.M * a.lda1 * 2 < 4.0e9 && (double)a.N * a.ldb1 * 2 < 4.0e9 &&
                     (a.K2 == 0 || ((double)a.M * a.lda2 * 2 < 4.0e9 && (double)a.N * a.ldb2 * 2 < 4.0e9));   // buffer offsets are 32-bit
  const int pp_tiles = (a.N % P_BN == 0 && off32) ? ((a.M + P_BM - 1) / P_BM) * (a.N / P_BN) : 0;
  // one 256x256 block per CU and round: below 1.25 rounds the big tile only pays when its last round fills the chip
  // (M = 32362: N = 512 -> 254 blocks, w4 1.10-1.18 PF/s vs 0.75-0.79 for the 128x128 kernel; N = 768 -> 381 blocks = 1.49
  // rounds, w4 1.10 vs 0.84; N = 256 -> 127 blocks, the 128x128 kernel wins 0.77 vs 0.63)
  // the 4-wave kernels' hot instantiations take alpha-only epilogues; bias / quick-GELU / GELU problems (+ an fp32 residual into an fp32 result) run on
  // their BA instantiations (round 5: the vision tower's and the projector's GEMMs left the 8-wave kernel / the 128x128 kernel for them)
  static const int env_ba = getenv("OPADPO_W4_BA") ? atoi(getenv("OPADPO_W4_BA")) : 1;
  const bool ba = env_ba && (g_gemm_variant == 10 || g_gemm_variant == 31) && w4_direct_ba_ok(a) && a.N % P_BN == 0 && off32 &&
                  (double)(a.M + 256) * a.ldc * (a.out_f32 ? 4 : 2) < 4.0e9;
  const bool plain = (!a.bias && !a.act) || ba;
  const int pp_slots = ((pp_tiles + 255) / 256) * 256;
  // ragged rows (M ~ 24-26 k): N = 512 -> 188-200 blocks in ONE round, w4 0.088 / 0.256 ms (K = 4096 / 11008) vs 0.121 / 0.424 for the
  // 128x128 kernel; N = 768 -> 282-300 blocks: a tie; N = 256 -> 94-100 blocks: the 128x128 kernel keeps a 5-20 % lead
  // with the quarter-tile tail (below) a second, short round costs 0.5-0.6 of a round: N = 768 (282-300 blocks) moves to the 256x256 kernel as
  // well, and a deep-K one-round problem of <= 128 tiles (x . A_d^T: N = 256, K = 11008) runs as quarter tiles entirely (4x the blocks)
  const int pp_rem = pp_tiles % 256, pp_nt = (a.K1 + a.K2) / P_BK;
  const bool deep_small = plain && pp_tiles >= 64 && pp_tiles <= 128 && pp_nt >= 128;      // K = 4096 one-round problems measured no gain (0.085 vs 0.081 ms)
  const bool big = g_gemm_variant == 17 || g_gemm_variant == 31 || (g_w4s_few && g_gemm_variant == 10 && plain && pp_tiles >= 16) ||      // (test switch: small problems stream on 8 workgroups)
                   (g_gemm_variant != 4 && (pp_tiles >= 320 || (plain && pp_tiles >= 150 && (pp_tiles <= 256 || pp_rem <= 128)) || deep_small ||
                                            (plain && pp_tiles >= 200 && pp_tiles * 100 >= pp_slots * 88)));
  if (big && pp_tiles > 0) {
    if (plain && g_gemm_variant != 17) {    // 4 waves x 128x128, long-lead DMA schedule, M0 one MFMA ahead of each DMA
      const int full = pp_tiles / 256 * 256, rem = pp_tiles - full;
      // a partly filled last round (<= 64 tiles) runs as QUARTER tiles on the 128x128 kernel - 4 blocks per 256x256 tile, each over the
      // FULL K range in the same k order, so every element of C is bit-identical to what the 256x256 kernel writes and the result of a row
      // does not depend on how many rows share the batch (WHICH tiles are tail tiles depends on M).
      // (a deep-K problem of <= 128 tiles - x . A_d^T: N = 256, K = 11008 - runs as quarter tiles entirely: 4x the blocks on a chip it
      // would fill to a third; variant 31 = the 256x256 kernel on every tile, the bit-for-bit cross-check of the tests).
      // Measured (GB_ONLY=tail, N = 4096): a 16- or 64-tile tail costs 0.5-0.6 of a round as quarter tiles, a 128-tile tail 1.1-1.3 rounds
      // (two quarter blocks share a CU there) - more than the plain partly filled round: quarter tiles up to 64 tiles only.
      // Round 2's split-K tail (2 / 4 / 8 K-slices per tail tile + a reduce launch, a process-global fp32 workspace) was 1.1 % faster per step
      // (964.4 vs 975.2 ms; no tail handling: 982.8) but re-associated the fp32 sums of the tail tiles, i.e. made a row's result depend on
      // the batch's row count; removed.
      if (g_gemm_variant != 31 && (full > 0 || deep_small) && rem > 0 && rem <= (deep_small && full == 0 ? 128 : 64)) {
        if (full > 0) { if (ba) W4_LAUNCH_BA(full); else W4_LAUNCH(full); }
        GemmNTArgs t = a;
        t.quarter = 1; t.tile0 = full;
        // (a four-stage ring for these blocks - three K-tiles in flight, one block per CU - measured 973.1 vs 969.9 ms per step: no gain,
        // a lone 4-wave 128x128 block runs at ~0.6 PF/s per CU whatever its prefetch depth; removed)
        // (round 4) tails of a multi-round launch run as SIXTEENTH tiles (16 blocks of 64x64 per tile: a 16-tile tail already fills the chip);
        // the deep-K one-round problem keeps its quarter tiles (every tile of it is a "tail" tile: rate matters there).  OPADPO_TAIL16=0: quarter tiles
        // Measured (GB_ONLY=tail, N = 4096; quarter tiles -> sixteenth tiles): 16-tile tail +38 -> +24 us (K = 4096), +115 -> +44 us (K = 11008);
        // 32 tiles +51 -> +34, +115 -> +82; 48 tiles: a tie; 64 tiles (1024 blocks, four per CU): +50 -> +65, i.e. quarter tiles from 48 tiles on.
        static const int tail16 = getenv("OPADPO_TAIL16") ? atoi(getenv("OPADPO_TAIL16")) : 1;
        if (full > 0 && tail16 && rem <= 40) {
          if (rem <= 16) hipLaunchKernelGGL(gemm_nt_tail64_kernel<8>, dim3(rem * 16), dim3(256), 8 * 16384, st, t);
          else hipLaunchKernelGGL(gemm_nt_tail64_kernel<4>, dim3(rem * 16), dim3(256), 4 * 16384, st, t);
        }
        else hipLaunchKernelGGL((gemm_nt_kernel_x<64, true, 2>), dim3(rem * 4), dim3(256), 65536, st, t);
        return hipGetLastError();
      }
      if (ba) W4_LAUNCH_BA(pp_tiles); else W4_LAUNCH(pp_tiles);
    }
    else                                    // bias / activation epilogues (vision tower, projector): 8 waves x 128x64, 4 phases per K-tile
      hipLaunchKernelGGL(gemm_nt_p8_kernel, dim3(pp_tiles), dim3(512), 2 * P_STAGE, st, a);
    return hipGetLastError();
  }
  const int tiles_x = ((a.M + BM - 1) / BM) * (a.N / BN);      // everything else (and variant 4): the 128x128 kernel
  hipLaunchKernelGGL((gemm_nt_kernel_x<64, true, 2>), dim3(tiles_x), dim3(256), 65536, st, a);
  return hipGetLastError();
}

// grid + flush geometry of one launch of gemm_tn_w4_kernel over the group's tile list
static void tn_w4_geometry(GemmTNGroup& G, int& n_runs, int& smax) {
  const int tiles = G.tile_end[G.n - 1], ksteps = (G.g[0].M + 63) / 64;
  G.splits = (256 + tiles - 1) / tiles;                          // K-chunks: about one run per CU and chunk-tile
  if (G.splits > ksteps) G.splits = ksteps;
  const int chunk_len = (ksteps + G.splits - 1) / G.splits;
  const long long total = (long long)G.splits * tiles * chunk_len;
  n_runs = (int)(total < 256 ? total : 256);
  const long long per = (total + n_runs - 1) / n_runs;
  smax = (int)((per - 1) / chunk_len) + 2;                       // units a run of `per` consecutive K-steps can touch
}
static hipError_t tn_w4_launch(GemmTNGroup& G, hipStream_t st, void* ws, size_t ws_bytes, bool* ordered = nullptr) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)gemm_tn_w4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * P_STAGE);
    attr = true;
  }
  int n_runs, smax;
  tn_w4_geometry(G, n_runs, smax);
  const size_t need = (size_t)n_runs * smax * 65536 * sizeof(float);
  const bool det = ws && ws_bytes >= need;
  if (ordered) *ordered = det;
  G.ws = det ? (float*)ws : nullptr;
  G.smax = det ? smax : 0;
  hipLaunchKernelGGL(gemm_tn_w4_kernel, dim3(n_runs), dim3(256), 2 * P_STAGE, st, G);
  if (det) hipLaunchKernelGGL(gemm_tn_reduce_kernel, dim3(G.tile_end[G.n - 1], 4), dim3(256), 0, st, G, n_runs);      // 4 x 4096 float4 = one 256x256 tile
  return hipGetLastError();
}
static bool tn_w4_ok(const GemmTNArgs& a, int M0) {
  const bool off32 = (double)a.M * a.ldp * 2 < 4.0e9 && (double)a.M * a.ldq * 2 < 4.0e9;
  return a.M == M0 && a.M > 0 && a.splits <= 0 && off32 && a.N1 % 256 == 0 && a.N2 % 256 == 0 && (a.q_group_n1 <= 0 || a.q_group_n1 % 256 == 0) &&
         (a.use_tr < 0 ? g_tn_w4 != 0 : (a.use_tr & 8) == 0);
}
size_t gemm_tn_group_workspace_bytes(const GemmTNArgs* list, int n) {
  // worst case over the ways launch_gemm_tn_group may run the list: one grouped launch, or one launch per problem (problems of different M)
  size_t need = 0;
  auto one = [&](const GemmTNArgs* l, int k) {
    GemmTNGroup G;
    G.n = k;
    int tiles = 0;
    for (int i = 0; i < k; ++i) { G.g[i] = l[i]; tiles += (l[i].N1 / 256) * (l[i].N2 / 256); G.tile_end[i] = tiles; }
    int n_runs, smax;
    tn_w4_geometry(G, n_runs, smax);
    need = std::max(need, (size_t)n_runs * smax * 65536 * sizeof(float));
  };
  if (n <= 0 || n > 8) return 0;
  // 0 unless EVERY problem of the list runs on the 256x256 kernel (grouped or alone): a mixed list would send the others to the 128x128
  // kernel's fp32 atomics behind an API that promises a bit-reproducible result
  for (int i = 0; i < n; ++i) if (list[i].M > 0 && !tn_w4_ok(list[i], list[i].M)) return 0;
  bool group = true;
  for (int i = 0; i < n; ++i) group = group && tn_w4_ok(list[i], list[0].M);
  if (group) one(list, n);
  for (int i = 0; i < n; ++i) if (tn_w4_ok(list[i], list[i].M)) one(list + i, 1);
  return need;
}

hipError_t launch_gemm_tn(const GemmTNArgs& a, hipStream_t st) {
  if (a.M <= 0) return hipSuccess;
  if (a.N1 % 128 || a.N2 % 128) return hipErrorInvalidValue;
  const bool use_tr = a.use_tr >= 0 ? (a.use_tr & 1) != 0 : g_use_tr;       // per-call override (opadpo_ctx_set_flags)
  const bool tn_w4 = a.use_tr >= 0 ? (a.use_tr & 8) == 0 : g_tn_w4 != 0;
  const bool off32 = (double)a.M * a.ldp * 2 < 4.0e9 && (double)a.M * a.ldq * 2 < 4.0e9;
  if (tn_w4 && a.splits <= 0 && off32 && a.N1 % 256 == 0 && a.N2 % 256 == 0 && (a.q_group_n1 <= 0 || a.q_group_n1 % 256 == 0)) {
    GemmTNGroup G;
    G.g[0] = a; G.n = 1;
    G.tile_end[0] = (a.N1 / 256) * (a.N2 / 256);
    return tn_w4_launch(G, st, nullptr, 0);
  }
  const int tiles = (a.N1 / 128) * (a.N2 / 128);
  int splits = a.splits;
  if (splits <= 0) {
    splits = (1024 + tiles - 1) / tiles;
    const int max_splits = (a.M + 255) / 256;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
  }
  const dim3 grid(tiles, splits);
  if (use_tr) hipLaunchKernelGGL(gemm_tn_kernel<true>, grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL(gemm_tn_kernel<false>, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

// decode GEMM for up to 64 tokens: C = A[M,K] . B[N,K]^T, no bias / residual / LoRA tail (a merged or adapter-free rollout).
// mode 0: bf16 C[M,N]; 1: fp32, `splits` K-slices -> C[splits][M,N] partial tiles (the consumer adds them: launch_rmsnorm_sum_fwd);
// 2: SwiGLU pair -> bf16 C[M, N/2].  splits <= 0: chosen so that about two workgroups per CU exist.
// ---------------------------------------------------------------------------------------------------
// gemm_nt "dec64x" kernel (round 4, second session): the decode GEMM for 9..64 tokens, rebuilt around what a weight STREAM wants
// (tools/micro/wstream.hip, profiles/r04_decode_stream.txt).  Measured first, on the q|k|v weight (100 MB): a pure read runs at 5.3-5.4 TB/s
// when every load instruction covers whole 128-byte lines (8 rows x 128 B), every wave walks its OWN 16 rows along K with 8+ loads in
// flight and nothing synchronises the waves; at 4.3 TB/s with 16 rows x 64 B per instruction (the register-streaming kernel of the first
// session); at 2.5-4.4 TB/s through a barrier per k-tile (the ring kernel above with its activation half and its MFMAs switched off,
// OPADPO_DEC64_DIAG=5; the barrier alone costs 11-29 %).  Here:
//   * a workgroup = NW weight waves x 16 rows (48 / 64 / 96 / 128 rows; MODE 2: half gate, half up rows - 48 + 48 for the 7B gate|up: 230
//     workgroups on 256 CUs instead of 172 of 128 rows, 44.8 -> 39.6 us at 64 tokens) x one K range (gridDim.y slices, as
//     the ring kernel) + TWO loader waves;
//   * WEIGHTS go global -> registers, two 1-KiB loads per 64-deep k-tile and wave (rows 0-7 and 8-15 of the wave x 128 B, non-temporal), four
//     k-tiles (8 loads) in flight per wave, no LDS.  A load covers 8 rows x 8 chunks of 16 B, an MFMA operand wants 16 rows x 4 chunks:
//     lane l fetches row l & 7, chunk (l >> 4) + 4 * bit3(l) in the first load and the other four chunks in the second, so that
//     A(k 0..31) = bit3 ? second : first is a per-lane select and A(k 32..63) the complementary select rotated by 8 lanes (DPP row_ror:8) -
//     12 VALU per 2 x MF MFMAs;
//   * ACTIVATIONS (<= 64 x K, L2-resident) are shared by the weight waves through LDS in 256-deep chunks: two 32-KiB stages (64 token rows x
//     512 B, 16-byte chunks XOR-ed with token & 15: conflict-free ds_read_b128), filled by LDS-DMA one chunk ahead by the loader waves, ONE
//     barrier per chunk = per 4 k-tiles.  The loaders own that stream because vmcnt counts in order: a weight wave waiting for its own pieces
//     of the next chunk would wait for every older weight load too.
// What bounds it (diagnostic builds, OPADPO_DEC64X_DIAG): NOT the MFMAs / fragment reads (off: +-0), not the barriers (off: 0 ... -13 %), not the
// weight loads' lead (16 instead of 8 in flight: +-0) and not where the activations come from (always the same 32 KiB: +-0) - the ACTIVATION
// PIECES themselves: without them q|k|v runs in 20.3 us = 4.97 TB/s (the pure-stream figure), with them in 29.8 us, linear in the token
// count.  A CU's vector-memory pipeline moves ~35 GB/s whatever the source (the guide's ~10-13 B/clk/CU), weights + activation pieces alike:
// time = bytes per CU.  Hence the rows per workgroup are chosen by shape (launch_gemm_nt_dec64): 48 rows where that makes 256 workgroups
// (q|k|v: one per CU instead of 192 on three quarters of the chip), 128 rows (half the activation bytes per weight byte) where the grid still
// covers the chip (gate|up, the K-split o / down projections, lm_head).  At 64 tokens: q|k|v 36.3 -> 25.8 us, gate|up 53.9 -> 45.3, down
// 23.4 -> 22.1, o 10.6 -> 10.4, lm_head 62.6 -> 57.9; decode step at B = 64 8.63 -> 8.01 ms, B = 32 6.34 -> 5.70, B = 16 5.15 -> 4.84.
// MODEs, K-split and the partial-sum protocol are the ring kernel's (0: bf16 C; 1: fp32 C[split][M][N]; 2: SwiGLU pairs).
// ---------------------------------------------------------------------------------------------------
namespace {
constexpr int X_KT = 4;                                  // k-tiles (64 deep) per activation chunk
constexpr int X_STAGE = 64 * X_KT * D_BK * 2;            // 32 KiB
// OPADPO_DEC64X_DIAG (compile-time, diagnostics only - results WRONG): 1 loaders issue nothing, 2 no fragment reads / MFMAs (weights XOR-ed
// into the accumulators), 4 no barriers in the K loop
#ifndef OPADPO_DEC64X_DIAG
#define OPADPO_DEC64X_DIAG 0
#endif
#ifndef OPADPO_DEC64X_WAUX
#define OPADPO_DEC64X_WAUX 2      // cache policy of the weight loads (bit 0 sc0, bit 1 nt, bit 4 sc1)
#endif
#ifndef OPADPO_DEC64X_XAUX
#define OPADPO_DEC64X_XAUX 0      // cache policy of the activation pieces
#endif
template <int MF, int MODE, int NW>      // NW = weight waves (16 rows each) per workgroup: 3, 4, 6 (SwiGLU pairs) or 8; two more waves load the activations
__global__ __launch_bounds__(64 * (NW + 2), NW == 8 ? 1 : 2) void gemm_nt_dec64x_kernel(GemmNTArgs p) {
  static_assert(NW == 3 || NW == 4 || NW == 6 || NW == 8, "48, 64, 96 or 128 weight rows per workgroup");
  static_assert(MODE != 2 || NW != 3, "SwiGLU pairs: 32 + 32, 48 + 48 or 64 + 64 rows");
  static_assert(MODE == 2 || NW != 6, "96-row workgroups: SwiGLU pairs only");
  constexpr int XDG = OPADPO_DEC64X_DIAG;
  extern __shared__ __attribute__((aligned(16))) char smem[];          // 2 activation stages
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bt = blockIdx.x, split = blockIdx.y, splits = gridDim.y;
  const int nt = p.K1 / D_BK;
  const int per = (nt + splits - 1) / splits;
  const int t0 = split * per, t1 = min(nt, t0 + per);
  const int nk = max(t1 - t0, 0), nc = (nk + X_KT - 1) / X_KT;
  // weight rows of this workgroup: 16 * NW consecutive rows, or (MODE 2: rows per 128 = [64 gate | 64 up]) HALF gate rows + the HALF up rows 64 further
  constexpr int ROWS = 16 * NW, HALF = ROWS / 2;
  // MODE 2 in gate-row units: workgroup bt owns the gate rows g = bt * HALF .. + HALF - 1 and their up rows; gate row g is weight row (g / 64) * 128 +
  // g % 64, its up row 64 further (a wave's 16 rows never straddle a block of 64).  HALF = 48 does not divide the 11008 gate rows of the 7B MLP: the
  // waves of the last workgroup that lie past the end redo its last 16 valid rows (they keep the barriers' count) and store nothing.
  const int n0 = MODE == 2 ? 0 : bt * ROWS;
  const int gw = wave < NW ? (wave < NW / 2 ? wave : wave - NW / 2) : 0;                     // MODE 2: the wave's 16-row group inside its half
  const int g_raw = bt * HALF + gw * 16;
  const bool g_live = MODE != 2 || g_raw < p.N / 2;
  const int g0 = MODE == 2 ? (g_live ? g_raw : p.N / 2 - 16) : 0;                          // first gate row of the wave
  auto wrow = [&](int r) {                                                                // r = row inside the workgroup, wave-major
    if (MODE != 2) return n0 + r;
    const int g = g0 + (r & 15);
    return (g / 64) * 128 + g % 64 + (r >= HALF ? 64 : 0);
  };
  auto uni = [](const void* q) -> void* {
    const unsigned long long v = (unsigned long long)q;
    return (void*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                   (unsigned)__builtin_amdgcn_readfirstlane((int)v));
  };
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(uni(p.B1), 0, (int)(((unsigned)(p.N - 1) * (unsigned)p.ldb1 + (unsigned)p.K1) * 2u), 0x00020000);
  // the last activation chunk of a K range may reach past the end of the rows: the descriptor ends with the last row, the overshoot reads zeros
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(uni(p.A1), 0, (int)(((unsigned)(p.M - 1) * (unsigned)p.lda1 + (unsigned)p.K1) * 2u), 0x00020000);
  const int fr = lane & 15, fc = lane >> 4, hi = (lane >> 3) & 1;
  const unsigned voffWa = (unsigned)wrow(wave * 16 + (lane & 7)) * (unsigned)p.ldb1 * 2u + (unsigned)(fc + 4 * hi) * 16u;
  const unsigned voffWb = (unsigned)wrow(wave * 16 + 8 + (lane & 7)) * (unsigned)p.ldb1 * 2u + (unsigned)(fc + 4 * (hi ^ 1)) * 16u;
  // activation pieces: 1 KiB = 2 token rows x 512 B; the 8 * MF pieces of a chunk are dealt 4 * MF to each of the two LOADER waves (the last two waves).
  // The loaders own the activation stream because vmcnt counts in order: a weight-streaming wave that also waited for its activation pieces
  // of chunk c+1 would wait for every OLDER weight load as well, i.e. would never have more than one chunk of weights in flight across a
  // chunk boundary (measured: 8 or 16 weight loads in flight per wave made no difference, 28 us on q|k|v either way).
  constexpr int XP = 4 * MF;
  if (wave >= NW) {
    const int lw = wave - NW;
    unsigned voffX[XP];
#pragma unroll
    for (int i = 0; i < XP; ++i) {
      const int row = (lw * XP + i) * 2 + (lane >> 5);
      voffX[i] = (unsigned)min(row, p.M - 1) * (unsigned)p.lda1 * 2u + (unsigned)(((lane & 31) ^ (row & 15)) * 16);
    }
    for (int c = 0; c < nc; ++c) {
      // stage c & 1 held chunk c-2: every weight wave left it before it arrived at the barrier that ended chunk c-2 .. started chunk c-1
#pragma unroll
      for (int i = 0; i < XP; ++i) {
#if defined(__HIP_DEVICE_COMPILE__)      // (with this call in its view of the body the HOST pass silently drops the kernel's launch stub - found by bisection; the device pass is the only one that needs it)
        // pieces of token rows >= M are not fetched (49..63 tokens, and the second half of a 16-token fragment): their accumulator columns are never stored
        if (!(XDG & 1) && (lw * XP + i) * 2 < p.M) __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, LDS_PTR(void, smem + (c & 1) * X_STAGE + (lw * XP + i) * 1024), 16, voffX[i], ((XDG & 8) ? 0 : (t0 + c * X_KT)) * (D_BK * 2), 0, OPADPO_DEC64X_XAUX);
#endif
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(XDG & 4)) __syncthreads();                    // barrier c: chunk c is in LDS for everyone
    }
    if constexpr (MODE == 2) { __syncthreads(); __syncthreads(); }      // the epilogue's two barriers
    return;
  }
  // weight slots: the four k-tiles of ONE chunk in flight per wave (8 loads = 8 KiB; sixteen measured the same).  What bounds a wave is
  // its own dependent chain per k-tile - wait for the weights, 12 VALU, fragment reads, 2 x MF MFMAs - so the activation fragments of
  // k-tile j+1 are requested before the MFMAs of k-tile j (two register images).
  u32x4_t wa[X_KT], wb[X_KT];
  auto load_w = [&](int slot, int t) {
    // past the K range: an offset beyond the descriptor's end - the load returns zeros without touching memory, and the load COUNT of a
    // chunk stays static (the compiler's counted vmcnt waits survive)
    const bool in = t < t1;
    const int k2 = t * (D_BK * 2);
    wa[slot] = __builtin_amdgcn_raw_buffer_load_b128(rW, in ? voffWa : 0xfffffff0u, in ? k2 : 0, OPADPO_DEC64X_WAUX);       // aux 2 = nt: streamed once
    wb[slot] = __builtin_amdgcn_raw_buffer_load_b128(rW, in ? voffWb : 0xfffffff0u, in ? k2 : 0, OPADPO_DEC64X_WAUX);
  };
  f32x4_t acc[MF];
#pragma unroll
  for (int f = 0; f < MF; ++f) acc[f] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  bf16x8_t xr[2][MF][2];
  auto xread = [&](int buf, int kt, const char* xs) {     // activation fragments of k-tile kt of the chunk at xs
    if (XDG & 2) return;
#pragma unroll
    for (int f = 0; f < MF; ++f) {
      const char* xrow = xs + (f * 16 + fr) * (X_KT * D_BK * 2);
      xr[buf][f][0] = *(const bf16x8_t*)(xrow + (((kt * 8 + fc) ^ fr) << 4));
      xr[buf][f][1] = *(const bf16x8_t*)(xrow + (((kt * 8 + 4 + fc) ^ fr) << 4));
    }
  };
  auto mma = [&](int slot, int buf) {
    if (XDG & 2) { acc[0][0] += __uint_as_float((wa[slot][0] ^ wb[slot][1] ^ wa[slot][2] ^ wb[slot][3] ^ wb[slot][0] ^ wa[slot][1] ^ wb[slot][2] ^ wa[slot][3]) & 0x3f800000u); return; }
    u32x4_t a1, tt, a2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      a1[e] = hi ? wb[slot][e] : wa[slot][e];
      tt[e] = hi ? wa[slot][e] : wb[slot][e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) a2[e] = (unsigned)__builtin_amdgcn_update_dpp(0, (int)tt[e], 0x128, 0xf, 0xf, true);      // row_ror:8
#pragma unroll
    for (int f = 0; f < MF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&a1, xr[buf][f][0], acc[f], 0, 0, 0);
#pragma unroll
    for (int f = 0; f < MF; ++f) acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8_t*)&a2, xr[buf][f][1], acc[f], 0, 0, 0);
  };
  if (nc > 0) {
#pragma unroll
    for (int j = 0; j < X_KT; ++j) load_w(j, t0 + j);
    if (!(XDG & 4)) __syncthreads();                      // barrier 0: chunk 0 of the activations has landed (loader waves)
    int c = 0;
    for (; c + 1 < nc; ++c) {                             // a full chunk with a successor: static issue pattern, counted waits
      const char* xs = smem + (c & 1) * X_STAGE;
      xread(0, 0, xs);
#pragma unroll
      for (int j = 0; j < X_KT; ++j) {
        if (j + 1 < X_KT) xread((j + 1) & 1, j + 1, xs);
        mma(j, j & 1);
        __builtin_amdgcn_sched_barrier(0);
        load_w(j, t0 + (c + 1) * X_KT + j);               // the same slot, one chunk on
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(XDG & 4)) __syncthreads();                    // barrier c+1: chunk c+1 is in (loaders); everyone is done reading chunk c
    }
    const int rem = nk - c * X_KT;                        // 1..4 k-tiles in the last chunk, nothing left to issue
    const char* xs = smem + (c & 1) * X_STAGE;
    xread(0, 0, xs);
#pragma unroll
    for (int j = 0; j < X_KT; ++j)
      if (j < rem) {
        if (j + 1 < rem) xread((j + 1) & 1, j + 1, xs);
        mma(j, j & 1);
      }
  }
  // lane holds C[token f*16 + fr][4 consecutive columns fc*4 .. +3 of the wave's 16 weight rows]
  if constexpr (MODE == 2) {
    __syncthreads();                                           // the stages are dead: reuse them for the gate / up exchange
    float* ex = (float*)smem;                                  // [NW/2 up waves][MF][64 lanes][4]
    if (wave >= NW / 2) {
#pragma unroll
      for (int f = 0; f < MF; ++f) *(f32x4_t*)(ex + (((wave - NW / 2) * MF + f) * 64 + lane) * 4) = acc[f];
    }
    __syncthreads();
    if (wave < NW / 2) {
#pragma unroll
      for (int f = 0; f < MF; ++f) {
        const int m = f * 16 + fr;
        if (m >= p.M) continue;
        const f32x4_t u = *(const f32x4_t*)(ex + ((wave * MF + f) * 64 + lane) * 4);
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float gt = bf2f(f2bf(acc[f][e])), up = bf2f(f2bf(u[e]));
          o[e] = gt / (1.0f + __expf(-gt)) * up;
        }
        uint2 stv;
        stv.x = pack_bf2(o[0], o[1]);
        stv.y = pack_bf2(o[2], o[3]);
        if (g_live) *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + g0 + fc * 4) = stv;
      }
    }
    return;
  }
#pragma unroll
  for (int f = 0; f < MF; ++f) {
    const int m = f * 16 + fr;
    if (m >= p.M) continue;
    const int n = n0 + wave * 16 + fc * 4;
    if constexpr (MODE == 1) {
      *(float4*)((float*)p.C + ((size_t)split * p.M + m) * p.ldc + n) = make_float4(acc[f][0], acc[f][1], acc[f][2], acc[f][3]);
    } else {
      uint2 stv;
      stv.x = pack_bf2(acc[f][0], acc[f][1]);
      stv.y = pack_bf2(acc[f][2], acc[f][3]);
      *(uint2*)((bf16_t*)p.C + (size_t)m * p.ldc + n) = stv;
    }
  }
}
}  // namespace

int gemm_nt_dec64_splits(int N, int K, int splits);
hipError_t launch_gemm_nt_dec64(const GemmNTArgs& a, int mode, int splits, hipStream_t st) {
  if (a.M <= 0) return hipSuccess;
  if (a.M > 64 || a.N % 128 || a.K1 % D_BK || a.K1 <= 0 || a.K2 != 0 || a.lda1 % 8 || a.ldb1 % 8 || mode < 0 || mode > 2) return hipErrorInvalidValue;
  if ((double)a.M * a.lda1 * 2 >= 4.0e9 || (double)a.N * a.ldb1 * 2 >= 4.0e9) return hipErrorInvalidValue;          // 32-bit buffer offsets
  const int nt = a.K1 / D_BK, tiles = a.N / D_BN;
  if (mode != 1) splits = 1;
  splits = gemm_nt_dec64_splits(a.N, a.K1, splits);
  const int mf = (a.M + 15) / 16;
  // g_dec64_variant 1 keeps the LDS-ring kernel below (the kernel of rounds 2-4, the bit-for-bit reference of the tests); the register-streaming
  // dec64r kernel of round 4 (q|k|v only, +3 %) is gone: dec64x beats it by 29 % there (profiles/r04_decode_stream.txt).
  if (g_dec64_variant == 3 || g_dec64_variant == 0) {      // dec64x (the default): weights global -> registers in whole lines, activations through LDS in 256-deep chunks
    // Rows per workgroup (16 x NW) by shape.  What bounds these launches is the vector-memory pipeline of a CU (weights + activation pieces,
    // ~35 GB/s per CU whatever the source: profiles/r04_decode_stream.txt), so the estimate is bytes per CU = rounds of workgroups over the
    // 256 CUs x (weight rows + token rows) of a workgroup: q|k|v (12288 rows) -> 48-row workgroups (256 of them), gate|up -> 128 rows
    // (172), the K-split projections and the head -> 128 rows.
    static const int nw_env = getenv("OPADPO_DEC64X_NW") ? atoi(getenv("OPADPO_DEC64X_NW")) : 0;      // diagnostics
    const int nw_force = g_dec64x_nw == 1 ? 3 : g_dec64x_nw == 2 ? 4 : g_dec64x_nw == 3 ? 8 : nw_env;
    int nw = 4;
    {
      long best = -1;
      const int cand[4] = {8, 6, 4, 3};
      for (int ci = 0; ci < 4; ++ci) {
        const int c = cand[ci];
        // 96-row workgroups (48 gate + 48 up rows) exist for the SwiGLU-pair mode only, where the last workgroup may be partly filled
        if (mode == 2 ? (c == 3 || (c != 6 && a.N % (16 * c))) : (c == 6 || a.N % (16 * c))) continue;
        const long wgs = (mode == 2 ? (long)(a.N / 2 + 8 * c - 1) / (8 * c) : (long)(a.N / (16 * c))) * splits;
        long cost = ((wgs + 255) / 256) * (16 * c + 16 * mf);
        if (c == 4 && wgs > 448 && wgs <= 512) cost = cost * 3 / 4;      // two full rounds of 64-row workgroups, two per CU: measured ahead of one round of 128-row ones (lm_head, 500: 46.8 vs 52.9 us at 8 tokens, 49.8 vs 53.6 at 32, 57.5 vs 57.9 at 64)
        if (best < 0 || cost < best) { best = cost; nw = c; }
      }
      if (nw_force == 3 || nw_force == 4 || nw_force == 8) { if (a.N % (16 * nw_force) == 0 && !(mode == 2 && nw_force == 3)) nw = nw_force; }
      if (nw_force == 6 && mode == 2) nw = 6;
    }
    const int grid_x = mode == 2 ? (a.N / 2 + 8 * nw - 1) / (8 * nw) : a.N / (16 * nw);
#define DX_GO(MF_, MD_, NW_)                                                                                                          \
  do {                                                                                                                                \
    static bool at_ = false;                                                                                                          \
    if (!at_) { (void)hipFuncSetAttribute((const void*)gemm_nt_dec64x_kernel<MF_, MD_, NW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * X_STAGE); at_ = true; } \
    hipLaunchKernelGGL((gemm_nt_dec64x_kernel<MF_, MD_, NW_>), dim3(grid_x, splits), dim3(64 * (NW_ + 2)), 2 * X_STAGE, st, a); \
  } while (0)
#define DX_MF(MD_, NW_) do { if (mf == 1) DX_GO(1, MD_, NW_); else if (mf == 2) DX_GO(2, MD_, NW_); else if (mf == 3) DX_GO(3, MD_, NW_); else DX_GO(4, MD_, NW_); } while (0)
#define DX_MODE(MD_) do { if (nw == 8) DX_MF(MD_, 8); else if (nw == 3) DX_MF(MD_, 3); else DX_MF(MD_, 4); } while (0)
    if (mode == 0) DX_MODE(0);
    else if (mode == 1) DX_MODE(1);
    else { if (nw == 8) DX_MF(2, 8); else if (nw == 6) DX_MF(2, 6); else DX_MF(2, 4); }
#undef DX_MODE
#undef DX_MF
#undef DX_GO
    return hipGetLastError();
  }
  static const int kw2_max = getenv("OPADPO_DEC64_KW2") ? atoi(getenv("OPADPO_DEC64_KW2")) : 256;      // diagnostics
  if (mode == 0 && tiles <= kw2_max) {      // 32-row workgroups: twice the blocks for a projection that does not fill the chip
    static bool at[5] = {false, false, false, false, false};
    const dim3 g2(2 * tiles), b2(256);
#define D_GO2(MF_)                                                                                                                    \
  do {                                                                                                                                \
    if (!at[MF_]) { (void)hipFuncSetAttribute((const void*)gemm_nt_dec64_kernel<MF_, 0, 4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * D_STAGE); at[MF_] = true; } \
    hipLaunchKernelGGL((gemm_nt_dec64_kernel<MF_, 0, 4, 2>), g2, b2, 4 * D_STAGE, st, a);                                             \
  } while (0)
    if (mf == 1) D_GO2(1); else if (mf == 2) D_GO2(2); else if (mf == 3) D_GO2(3); else D_GO2(4);
#undef D_GO2
    return hipGetLastError();
  }
  const dim3 gr(tiles, splits), bl(256);
  static const int deep_max = getenv("OPADPO_DEC64_DEEP") ? atoi(getenv("OPADPO_DEC64_DEEP")) : 256;      // diagnostics
  const bool deep = tiles * splits <= deep_max;       // at most one workgroup per CU: a deeper ring instead of a second workgroup
#define D_GO(MF_, MD_)                                                                                                             \
  do {                                                                                                                             \
    if (deep) {                                                                                                                    \
      static bool at8 = false;                                                                                                     \
      if (!at8) { (void)hipFuncSetAttribute((const void*)gemm_nt_dec64_kernel<MF_, MD_, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * D_STAGE); at8 = true; } \
      hipLaunchKernelGGL((gemm_nt_dec64_kernel<MF_, MD_, 8>), gr, bl, 8 * D_STAGE, st, a);                                         \
    } else {                                                                                                                       \
      static bool at4 = false;                                                                                                     \
      if (!at4) { (void)hipFuncSetAttribute((const void*)gemm_nt_dec64_kernel<MF_, MD_, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * D_STAGE); at4 = true; } \
      hipLaunchKernelGGL((gemm_nt_dec64_kernel<MF_, MD_, 4>), gr, bl, 4 * D_STAGE, st, a);                                         \
    }                                                                                                                              \
  } while (0)
#define D_MODE(MD_) do { if (mf == 1) D_GO(1, MD_); else if (mf == 2) D_GO(2, MD_); else if (mf == 3) D_GO(3, MD_); else D_GO(4, MD_); } while (0)
  if (mode == 0) D_MODE(0); else if (mode == 1) D_MODE(1); else D_MODE(2);
#undef D_MODE
#undef D_GO
  return hipGetLastError();
}
int gemm_nt_dec64_splits(int N, int K, int splits) {
  const int nt = K / D_BK, tiles = N / D_BN;
  if (splits <= 0) {
    // workgroup target of the K-split, counted in 64-row tiles: 256 = four K-slices for the 7B o / down projections (the dec64x kernel then
    // runs them as 64-row workgroups x 4 slices).  Half the slices of round 2's target of 512 are half the partial tiles for the RMSNorm that
    // adds them: decode step B = 64 8.00 -> 7.90 ms, B = 32 5.65 -> 5.59 (384: 8.03, 128: 8.42; profiles/r04_decode_stream.txt section 6).
    // The LDS-ring kernel (OPADPO_DEC64_V=1) keeps 512.
    static const int env_target = getenv("OPADPO_DEC64_BLOCKS") ? atoi(getenv("OPADPO_DEC64_BLOCKS")) : 0;      // diagnostics
    const int target = env_target > 0 ? env_target : (g_dec64_variant == 1 ? 512 : 256);
    splits = (target + tiles - 1) / tiles;
    if (splits > nt / 4) splits = nt / 4 > 0 ? nt / 4 : 1;
  }
  return splits > nt ? nt : splits;
}

hipError_t launch_gemm_tn_group(const GemmTNArgs* list, int n, hipStream_t st, void* workspace, size_t workspace_bytes, int* all_ordered) {
  // *all_ordered (optional) = 1 when EVERY problem of this call left through the ordered reduce (bit-reproducible), 0 when any took fp32 atomics
  if (all_ordered) *all_ordered = 1;
  if (n <= 0) return hipSuccess;
  bool ok = n <= 8;
  for (int i = 0; i < n && ok; ++i) ok = tn_w4_ok(list[i], list[0].M);
  if (!ok) {                                   // not groupable (e.g. the compact top layer: two row counts): one launch per problem
    for (int i = 0; i < n; ++i) {
      hipError_t e;
      if (list[i].M > 0 && tn_w4_ok(list[i], list[i].M)) {
        GemmTNGroup G;
        G.g[0] = list[i]; G.n = 1;
        G.tile_end[0] = (list[i].N1 / 256) * (list[i].N2 / 256);
        bool ord = false;
        e = tn_w4_launch(G, st, workspace, workspace_bytes, &ord);      // launches of one stream run in order: the workspace is free again
        if (all_ordered && !ord) *all_ordered = 0;
      } else {
        e = launch_gemm_tn(list[i], st);
        if (all_ordered && list[i].M > 0) *all_ordered = 0;
      }
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  GemmTNGroup G;
  G.n = n;
  int tiles = 0;
  for (int i = 0; i < n; ++i) {
    G.g[i] = list[i];
    tiles += (list[i].N1 / 256) * (list[i].N2 / 256);
    G.tile_end[i] = tiles;
  }
  bool ord = false;
  const hipError_t e = tn_w4_launch(G, st, workspace, workspace_bytes, &ord);
  if (all_ordered && !ord) *all_ordered = 0;
  return e;
}
