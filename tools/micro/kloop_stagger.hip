// Micro-benchmark (diagnostic, not part of the library): what does the w4 GEMM K-loop pay for the things it carries between its MFMAs?
//   (1) LDS-DMA pieces issued by the four waves of a workgroup in the SAME MFMA slot (what gemm_nt_w4_kernel does: the waves run in lock
//       step behind the barriers, so the CU's one texture-address unit gets four 1-KiB requests at once) vs STAGGERED by wave;
//   (2) scalar fillers (s_add / s_cselect style) between MFMAs - the price of the SALU instructions the K-loop carried before its diet;
//   (3) the same streams with ONE wave per CU (no contention at all) as the floor.
//   hipcc --offload-arch=gfx950 -O3 -o kloop_stagger kloop_stagger.hip && ./kloop_stagger
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <type_traits>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// 64 MFMAs (16x16x32, 1024+ cycles) per iteration.  EVERY: one DMA per EVERY MFMAs; STAG: wave w issues STAG*w MFMAs later (mod EVERY);
// SALU: scalar filler instructions per MFMA gap (0..3); READS: one ds_read_b128 per READS MFMAs (0 = none)
template <int EVERY, int STAG, int SALU, int READS, int W, int AHEAD, int ADDR = 0>
__device__ __forceinline__ void body(const char* src, char* smem, int wave, int lane, int iters, f32x4_t (&acc)[16], bf16x8_t& sink, const i32x4_t rq, unsigned& sacc) {
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.01f); b[i] = (__bf16)(i * 0.5f); }
  const unsigned ldsb = (unsigned)(size_t)LDS_PTR(smem) + wave * 16384;
  unsigned pc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) pc[q] = (unsigned)__builtin_amdgcn_readfirstlane((int)(ldsb + q * 1024));
  constexpr int OFF = EVERY ? (STAG * W) % EVERY : 0;
  // ADDR 3: as 1 with the 8-row group drawn pseudo-randomly from a 1-GiB buffer (every piece misses the L2s and the Infinity Cache);
  // ADDR 0: a piece = 1 KiB contiguous; 1: 8 rows x 128 B, rows 8 KiB apart (what a GEMM operand tile is); 2: the same with the 16-byte chunks of a row
  // XOR-permuted by the row (the swizzle gemm_nt_w4 applies on the source side)
  const unsigned vo = ADDR == 0 ? lane * 16u : ((unsigned)((ADDR == 3 ? 0 : wave * 128) + (lane >> 3)) * 8192u + (unsigned)(((lane & 7) ^ (ADDR == 2 ? (lane >> 3) & 7 : 0)) * 16));
  for (int it = 0; it < iters; ++it) {
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(((it * 16) & 1023) * 1024);
#pragma unroll
    for (int m = 0; m < 64; ++m) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m & 15]) : "v"(b), "v"(a));
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (EVERY > 0) {
        // AHEAD = MFMAs between the M0 write and its DMA (0: back to back; EVERY: right after the previous piece's DMA); AHEAD >= 100: s_add_u32 m0, m0, imm form
        constexpr int AH = AHEAD % 100;
        auto m0w = [&](int q) {
          if constexpr (AHEAD >= 100) asm volatile("s_add_u32 m0, m0, 0x400" ::: "memory", "scc");
          else asm volatile("s_pack_ll_b32_b16 m0, %0, %1" :: "s"(pc[q & 15]), "s"(0) : "memory");
        };
        if (AH < EVERY && (m + 1 + AH + EVERY - OFF) % EVERY == 0) m0w((m + 1 + AH + EVERY - OFF) / EVERY - 1);
        if ((m + 1 + EVERY - OFF) % EVERY == 0) {
          const int q = ((m + 1 + EVERY - OFF) / EVERY - 1) & 15;
          asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(vo), "s"(rq), "s"(ADDR == 0 ? soff + q * 1024 : ADDR == 3 ? (unsigned)((((unsigned)(it * 1024 + blockIdx.x * 4 + wave) * 16u + q) * 2654435761u) % 130000u) * 8192u + (unsigned)((it & 63) * 128) : (unsigned)(q * 8 * 8192 + (it & 63) * 128)) : "memory");
          if (AH >= EVERY) m0w(q + 1);
        }
      }
      if constexpr (READS > 0) {
        if ((m + 1) % READS == 0) sink = *(const bf16x8_t*)(smem + wave * 16384 + (m & 15) * 1024 + lane * 16);
      }
#pragma unroll
      for (int s = 0; s < SALU; ++s) asm volatile("s_add_u32 %0, %0, %1" : "+s"(sacc) : "s"(soff) : "scc");
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (EVERY > 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
}

template <int EVERY, int STAG, int SALU, int READS, int AHEAD, int ADDR = 0>
__global__ __launch_bounds__(256) void k(const char* src, unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  i32x4_t rq;
  rq[0] = __builtin_amdgcn_readfirstlane((int)(unsigned long long)src); rq[1] = __builtin_amdgcn_readfirstlane((int)((unsigned long long)src >> 32));
  rq[2] = -1; rq[3] = 0x00020000;
  f32x4_t acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4_t{0, 0, 0, 0};
  bf16x8_t sink; for (int i = 0; i < 8; ++i) sink[i] = (__bf16)0.f;
  unsigned sacc = 0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  if (STAG == 0 || wave == 0) body<EVERY, STAG, SALU, READS, 0, AHEAD, ADDR>(src, smem, wave, lane, iters, acc, sink, rq, sacc);
  else if (wave == 1) body<EVERY, STAG, SALU, READS, 1, AHEAD, ADDR>(src, smem, wave, lane, iters, acc, sink, rq, sacc);
  else if (wave == 2) body<EVERY, STAG, SALU, READS, 2, AHEAD, ADDR>(src, smem, wave, lane, iters, acc, sink, rq, sacc);
  else body<EVERY, STAG, SALU, READS, 3, AHEAD, ADDR>(src, smem, wave, lane, iters, acc, sink, rq, sacc);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = (float)sink[0] + (float)sacc;
  for (int i = 0; i < 16; ++i) s += acc[i][0];
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = (r1 - r0) + ((unsigned long long)(s == 12345.f) << 60); }
}

template <int EVERY, int STAG, int SALU, int READS, int AHEAD = 1, int ADDR = 0>
void run(const char* name, const char* src, unsigned long long* out, int threads) {
  const int iters = 20000, blocks = 256, nw = threads / 64;
  hipFuncSetAttribute((const void*)k<EVERY, STAG, SALU, READS, AHEAD, ADDR>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<EVERY, STAG, SALU, READS, AHEAD, ADDR>), dim3(blocks), dim3(threads), 65536, 0, src, out, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0, rt = 0; int n = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < nw; ++w) { c += h[(b * 4 + w) * 2]; rt += h[(b * 4 + w) * 2 + 1] & 0xffffffffffull; ++n; }
  const double mhz = c / rt * 100.0;
  c /= (double)n * iters;
  const int nmem = EVERY ? 64 / EVERY : 0;
  printf("%-66s waves/CU %d  %7.1f cycles per 64 MFMAs  (+%.1f per DMA piece if all of the excess is theirs)  clock %.0f MHz\n", name, nw, c,
         nmem ? (c - 1044.0) / nmem : 0.0, mhz);
}

int main() {
  char* src; unsigned long long* out;
  hipMalloc(&src, 1100u << 20); hipMemset(src, 1, 1100u << 20);
  hipMalloc(&out, 1 << 20);
  for (int threads : {256}) {
    run<0, 0, 0, 0>("bare MFMAs", src, out, threads);
    run<0, 0, 1, 0>("1 SALU per MFMA gap", src, out, threads);
    run<0, 0, 2, 0>("2 SALU per MFMA gap", src, out, threads);
    run<0, 0, 3, 0>("3 SALU per MFMA gap", src, out, threads);
    run<4, 0, 0, 0>("DMA every 4, same slot in all waves", src, out, threads);
    run<4, 1, 0, 0>("DMA every 4, staggered by wave", src, out, threads);
    run<8, 0, 0, 0>("DMA every 8, same slot", src, out, threads);
    run<8, 2, 0, 0>("DMA every 8, staggered by 2 per wave", src, out, threads);
    run<4, 0, 0, 2>("DMA every 4 same slot + ds_read_b128 every 2", src, out, threads);
    run<4, 1, 0, 2>("DMA every 4 staggered + ds_read_b128 every 2", src, out, threads);
    run<4, 0, 0, 0, 1, 1>("DMA every 4, piece = 8 rows x 128 B (8 KiB apart)", src, out, threads);
    run<4, 0, 0, 0, 1, 2>("DMA every 4, piece = 8 rows x 128 B, chunks XOR-permuted by row", src, out, threads);
    run<4, 0, 0, 2, 1, 1>("DMA every 4 (8 rows x 128 B) + ds_read_b128 every 2", src, out, threads);
    run<4, 0, 0, 2, 1, 2>("DMA every 4 (8 rows x 128 B, permuted) + ds_read_b128 every 2", src, out, threads);
    run<4, 0, 0, 0, 1, 3>("DMA every 4, piece = 8 rows x 128 B from HBM (no cache hits)", src, out, threads);
    run<8, 0, 0, 0, 1, 3>("DMA every 8, piece = 8 rows x 128 B from HBM (no cache hits)", src, out, threads);
    run<4, 0, 0, 0, 0>("DMA every 4, M0 written back to back with the DMA", src, out, threads);
    run<4, 0, 0, 0, 2>("DMA every 4, M0 written 2 MFMAs ahead", src, out, threads);
    run<4, 0, 0, 0, 3>("DMA every 4, M0 written 3 MFMAs ahead", src, out, threads);
    run<4, 0, 0, 0, 4>("DMA every 4, M0 written right after the previous DMA", src, out, threads);
    run<4, 0, 0, 0, 104>("DMA every 4, s_add_u32 m0, m0, imm right after the previous DMA", src, out, threads);
    run<4, 0, 0, 0, 101>("DMA every 4, s_add_u32 m0, m0, imm 1 MFMA ahead", src, out, threads);
    run<4, 0, 0, 2, 4>("DMA every 4 (M0 after previous DMA) + ds_read_b128 every 2", src, out, threads);
    run<2, 0, 0, 0, 1>("DMA every 2, M0 1 ahead", src, out, threads);
    run<2, 0, 0, 0, 2>("DMA every 2, M0 right after the previous DMA", src, out, threads);
    run<0, 0, 0, 2>("ds_read_b128 every 2", src, out, threads);
    run<0, 0, 0, 1>("ds_read_b128 every 1", src, out, threads);
  }
  return 0;
}
