// Micro-benchmark (diagnostic, not part of the library): how many cycles does one LDS-DMA instruction
// (buffer_load_dwordx4 ... lds) cost a single wave per SIMD that is otherwise issuing back-to-back MFMAs?
//   hipcc --offload-arch=gfx950 -O3 -o mfma_dma mfma_dma.hip && ./mfma_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

// MODE 0: 16x16x32 (16 cycles), MODE 1: 32x32x16 (32 cycles).  EVERY: one memory instruction per EVERY MFMAs (0 = none).
// KIND 0: LDS-DMA, 1: ds_read_b128, 2: plain buffer_load_dwordx4 into VGPRs
template <int MODE, int EVERY, int KIND>
__global__ __launch_bounds__(256) void k(const char* src, unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0xffffffffu, 0x00020000);
  typedef __attribute__((ext_vector_type(4))) int i32x4_t;
  i32x4_t rq;
  rq[0] = __builtin_amdgcn_readfirstlane((int)(unsigned long long)src); rq[1] = __builtin_amdgcn_readfirstlane((int)((unsigned long long)src >> 32));
  rq[2] = -1; rq[3] = 0x00020000;
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(lane * 0.01f); b[i] = (__bf16)(i * 0.5f); }
  f32x4_t acc4[16]; f32x16_t acc16[4];
  for (int i = 0; i < 16; ++i) acc4[i] = f32x4_t{0, 0, 0, 0};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0;
  bf16x8_t sink = a; uint4 vs = {0, 0, 0, 0};
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    constexpr int N = MODE == 0 ? 64 : 32;       // same MFMA-pipe time per iteration: 1024 cycles
#pragma unroll
    for (int m = 0; m < N; ++m) {
      if constexpr (EVERY > 0 && KIND == 3) {
        if ((m + 2) % EVERY == 0) {       // one MFMA ahead of the DMA: M0 = LDS destination
          const int q = (m + 2) / EVERY - 1;
          const unsigned dst = (unsigned)(size_t)LDS_PTR(smem) + wave * 16384 + (q & 15) * 1024;
          asm volatile("s_mov_b32 m0, %0" :: "s"(dst) : "memory");
        }
      }
      if constexpr (MODE == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc4[m & 15]) : "v"(b), "v"(a));
      else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc16[m & 3]) : "v"(b), "v"(a));
      if constexpr (EVERY > 0) {
        if ((m + 1) % EVERY == 0) {
          const int q = (m + 1) / EVERY - 1;
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (KIND == 0)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, LDS_PTR(smem + wave * 16384 + (q & 15) * 1024), 16, lane * 16, ((it * 16 + q) & 1023) * 1024, 0, 0);
          else if constexpr (KIND == 3)
            asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" :: "v"(lane * 16), "s"(rq), "s"(((it * 16 + q) & 1023) * 1024) : "memory");
          else if constexpr (KIND == 4) {
            typedef __attribute__((ext_vector_type(4))) short s16x4_t;
            union { bf16x8_t v; s16x4_t h[2]; } u;
            u.v = sink;
            u.h[q & 1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(smem + wave * 16384 + (q & 15) * 1024 + ((lane >> 4) * 8 + ((lane & 15) >> 2)) * 64 + (lane & 3) * 8));
            sink = u.v;
          } else if constexpr (KIND == 1)
            sink = *(const bf16x8_t*)(smem + wave * 16384 + (q & 15) * 1024 + lane * 16);
          else
            vs = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, ((it * 16 + q) & 1023) * 1024, 0));
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if constexpr (EVERY > 0 && KIND != 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = sink[0] + (float)vs.x;
  for (int i = 0; i < 16; ++i) s += acc4[i][0];
  for (int i = 0; i < 4; ++i) s += acc16[i][0];
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = t1 - t0; out[(blockIdx.x * 4 + wave) * 2 + 1] = (r1 - r0) + ((unsigned long long)(s == 12345.f) << 60); }
}

template <int MODE, int EVERY, int KIND>
void run(const char* name, const char* src, unsigned long long* out, int blocks) {
  const int iters = 20000;
  hipFuncSetAttribute((const void*)k<MODE, EVERY, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, EVERY, KIND>), dim3(blocks), dim3(256), 65536, 0, src, out, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
  double c = 0, rt = 0; for (int i = 0; i < blocks * 4; ++i) { c += h[i * 2]; rt += h[i * 2 + 1] & 0xffffffffffull; }
  const double mhz = c / rt * 100.0;
  c /= blocks * 4.0 * iters;
  const int nmem = EVERY ? (MODE == 0 ? 64 : 32) / EVERY : 0;
  printf("%-44s %7.1f cycles per 1024-cycle MFMA block  (%d mem instr -> %.1f extra cycles each)  clock %.0f MHz -> %.2f PF/s dense-equivalent\n", name, c, nmem,
         nmem ? (c - 1024.0) / nmem : 0.0, mhz, 1024.0 / c * mhz * 1e6 * 1024 * 1017.0 / 1e15 * (blocks / 256.0 > 1 ? 1 : blocks / 256.0));
}

int main() {
  char* src; unsigned long long* out;
  hipMalloc(&src, 64 << 20); hipMemset(src, 1, 64 << 20);
  hipMalloc(&out, 1 << 20);
  const int B = 256;
  run<0, 0, 0>("16x16x32, no memory instr", src, out, B);
  run<1, 0, 0>("32x32x16, no memory instr", src, out, B);
  run<0, 8, 0>("16x16x32 + LDS-DMA every 8 (8 per block)", src, out, B);
  run<0, 4, 0>("16x16x32 + LDS-DMA every 4 (16 per block)", src, out, B);
  run<1, 4, 0>("32x32x16 + LDS-DMA every 4 (8 per block)", src, out, B);
  run<1, 2, 0>("32x32x16 + LDS-DMA every 2 (16 per block)", src, out, B);
  run<0, 4, 3>("16x16x32 + LDS-DMA (M0 set 1 MFMA ahead) /4", src, out, B);
  run<0, 2, 3>("16x16x32 + LDS-DMA (M0 set 1 MFMA ahead) /2", src, out, B);
  run<0, 4, 2>("16x16x32 + buffer_load->VGPR every 4 (16)", src, out, B);
  run<1, 2, 2>("32x32x16 + buffer_load->VGPR every 2 (16)", src, out, B);
  run<0, 1, 4>("16x16x32 + ds_read_b64_tr_b16 every 1 (64)", src, out, B);
  run<0, 2, 4>("16x16x32 + ds_read_b64_tr_b16 every 2 (32)", src, out, B);
  run<0, 1, 1>("16x16x32 + ds_read_b128 every 1 (64)", src, out, B);
  run<0, 2, 1>("16x16x32 + ds_read_b128 every 2 (32)", src, out, B);
  run<1, 1, 1>("32x32x16 + ds_read_b128 every 1 (32)", src, out, B);
  return 0;
}
