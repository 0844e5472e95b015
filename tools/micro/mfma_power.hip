// Round 6 probe: does the MFMA SHAPE change what a power-limited MI355X sustains?  The training GEMM's main loop runs v_mfma_f32_16x16x32_bf16 (a 16x32 fragment per
// operand, each fragment register feeds 8 MFMAs of 16 KFLOP); v_mfma_f32_32x32x16_bf16 does the same FLOPs per 128x128x64 wave tile with HALF the operand-register reads
// (each fragment feeds 4 MFMAs of 32 KFLOP).  Bare MFMA loops (no memory), one wave per SIMD, 256 accumulator registers, random bf16 operands; ~3 s each; the caller samples
// rocm-smi beside it.   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_power.hip -o tools/micro/mfma_power.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
#include <chrono>
#include <thread>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ bf16x8_t frag(const uint4* src, int idx) { const uint4 v = src[idx]; return *(const bf16x8_t*)&v; }

template <int MODE>      // 0: 16x16x32, 8 x 8 tiles of 4 registers; 1: 32x32x16, 4 x 4 tiles of 16 registers; 2: 16x16x32 with the B fragments walked back and forth (the MFMA at a row change keeps its B operand); 3: 16x16x32 column-major (B fragment fixed over 8 MFMAs, A changes)
__global__ __launch_bounds__(256) void mfma_loop(const uint4* __restrict__ src, float* __restrict__ out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float sink = 0.f;
  if (MODE == 0 || MODE == 2 || MODE == 3) {
    f32x4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t a[2][8], b[2][8];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[k][i] = frag(src, ((wave * 2 + k) * 8 + i) * 64 + lane); b[k][i] = frag(src, 4096 + ((wave * 2 + k) * 8 + i) * 64 + lane); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            if (MODE == 0) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k][i], b[k][j], acc[i][j], 0, 0, 0);
            else if (MODE == 2) { const int jj = (i & 1) ? 7 - j : j; acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k][i], b[k][jj], acc[i][jj], 0, 0, 0); }
            else acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[k][j], b[k][i], acc[j][i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) sink += acc[i][j][0] + acc[i][j][3];
  } else {
    f32x16_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    bf16x8_t a[4][4], b[4][4];      // [k16 step][fragment]
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[k][i] = frag(src, ((wave * 4 + k) * 4 + i) * 64 + lane); b[k][i] = frag(src, 4096 + ((wave * 4 + k) * 4 + i) * 64 + lane); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k][i], b[k][j], acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) sink += acc[i][j][0] + acc[i][j][15];
  }
  if (sink == 123.456f) out[threadIdx.x] = sink;
}

int main(int argc, char** argv) {
  const int zero = argc > 1 ? atoi(argv[1]) : 0;      // 1: zero operands (the guide's DVFS observation: +19 % TF/s)
  const double secs = argc > 2 ? atof(argv[2]) : 3.0;
  std::vector<uint16_t> h(8192 * 64 * 8);
  uint32_t s = 12345;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; const float f = ((int)(s >> 9) % 2001 - 1000) * 1e-3f; uint32_t u; memcpy(&u, &f, 4); v = zero ? 0 : (uint16_t)(u >> 16); }
  uint4* d; float* o;
  hipMalloc(&d, h.size() * 2); hipMalloc(&o, 4096);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount;
  for (int mode = 0; mode < 4; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      const int iters = 20000;                                        // 128 (64) MFMAs = 2 * 128 * 128 * 64 FLOP per wave and iteration
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      int launches = 0;
      const auto t0 = std::chrono::steady_clock::now();
      hipEventRecord(e0);
      while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
        for (int q = 0; q < 4; ++q) {
          if (mode == 0) hipLaunchKernelGGL(mfma_loop<0>, dim3(cus), dim3(256), 0, 0, d, o, iters);
          else if (mode == 1) hipLaunchKernelGGL(mfma_loop<1>, dim3(cus), dim3(256), 0, 0, d, o, iters);
          else if (mode == 2) hipLaunchKernelGGL(mfma_loop<2>, dim3(cus), dim3(256), 0, 0, d, o, iters);
          else hipLaunchKernelGGL(mfma_loop<3>, dim3(cus), dim3(256), 0, 0, d, o, iters);
          ++launches;
        }
        hipDeviceSynchronize();
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double fl = (double)launches * cus * 4 * iters * 2.0 * 128 * 128 * 64;
      printf("mode %d (%s) %s operands rep %d: %.1f TF/s over %.2f s (%d launches)\n", mode, mode == 1 ? "32x32x16" : mode == 2 ? "16x16x32 serpentine" : mode == 3 ? "16x16x32 column-major" : "16x16x32", zero ? "zero" : "random", rep, fl / (ms * 1e-3) / 1e12, ms * 1e-3, launches);
      fflush(stdout);
    }
  }
  return 0;
}
