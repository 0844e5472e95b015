// Micro-benchmark (diagnostic, not part of the library): does filling the HBM-idle window of a tiny kernel (the decode step's RMSNorm: 8-64
// workgroups, ~4.8 us, moves nothing) with PREFETCH workgroups that touch the head of the NEXT launch's weight stream shorten the pair
// (tiny kernel + weight-stream kernel)?  The weight matrix of a decode GEMM does not depend on the activations, so its first bytes can be pulled
// into the XCD's L2 / the Infinity Cache while the queue would otherwise idle on a row-wise kernel and a launch boundary.
//   chain per "layer":  norm(rows, + PF workgroups over the first FRAC of every weight row)  ->  stream(whole matrix)
//   FRAC = 0 is the baseline (no prefetch workgroups); SHIFT = 1 prefetches for the consumer workgroup of the NEXT XCD (Infinity Cache only).
//   hipcc --offload-arch=gfx950 -O3 -o prefetch.bin prefetch.hip && ./prefetch.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

// the consumer: 8 rows x 128 B per instruction, 16 rows per wave, 8 k-steps in flight (the best pure stream of wstream.hip), 64 rows per workgroup
template <bool NT>
__global__ __launch_bounds__(256) void stream_k(const char* w, int N, int Kb, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int U = 8, IPS = 2, RUN = 128;
  const int gw = blockIdx.x * 4 + wave;
  if (gw * 16 >= N) return;
  const int steps = Kb / RUN;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)0xffffffffu, 0x00020000);
  unsigned voff[IPS];
#pragma unroll
  for (int i = 0; i < IPS; ++i) voff[i] = (unsigned)(gw * 16 + i * 8 + lane / 8) * (unsigned)Kb + (lane % 8) * 16u;
  u32x4_t acc = {0, 0, 0, 0};
  u32x4_t buf[U][IPS];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int i = 0; i < IPS; ++i) buf[u][i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff[i], (u < steps ? u : 0) * RUN, NT ? 2 : 0);
  for (int s = 0; s < steps; s += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int i = 0; i < IPS; ++i) acc ^= buf[u][i];
      __builtin_amdgcn_sched_barrier(0);
      const int nx = s + u + U;
#pragma unroll
      for (int i = 0; i < IPS; ++i) buf[u][i] = __builtin_amdgcn_raw_buffer_load_b128(r, voff[i], (nx < steps ? nx : 0) * RUN, NT ? 2 : 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

// the tiny kernel: `rows` workgroups do an RMSNorm of one fp32 row of H each; workgroups rows .. rows + pf_blocks - 1 prefetch: workgroup p touches
// the first pf_bytes of the 64 weight rows of consumer workgroups (p + shift) % cblocks, + pf_blocks, .. (same XCD as the consumer when shift = 0:
// rows and pf_blocks are multiples of 8)
__global__ __launch_bounds__(256) void norm_pf_k(const float* x, unsigned short* y, int rows, int H, const char* w, int Kb, int cblocks, int pf_blocks,
                                                 int pf_bytes, int shift, unsigned* sink) {
  __shared__ float red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x < rows) {
    const float* xr = x + (size_t)blockIdx.x * H;
    float s = 0.f;
    for (int i = tid * 4; i < H; i += 1024) { const float4 v = *(const float4*)(xr + i); s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave] = s;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / H + 1e-5f);
    for (int i = tid * 4; i < H; i += 1024) {
      const float4 v = *(const float4*)(xr + i);
      unsigned short* o = y + (size_t)blockIdx.x * H + i;
      o[0] = (unsigned short)(__float_as_uint(v.x * rstd) >> 16); o[1] = (unsigned short)(__float_as_uint(v.y * rstd) >> 16);
      o[2] = (unsigned short)(__float_as_uint(v.z * rstd) >> 16); o[3] = (unsigned short)(__float_as_uint(v.w * rstd) >> 16);
    }
    return;
  }
  const int p = (int)blockIdx.x - rows;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)0xffffffffu, 0x00020000);
  u32x4_t acc = {0, 0, 0, 0};
  const int steps = pf_bytes / 128;
  for (int cb = p; cb < cblocks; cb += pf_blocks) {
    const int c = (cb + shift) % cblocks;
    // wave `wave` covers the consumer wave's 16 rows: two instructions of 8 rows x 128 B per 128-byte k-step
    const unsigned v0 = (unsigned)(c * 64 + wave * 16 + lane / 8) * (unsigned)Kb + (lane % 8) * 16u, v1 = v0 + 8u * (unsigned)Kb;
    for (int s = 0; s < steps; s += 8) {
      u32x4_t b[16];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        b[2 * u] = __builtin_amdgcn_raw_buffer_load_b128(r, v0, (s + u < steps ? s + u : 0) * 128, 0);
        b[2 * u + 1] = __builtin_amdgcn_raw_buffer_load_b128(r, v1, (s + u < steps ? s + u : 0) * 128, 0);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) acc ^= b[u];
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <bool NT>
static void run(const char* name, const char* w, size_t total, int N, int K, int rows, int pf_bytes, int shift, const float* x, unsigned short* y, unsigned* sink) {
  const int Kb = K * 2, H = 4096;
  const size_t stride = (size_t)N * Kb, copies = total / stride;
  const int cblocks = N / 64;
  const int pf_blocks = pf_bytes > 0 ? (cblocks < 248 ? cblocks : 248) : 0;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto chain = [&](size_t c) {
    hipLaunchKernelGGL(norm_pf_k, dim3(rows + pf_blocks), dim3(256), 0, 0, x, y, rows, H, w + c * stride, Kb, cblocks, pf_blocks, pf_bytes, shift, sink);
    hipLaunchKernelGGL((stream_k<NT>), dim3(cblocks), dim3(256), 0, 0, w + c * stride, N, Kb, sink);
  };
  for (size_t c = 0; c < copies; ++c) chain(c);
  hipEventRecord(a);
  const int reps = 3;
  for (int rep = 0; rep < reps; ++rep)
    for (size_t c = 0; c < copies; ++c) chain(c);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / (reps * copies);
  printf("%-8s N %5d K %5d rows %2d nt %d  prefetch %5d B/row (%4.1f %% = %5.1f MB, %3d wgs, shift %d)  norm+stream %7.2f us  (%5.0f GB/s of weights)\n", name, N, K, rows, (int)NT,
         pf_bytes, 100.0 * pf_bytes / Kb, (double)pf_bytes * N / 1e6, pf_blocks, shift, us, (double)stride / us / 1e3);
  hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
  const size_t total = 1536ull << 20;                  // weights rotate over 1.5 GB: the 256-MiB Infinity Cache cannot hold them
  char* w; unsigned* sink; float* x; unsigned short* y;
  hipMalloc(&w, total); hipMemset(w, 1, total); hipMalloc(&sink, 64);
  hipMalloc(&x, 64 * 4096 * 4); hipMemset(x, 0, 64 * 4096 * 4); hipMalloc(&y, 64 * 4096 * 2);
  struct Sh { const char* n; int N, K; } shapes[] = {{"qkv", 12288, 4096}, {"gate_up", 22016, 4096}, {"o", 4096, 4096}, {"down", 4096, 11008}};
  for (int rows : {8, 64})
    for (auto& sh : shapes) {
      const int Kb = sh.K * 2;
      for (int pf : {0, Kb / 16, Kb / 8, Kb / 4, Kb * 3 / 8, Kb / 2}) {
        const int pfb = pf / 1024 * 1024;                 // whole 8-step groups of 128 B
        if (pf && !pfb) continue;
        run<true>(sh.n, w, total, sh.N, sh.K, rows, pfb, 0, x, y, sink);
        if (pfb && pf == Kb / 4) {
          run<true>(sh.n, w, total, sh.N, sh.K, rows, pfb, 1, x, y, sink);
          run<false>(sh.n, w, total, sh.N, sh.K, rows, pfb, 0, x, y, sink);
        }
        if (!pf) run<false>(sh.n, w, total, sh.N, sh.K, rows, 0, 0, x, y, sink);
      }
    }
  return 0;
}
