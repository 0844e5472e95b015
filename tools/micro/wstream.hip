// Micro-benchmark (diagnostic, not part of the library): how fast can a [N, K] bf16 weight matrix be STREAMED from HBM by 16-byte-per-lane loads,
// as a function of the shape of one wave's request stream?  (The decode GEMMs for <= 64 tokens are pure weight streams; their kernels reach
// 2.2-4.9 TB/s depending on geometry, a linear copy 5-6 TB/s.)
//   RPI : rows one load instruction covers (64 lanes x 16 B = RPI rows x 1024/RPI contiguous bytes)
//   RPW : rows a wave owns (RPW / RPI instructions per k-step); a workgroup = 4 waves on consecutive row groups
//   KS  : K-slices (the row's K range is cut into KS contiguous slices, one wave each)
//   U   : k-steps in flight per wave (each k-step = RPW / RPI loads)
//   hipcc --offload-arch=gfx950 -O3 -o wstream.bin wstream.hip && ./wstream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int RPI, int RPW, int U, bool NT, bool MF = false>
__global__ __launch_bounds__(256) void k(const char* w, int N, int Kb, int KS, unsigned* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int IPS = RPW / RPI;                       // instructions per k-step
  constexpr int RUN = 1024 / RPI;                      // contiguous bytes per row and instruction
  const int gw = blockIdx.x * 4 + wave;                // global wave id
  const int groups = N / RPW;
  const int grp = gw % groups, ks = gw / groups;
  if (ks >= KS) return;
  const int slice = Kb / KS;                           // bytes per row and slice
  const int steps = slice / RUN;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, (int)0xffffffffu, 0x00020000);
  unsigned voff[IPS];
#pragma unroll
  for (int i = 0; i < IPS; ++i) {
    // MF (RPI = 8 only): the MFMA-compatible lane map of the planned decode kernel - lane l fetches row l & 7, chunk (l >> 4) + 4 * bit 3
    // (first instruction of a 16-row pair) or the other half (second): a quad of lanes spans four rows, the instruction still covers 8 whole lines
    const int row = grp * RPW + i * RPI + (MF ? (lane & 7) : lane / (64 / RPI));
    const int chunk = MF ? (lane >> 4) + 4 * (((lane >> 3) & 1) ^ (i & 1)) : lane % (64 / RPI);
    voff[i] = (unsigned)row * (unsigned)Kb + (unsigned)ks * (unsigned)slice + chunk * 16u;
  }
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

template <int RPI, int RPW, int U, bool NT, bool MF = false>
void run(const char* name, const char* w, size_t copies, int N, int K, int KS, unsigned* sink) {
  const int Kb = K * 2;
  if (Kb / KS / (1024 / RPI) < U || (Kb / KS) % (1024 / RPI)) return;
  const int waves = N / RPW * KS, blocks = (waves + 3) / 4;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const size_t stride = (size_t)N * Kb;
  for (size_t c = 0; c < copies; ++c) hipLaunchKernelGGL((k<RPI, RPW, U, NT, MF>), dim3(blocks), dim3(256), 0, 0, w + c * stride, N, Kb, KS, sink);
  hipEventRecord(a);
  const int reps = 3;
  for (int rep = 0; rep < reps; ++rep)
    for (size_t c = 0; c < copies; ++c) hipLaunchKernelGGL((k<RPI, RPW, U, NT, MF>), dim3(blocks), dim3(256), 0, 0, w + c * stride, N, Kb, KS, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double us = ms * 1e3 / (reps * copies);
  printf("%-58s N %5d K %5d KS %2d  waves %5d  %7.1f us  %6.0f GB/s\n", name, N, K, KS, waves, us, (double)stride / us / 1e3);
}

int main() {
  const size_t total = 1536ull << 20;                  // weights rotate over 1.5 GB: the 256-MiB Infinity Cache cannot hold them
  char* w; unsigned* sink;
  hipMalloc(&w, total); hipMemset(w, 1, total); hipMalloc(&sink, 64);
  struct Sh { const char* n; int N, K; } shapes[] = {{"qkv", 12288, 4096}, {"gate_up", 22016, 4096}, {"down", 4096, 11008}, {"o", 4096, 4096}, {"lm_head", 32000, 4096}};
  for (auto& sh : shapes) {
    const size_t copies = total / ((size_t)sh.N * sh.K * 2);
    printf("== %s\n", sh.n);
    const int KSs[3] = {1, 4, 8};
    for (int ks : KSs) {
      run<16, 16, 8, true>("16 rows x 64 B per instr, 16 rows/wave, U=8", w, copies, sh.N, sh.K, ks, sink);
      run<8, 16, 8, true>("8 rows x 128 B, 16 rows/wave, U=8", w, copies, sh.N, sh.K, ks, sink);
      run<8, 16, 8, false>("8 rows x 128 B, 16 rows/wave, U=8, default policy", w, copies, sh.N, sh.K, ks, sink);
      run<8, 16, 8, true, true>("8 rows x 128 B, 16 rows/wave, U=8, MFMA lane map", w, copies, sh.N, sh.K, ks, sink);
      run<8, 16, 4, true, true>("8 rows x 128 B, 16 rows/wave, U=4, MFMA lane map", w, copies, sh.N, sh.K, ks, sink);
      run<8, 32, 4, true, true>("8 rows x 128 B, 32 rows/wave, U=4, MFMA lane map", w, copies, sh.N, sh.K, ks, sink);
    }
  }
  return 0;
}
