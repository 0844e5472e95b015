// Common device helpers for the gfx950 (MI355X / CDNA4) kernels of libopadpo_hip.so.
// Wave = 64 lanes, MFMA fragments as documented in DESIGN.md §kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 storage (torch.bfloat16 bit pattern)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

__device__ __forceinline__ float bf2f(bf16_t x) { return __uint_as_float(((uint32_t)x) << 16); }

// fp32 -> bf16, round-to-nearest-even, NaN preserved (same as torch .to(bfloat16)): gfx950 has the conversion in hardware
// (v_cvt_pk_bf16_f32, two values per instruction) — a software rounding sequence costs ~6 VALU per value.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {
  const bf16x2_t r = __builtin_convertvector(f32x2_t{a, b}, bf16x2_t);
  return *(const uint32_t*)&r;
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

// bare v_exp_f32 (2^x; exp2f() wraps it in denormal range scaling: +3 VALU per call)
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack_bf2(f[0], f[1]); v.y = pack_bf2(f[2], f[3]);
  v.z = pack_bf2(f[4], f[5]); v.w = pack_bf2(f[6], f[7]);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves); `red` is >= 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max_256(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// XCD-aware bijective block remap (8 XCDs, block b lands on XCD b % 8): gives every XCD a
// contiguous chunk of the logical tile order so neighbouring tiles share its private L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, idx = bid >> 3;
  const int q = nwg >> 3, r = nwg & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + idx;
}

// The same blocks of 32 consecutive logical tiles per XCD and round (one XCD's 32 workgroups keep their 8 x 4 footprint in its L2), but dealt to the XCDs
// block-CYCLICALLY: in round k XCD x takes block 8 k + x, so at any time the eight XCDs work on neighbouring column blocks of the SAME row group - its A panels
// are fetched from HBM once for the chip (Infinity-Cache hits for seven XCDs) and the B matrix stays within the cache's reach between two passes, where the
// contiguous-chunk deal above has eight row groups' A panels plus B in flight.  Assumes 256 resident workgroups per round (block b on XCD b % 8); the last,
// partial round falls back to the contiguous deal.
__device__ __forceinline__ int xcd_remap_cyclic(int bid, int nwg) {
  const int full = nwg & ~255;
  if (bid >= full) return full + xcd_remap(bid - full, nwg - full);
  return (bid & ~255) + ((bid & 7) << 5) + ((bid >> 3) & 31);
}

// Read one MFMA 16x16x32 operand fragment whose contraction index runs along the ROWS of a
// row-major LDS tile (row stride `ld_bytes`): lane (c = lane&15, g = lane>>4) receives
// tile[k0 + g*8 + j][c0 + c], j = 0..7.  TR=true uses the gfx950 transpose read
// (ds_read_b64_tr_b16: inside a 16-lane group lane a supplies the 8-byte address of row a>>2,
// column chunk a&3; lane c gets column c of that 4x16 block); TR=false gathers 8 scalars.
template <bool TR>
__device__ __forceinline__ bf16x8_t lds_frag_rows(const char* tile, int ld_bytes, int k0, int c0, int lane) {
  union { bf16x8_t v; s16x4_t h[2]; uint16_t s[8]; } u;
  const int g = lane >> 4, c = lane & 15;
  if constexpr (TR) {
    const char* p = tile + (size_t)(k0 + g * 8 + (c >> 2)) * ld_bytes + (c0 + (c & 3) * 4) * 2;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, p + 4 * ld_bytes));
  } else {
    const char* p = tile + (size_t)(k0 + g * 8) * ld_bytes + (c0 + c) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) u.s[j] = *(const uint16_t*)(p + j * ld_bytes);
  }
  return u.v;
}
