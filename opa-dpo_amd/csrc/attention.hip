// Flash attention forward / backward for gfx950 (bf16 in, fp32 softmax + accumulation).
//
// Forward ("swapped" formulation, everything keyed on q = lane & 15):
//   S^T[key][q] = K . Q^T         (MFMA A = K fragment from LDS, B = Q fragment in registers)
//   online softmax over keys: in-lane over 16 values + 2 cross-lane-group shuffles
//   O^T[d][q]  += V^T . P^T       (A = V fragment through ds_read_b64_tr_b16, B = P straight
//                                  from the S accumulators — no LDS round trip for P)
// Block = 4 waves x 16 q rows = 64 q rows of one (sequence, head); KV tiles of 64 keys.
// Masking: causal and/or an arbitrary per-key byte mask [S,L] (left-padded queries,
// right-padded responses, CoPO 'attention' image-key dropping).
//
// Backward is two atomic-free kernels (the fp32-atomic dQ of a single KV-outer kernel was the
// bottleneck on MI355X):
//   dkdv : block = one KV tile (64 keys), wave w owns keys w*16..+15 (dK/dV in registers),
//          loops over q tiles:  S, dP (contraction over d) ; dV += dO^T P, dK += Q^T dS
//          (contraction over q through transposed LDS reads).
//   dq   : block = 64 q rows (forward geometry), loops over KV tiles, recomputes S^T and dP^T in
//          the swapped layout so that dS^T feeds  dQ^T[d][q] += K^T . dS^T  straight from registers.
//
// LDS tiles are [64][HD] bf16 with a 16-byte-chunk XOR swizzle that is conflict-free for BOTH
// access patterns (row fragments via ds_read_b128, transposed fragments via ds_read_b64_tr_b16):
//   HD=128 (256-B rows): chunk ^= (row & 7) << 1      HD=64 (128-B rows): chunk ^= ((row >> 1) & 3) << 1
#include <stdlib.h>
#include <type_traits>
#include "common.h"
#include "kernels.h"

namespace {

constexpr float NEG_BIG = -1.0e30f;

template <int HD>
__device__ __forceinline__ int swz_mask(int row) {
  return HD == 128 ? ((row & 7) << 1) : (((row >> 1) & 3) << 1);
}

// address of element (row, col) [col in bf16 elements] inside a swizzled [64][HD] tile
template <int HD>
__device__ __forceinline__ const char* tile_at(const char* tile, int row, int col) {
  const int b = col * 2;
  return tile + row * (HD * 2) + ((((b >> 4) ^ swz_mask<HD>(row))) << 4) + (b & 15);
}

// operand fragment whose contraction index runs along d (row-contiguous): 8 bf16 of row `row`
// starting at d = kk*32 + g*8
template <int HD>
__device__ __forceinline__ bf16x8_t frag_row(const char* tile, int row, int kk, int g) {
  return *(const bf16x8_t*)(tile + row * (HD * 2) + (((kk * 4 + g) ^ swz_mask<HD>(row)) << 4));
}

// operand fragment whose contraction index runs along the tile ROWS: lane (c = lane&15) gets
// tile[r0 + j][c0 + c] (slots 0..3) and tile[r1 + j][c0 + c] (slots 4..7); r0/r1 include the
// lane-group term.  TR: gfx950 transpose read (inside a 16-lane group lane a supplies the 8-byte
// address of row a>>2, column chunk a&3; lane c receives column c of that 4x16 block).
template <int HD, bool TR>
__device__ __forceinline__ bf16x8_t frag_col(const char* tile, int r0, int r1, int c0, int lane) {
  union { bf16x8_t v; s16x4_t h[2]; uint16_t s[8]; } u;
  const int c = lane & 15;
  if constexpr (TR) {
    const int col = c0 + (c & 3) * 4;
    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, tile_at<HD>(tile, r0 + (c >> 2), col)));
    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, tile_at<HD>(tile, r1 + (c >> 2), col)));
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      u.s[j] = *(const uint16_t*)tile_at<HD>(tile, r0 + j, c0 + c);
      u.s[4 + j] = *(const uint16_t*)tile_at<HD>(tile, r1 + j, c0 + c);
    }
  }
  return u.v;
}

__device__ __forceinline__ bf16x8_t pack_frag(const f32x4_t& a, const f32x4_t& b) {
  union { bf16x8_t v; uint32_t w[4]; } u;
  u.w[0] = pack_bf2(a[0], a[1]); u.w[1] = pack_bf2(a[2], a[3]);
  u.w[2] = pack_bf2(b[0], b[1]); u.w[3] = pack_bf2(b[2], b[3]);
  return u.v;
}

// Inverse rotary embedding of a gradient row held in the 16x16 accumulator layout (lane holds d = df*16 + g*4 + r, df = 0 .. HD/16 - 1):
// the pair (d, d + HD/2) sits in the SAME lane (df, df + HD/32), so the rotation is lane-local.  g' = g*cos - rot(g)*sin with
// rot(x) = [-x2, x1]  ->  o1 = x1*c + x2*s, o2 = x2*c - x1*s.  Angles: hardware sin / cos of the fractional revolution
// pos * theta^(-2i/HD) / 2pi, once per row (the forward's table-free epilogue computes the same angles).
template <int HD>
__device__ __forceinline__ void rope_inverse_rows(f32x4_t (&acc)[HD / 16], int g, int pos, float l2theta) {
  constexpr int HF = HD / 32;
  const float posf = (float)pos;
#pragma unroll
  for (int df = 0; df < HF; ++df)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = df * 16 + g * 4 + r;                                        // frequency index d % (HD/2)
      const float frev = __builtin_amdgcn_exp2f(-(float)(2 * i) * (1.0f / HD) * l2theta) * 0.15915494309189535f;
      const float x = __builtin_amdgcn_fractf(posf * frev);
      const float c = __builtin_amdgcn_cosf(x), sn = __builtin_amdgcn_sinf(x);
      const float x1 = acc[df][r], x2 = acc[df + HF][r];
      acc[df][r] = x1 * c + x2 * sn;
      acc[df + HF][r] = x2 * c - x1 * sn;
    }
}

// Global side of a [64][HD] tile stream: one buffer descriptor per (operand, sequence) whose extent is exactly the L rows
// of the sequence, so rows past L read as ZERO in hardware (no per-lane predicate, no exec juggling) and the per-tile
// offset lives in the scalar soffset: a tile fetch is HD/32 buffer_load_dwordx4 with ONE precomputed VGPR offset.
template <int HD>
struct TileSrc {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff;      // bytes: (row tid / CPR) * ld + head * HD + (tid % CPR) * 8 elements
  int row_step;  // bytes between the row groups of consecutive passes
  __device__ __forceinline__ TileSrc(const bf16_t* src, int ld, size_t row0, int L, int head, int tid) {
    constexpr int CPR = HD / 8;
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + row0 * ld), 0, L * ld * 2, 0x00020000);
    voff = ((tid / CPR) * ld + head * HD + (tid % CPR) * 8) * 2;
    row_step = (256 / CPR) * ld * 2;
  }
};
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

// split staging (issue-early / write-late): global -> registers while the previous tile is being consumed,
// registers -> swizzled LDS after the barrier that retires it.
template <int HD>
struct TileRegs { u32x4_t v[(64 * (HD / 8)) / 256]; };

template <int HD>
__device__ __forceinline__ void tile_fetch(TileRegs<HD>& r, const TileSrc<HD>& ts, int ld, int pos0) {
  constexpr int CPR = HD / 8;
  const int base = pos0 * ld * 2;
#pragma unroll
  for (int i = 0; i < (64 * CPR) / 256; ++i)
    r.v[i] = __builtin_amdgcn_raw_buffer_load_b128(ts.rsrc, ts.voff, base + i * ts.row_step, 0);
}
template <int HD>
__device__ __forceinline__ void tile_commit(char* dst, const TileRegs<HD>& r, int tid) {
  constexpr int CPR = HD / 8;
#pragma unroll
  for (int i = 0; i < (64 * CPR) / 256; ++i) {
    const int idx = tid + i * 256;
    const int row = idx / CPR, c16 = idx % CPR;
    *(u32x4_t*)(dst + row * (HD * 2) + ((c16 ^ swz_mask<HD>(row)) << 4)) = r.v[i];
  }
}
// stage a [64][HD] tile (rows = positions pos0.. of the sequence) into swizzled LDS, zero past L
template <int HD>
__device__ __forceinline__ void stage_tile(char* dst, const TileSrc<HD>& ts, int ld, int pos0, int tid) {
  TileRegs<HD> r;
  tile_fetch<HD>(r, ts, ld, pos0);
  tile_commit<HD>(dst, r, tid);
}

// Direct-to-LDS staging (buffer_load_dwordx4 ... lds): no registers, no ds_write, completion tracked by vmcnt.  The DMA
// writes lane-linear (wave instruction j of wave w fills the 1 KiB piece (j*4 + w) of the tile = 64/CPR consecutive rows), so
// the tile's XOR swizzle is applied on the SOURCE side: lane (row r, LDS chunk c') fetches global chunk c' ^ swz(r).  swz only
// depends on r mod 8, which is the same for every j -> ONE per-lane offset, the row group / tile position go in soffset.
template <int HD>
struct TileDma {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff, row_step, w;
  __device__ __forceinline__ TileDma(const bf16_t* src, int ld, size_t row0, int L, int head, int tid) {
    constexpr int CPR = HD / 8, RPI = 64 / CPR;     // 16-B chunks per row, rows per wave instruction
    const int lane = tid & 63;
    w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = w * RPI + lane / CPR;             // row inside the 4*RPI-row group of one pass
    const int c = (lane % CPR) ^ swz_mask<HD>(r);
    rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(src + row0 * ld), 0, L * ld * 2, 0x00020000);
    voff = (r * ld + head * HD + c * 8) * 2;
    row_step = 4 * RPI * ld * 2;
  }
  __device__ __forceinline__ void issue(char* tile, int ld, int pos0) const {
    constexpr int CPR = HD / 8, RPI = 64 / CPR;
    const int base = pos0 * ld * 2;
#pragma unroll
    for (int j = 0; j < 64 / (4 * RPI); ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_PTR(void, tile + (j * 4 + w) * 1024), 16, voff, base + j * row_step, 0, 0);
  }
};

// key-mask bytes of one K/V tile -> Ms[0..63]; Ms[64] = 1 when any key of the tile is masked or past L (block-uniform)
__device__ __forceinline__ void stage_mask(uint8_t* Ms, const uint8_t* key_mask, size_t row0, int L, int k0, int tid) {
  if (tid < 64) {
    const int kp = k0 + tid;
    const uint8_t m = (kp < L) ? (key_mask ? key_mask[row0 + kp] : (uint8_t)1) : (uint8_t)0;
    Ms[tid] = m;
    const uint64_t dead = __ballot(m == 0);
    if (tid == 0) { Ms[64] = dead != 0; *(uint64_t*)(Ms + 72) = ~dead; }      // Ms[72..79]: bit j = key j of the tile is a valid key
  }
}

// ---- packed responses with a shared prefix (seg_len > 0) ------------------------------------------------------------
// A row holds [prefix (seg_prefix positions: image + query) | response 0 | response 1 | ...], every response seg_len long.
// Response a attends the prefix and itself, never another response: for a query in segment a >= 1 the keys in
// [seg_prefix, start of segment a) are excluded on top of the causal / key masks.  Causality makes the prefix states
// independent of the responses, so this equals running prefix+response_a as separate sequences (what the reference does,
// rl_models.py:95-112) while computing the prefix ONCE.  K/V tiles that lie wholly inside the excluded range are skipped.
// Geometry of one sequence.  Padded layout: S rows of p.L positions, responses of p.seg_len positions after p.seg_prefix.  RAGGED
// layout (p.seq_meta != NULL; padding rows removed from every row-wise operator of the model): sequence s occupies rows
// [meta[0], meta[0] + L_s) of the flat [rows, ld] buffers and meta[1 .. 1 + n_seg] are its segment boundaries b_0 (= end of the
// prefix) .. b_nseg (= L_s), all relative to the sequence start - prefix and response lengths differ per sequence.
struct Geo {
  size_t row0;
  int L;
  const int32_t* b;          // ragged: boundaries b[0..nseg]; padded: nullptr
  int nseg, pfx, slen;
};
__device__ __forceinline__ Geo load_geo(const AttnArgs& p, int s) {
  Geo g;
  if (p.seq_meta) {
    const int32_t* m = p.seq_meta + (size_t)s * p.meta_stride;
    g.row0 = (size_t)m[0]; g.b = m + 1; g.nseg = p.n_seg; g.L = m[1 + p.n_seg]; g.pfx = m[1]; g.slen = 0;
  } else {
    g.row0 = (size_t)s * p.L; g.L = p.L; g.b = nullptr; g.nseg = p.seg_len > 0 ? 1 : 0; g.pfx = p.seg_prefix; g.slen = p.seg_len;
  }
  return g;
}
__device__ __forceinline__ bool seg_on(const Geo& g) { return g.b ? g.nseg > 0 : g.slen > 0; }
// first key of the response area (keys from here to the start of the query's own segment are excluded)
__device__ __forceinline__ int seg_xlo(const Geo& g) { return seg_on(g) ? g.pfx : 0x7fffffff; }
// start of the segment position `pos` lies in (the prefix end for prefix positions and for segment 0)
__device__ __forceinline__ int seg_qstart(const Geo& g, int pos) {
  if (!seg_on(g)) return 0x7fffffff;
  if (g.b) {
    int st = g.pfx;
    for (int a = 1; a < g.nseg; ++a) if (pos >= g.b[a]) st = g.b[a];
    return st;
  }
  return pos >= g.pfx + g.slen ? g.pfx + ((pos - g.pfx) / g.slen) * g.slen : g.pfx;
}
// end of the segment position `pos` (>= prefix end) lies in
__device__ __forceinline__ int seg_end(const Geo& g, int pos) {
  if (g.b) {
    int e = g.b[g.nseg];
    for (int a = g.nseg - 1; a >= 1; --a) if (pos < g.b[a]) e = g.b[a];
    return e;
  }
  return g.pfx + ((pos - g.pfx) / g.slen + 1) * g.slen;
}
struct SegSkip {       // block-uniform: K/V tiles [lo, hi) are excluded for every row of the q tile starting at q0
  int lo, hi;
  __device__ __forceinline__ SegSkip(const Geo& g, int q0, int n_kt) : lo(n_kt), hi(n_kt) {
    if (seg_on(g)) {
      const int qs = seg_qstart(g, q0);
      if (qs > g.pfx) {
        const int l = (g.pfx + 63) / 64, h = qs / 64;
        if (h > l) { lo = l; hi = h; }
      }
    }
  }
  __device__ __forceinline__ int first() const { return lo == 0 ? hi : 0; }
  __device__ __forceinline__ int next(int kt) const { const int n = kt + 1; return (n >= lo && n < hi) ? hi : n; }
};
// lse / delta element of (sequence s, head h, position pos): padded [S, nh, L]; ragged [nh, rows_total] (row-major over the flat rows)
__device__ __forceinline__ size_t stat_idx(const AttnArgs& p, const Geo& g, int s, int h, int pos) {
  return p.seq_meta ? (size_t)h * p.rows_total + g.row0 + pos : ((size_t)s * p.nh + h) * p.L + pos;
}

// ------------------------------------------------------------------------------------------------
// DMA: K/V tiles land in a double-buffered LDS ring through direct-to-LDS loads, one barrier per tile (the tile of the
// next iteration is in flight while this one is consumed); !DMA: register-staged single buffer, two barriers per tile.
// 1-D grid -> (tile, head, sequence).  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): with the plain
// (tile, head, sequence) order the tiles of one (sequence, head) land on all 8 XCDs and its K / V (or Q / dO) stream is pulled into
// 8 L2s.  Here the j-th block of an XCD walks the tiles of ONE (sequence, head) pair before the next pair: every pair is read by
// one L2.  Needs nh * S to be a multiple of 8 (else the plain order).
__device__ __forceinline__ void attn_block_map(const AttnArgs& p, int n_tiles, int& tile, int& h, int& s) {
  const int b = blockIdx.x, pairs = p.nh * p.S;
  int pair, t;
  if (pairs % 8 == 0) {
    const int xcd = b & 7, j = b >> 3;
    pair = xcd + 8 * (j / n_tiles);
    t = j % n_tiles;
  } else {
    pair = b / n_tiles;
    t = b % n_tiles;
  }
  tile = t; h = pair % p.nh; s = pair / p.nh;
}

template <int HD, bool TR, bool DMA>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TILE = 64 * HD * 2;
  uint8_t* const ms_base = (uint8_t*)(smem + (DMA ? 4 : 2) * TILE);
  constexpr int KK = HD / 32, DF = HD / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n_qt = (p.L + 63) / 64;
  int s, h, qi;
  attn_block_map(p, n_qt, qi, h, s);
  const int qt = n_qt - 1 - qi;   // heavier (later) causal tiles first
  const int q0 = qt * 64;
  const Geo ge = load_geo(p, s);
  const int L = ge.L;
  if (q0 >= L) return;                      // ragged rows: this sequence is shorter than the longest one
  const int qpos = q0 + w * 16 + c;
  const int qrow = min(qpos, L - 1);
  // causal bit 1 (OPADPO_ATTN_SKIP_MASKED_Q): a q tile whose 64 positions are ALL masked as keys is padding (the trailing pad of a
  // right-padded response, 22 % of the rows of a synthetic seq512 pair): nothing downstream reads those rows (they are masked as
  // keys, their labels are pad, their gradient is exactly zero), so the tile writes zeros and leaves.
  if ((p.causal & 2) && p.key_mask) {
    const int qp_ = q0 + lane;
    const uint8_t mq = qp_ < L ? p.key_mask[ge.row0 + qp_] : (uint8_t)0;
    if (__ballot(mq != 0) == 0) {
      if (qpos < L) {
        bf16_t* op = p.o + (ge.row0 + qpos) * p.ldo + h * HD;
#pragma unroll
        for (int d = 0; d < DF; ++d) *(uint2*)(op + d * 16 + g * 4) = make_uint2(0u, 0u);
        if (g == 0 && p.lse) p.lse[stat_idx(p, ge, s, h, qpos)] = NEG_BIG;
      }
      return;
    }
  }

  bf16x8_t qf[KK];
  {
    const bf16_t* qp = p.q + (ge.row0 + qrow) * p.ld + h * HD;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      uint4 v = *(const uint4*)(qp + kk * 32 + g * 8);
      qf[kk] = *(bf16x8_t*)&v;
    }
  }
  f32x4_t o[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) o[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run = NEG_BIG, l_run = 0.f;

  const int n_kt = p.causal ? (min(L, q0 + 64) + 63) / 64 : (L + 63) / 64;
  const SegSkip sk(ge, q0, n_kt);
  const int xlo = seg_xlo(ge), xhi = seg_qstart(ge, qpos);
  const int xhi_blk = seg_on(ge) ? seg_qstart(ge, q0 + 63) : 0;     // end of the excluded key range of the tile's LAST row
  const float scale2 = p.scale * 1.4426950408889634f;
  const TileSrc<HD> ksrc(p.k, p.ld, ge.row0, L, h, tid), vsrc(p.v, p.ld, ge.row0, L, h, tid);
  const TileDma<HD> kdma(p.k, p.ld, ge.row0, L, h, tid), vdma(p.v, p.ld, ge.row0, L, h, tid);
  TileRegs<HD> kreg, vreg;
  if constexpr (DMA) {
    kdma.issue(smem, p.ld, sk.first() * 64);
    vdma.issue(smem + TILE, p.ld, sk.first() * 64);
    stage_mask(ms_base, p.key_mask, ge.row0, L, sk.first() * 64, tid);
  } else {
    tile_fetch<HD>(kreg, ksrc, p.ld, sk.first() * 64);
    tile_fetch<HD>(vreg, vsrc, p.ld, sk.first() * 64);
  }
  int cur = 0;
  for (int kt = sk.first(), nxt; kt < n_kt; kt = nxt) {
    nxt = sk.next(kt);
    const int k0 = kt * 64;
    char* const Ks = smem + (DMA ? cur * 2 * TILE : 0);
    char* const Vs = Ks + TILE;
    uint8_t* const Ms = ms_base + (DMA ? cur * 80 : 0);
    if constexpr (DMA) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();        // tile kt has landed for everyone AND everyone is done reading the other buffer
      if (nxt < n_kt) {
        kdma.issue(smem + (cur ^ 1) * 2 * TILE, p.ld, nxt * 64);
        vdma.issue(smem + (cur ^ 1) * 2 * TILE + TILE, p.ld, nxt * 64);
        stage_mask(ms_base + (cur ^ 1) * 80, p.key_mask, ge.row0, L, nxt * 64, tid);
      }
      cur ^= 1;
    } else {
      tile_commit<HD>(Ks, kreg, tid);
      tile_commit<HD>(Vs, vreg, tid);
      stage_mask(Ms, p.key_mask, ge.row0, L, k0, tid);
      __syncthreads();
      if (nxt < n_kt) {      // next tile's global loads fly while this one is consumed
        tile_fetch<HD>(kreg, ksrc, p.ld, nxt * 64);
        tile_fetch<HD>(vreg, vsrc, p.ld, nxt * 64);
      }
    }

    f32x4_t sc[4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      sc[kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk)
        sc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<HD>(Ks, kf * 16 + c, kk, g), qf[kk], sc[kf], 0, 0, 0);
    }
    // sc[kf][r] = S^T[key = kf*16 + g*4 + r][q = c]; scores in log2 units (scale2 = scale * log2 e -> bare v_exp_f32)
    // `clean` (block-uniform): every key of the tile is visible to every row -> no mask arithmetic at all
    const bool clean = !Ms[64] && (!p.causal || k0 + 63 <= q0) && (k0 + 63 < xlo || k0 >= xhi_blk);
    float mx = NEG_BIG;
    if (clean) {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          sc[kf][r] *= scale2;
          mx = fmaxf(mx, sc[kf][r]);
        }
    } else {
      const bool nc = !p.causal;
      uint32_t mw[4];                     // the lane's 4 mask bytes per key block as one word
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) mw[kf] = *(const uint32_t*)(Ms + kf * 16 + g * 4);
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kl = kf * 16 + g * 4 + r;
          // bitwise, not short-circuit: a chain of && on per-lane values compiles to a saveexec / branch per term and element
          const bool ok = (((mw[kf] >> (8 * r)) & 0xffu) != 0) & (nc | ((k0 + kl) <= qpos)) & (((k0 + kl) < xlo) | ((k0 + kl) >= xhi));
          sc[kf][r] = ok ? sc[kf][r] * scale2 : -INFINITY;      // exp2(-inf - m) = 0 (m stays finite: NEG_BIG floor)
          mx = fmaxf(mx, sc[kf][r]);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    float psum = 0.f;
#pragma unroll
    for (int kf = 0; kf < 4; ++kf)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = fast_exp2(sc[kf][r] - m_new);
        sc[kf][r] = pv;
        psum += pv;
      }
    if (__ballot(m_new != m_run)) {        // lazy rescale: the running max settles after a few tiles
      const float alpha = fast_exp2(m_run - m_new);
      l_run *= alpha;
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        o[d][0] *= alpha; o[d][1] *= alpha; o[d][2] *= alpha; o[d][3] *= alpha;
      }
    }
    l_run += psum;
    m_run = m_new;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t pf = pack_frag(sc[2 * ks], sc[2 * ks + 1]);
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const bf16x8_t vf = frag_col<HD, TR>(Vs, (2 * ks) * 16 + g * 4, (2 * ks + 1) * 16 + g * 4, d * 16, lane);
        o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[d], 0, 0, 0);
      }
    }
    if constexpr (!DMA) __syncthreads();
  }
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  if (qpos < L) {
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    bf16_t* op = p.o + (ge.row0 + qpos) * p.ldo + h * HD;
#pragma unroll
    for (int d = 0; d < DF; ++d) {
      uint2 v;
      v.x = pack_bf2(o[d][0] * inv, o[d][1] * inv);
      v.y = pack_bf2(o[d][2] * inv, o[d][3] * inv);
      *(uint2*)(op + d * 16 + g * 4) = v;
    }
    if (g == 0 && p.lse) p.lse[stat_idx(p, ge, s, h, qpos)] = l_run > 0.f ? (m_run + log2f(l_run)) * 0.6931471805599453f : NEG_BIG;
  }
}

// ------------------------------------------------------------------------------------------------
// Forward, round 3 (head_dim 128): 32 q rows per wave on v_mfma_f32_32x32x16_bf16.
//
// The 16-row kernel above reads one 1-KiB LDS fragment per 16x16x32 MFMA (16 KFLOP): four SIMDs ask the CU's 128 B/clk LDS pipe for
// twice what it delivers before the matrix pipes are busy (PMC r02: LDS pipe 70 % busy at 16-22 % MFMA).  Here a wave owns 32 q rows:
//   S^T[key][q] = K . Q^T   A = K fragment (32 keys x 16 d, ds_read_b128 from the swizzled K tile), B = Q fragment in registers;
//                           D: lane (q = lane & 31, hi = lane >> 5) holds keys kb*32 + 8b + 4hi + (0..3), b = 0..3  (16 values per kb)
//   O^T[d][q] += V^T . P^T  A = V fragment (32 d x 16 keys) through two ds_read_b64_tr_b16 whose rows are EXACTLY the keys the lane's
//                           S accumulators hold (rows r0 = kb*32 + 16 s + 4hi .. +3 and r0 + 8 .. +3): P goes from the S registers
//                           through v_cvt_pk_bf16_f32 straight into the B operand - no cross-lane movement, no LDS round trip
// so every fragment feeds 32 KFLOP (half the LDS bytes per FLOP), the row maximum needs ONE cross-lane exchange (lane <-> lane ^ 32)
// instead of two, and scale * log2(e) is folded into the exponent's FMA.  Block = 4 waves = 128 q rows; K / V tiles of 64 keys in a
// double-buffered LDS ring (register-staged: the global loads of tile t+1 fly under the MFMAs of tile t, their ds_writes go to the
// other buffer - one barrier per tile); a wave skips the tiles none of its 32 rows can see (causal diagonal, other responses of a
// packed row) but keeps the block's barriers.  O leaves through LDS as whole 256-byte rows.
// V tile swizzle (32-byte slots): slot ^= ((row & 3) << 1) | ((row >> 2) & 1) - the two 16-lane groups of a half-wave read the SAME
// four rows at adjacent 32-byte column blocks; this keeps their eight row segments on eight different slots.
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ int vswz(int row) { return ((row & 3) << 1) | ((row >> 2) & 1); }

// Byte offset of 16-byte chunk c16 of row `row` in the 32-rows-per-wave kernels' [64][128] tiles.  Two access patterns meet here:
//  * transposing reads (ds_read_b64_tr_b16, serviced per 32-lane half): four rows x 64 contiguous bytes -> the 32-byte slot is XOR-ed with
//    vswz(row), which sends the four rows of a half to four different 64-byte segments of the 256-byte bank row;
//  * row fragments (ds_read_b128, lane = row & 31, chunk = 2 ks + (lane >> 5)): the hardware services a b128 read in the four 16-lane
//    groups {0-3,12-15,20-27} {4-11,16-19,28-31} {32-35,...} {36-43,...} (MI355X_MICROARCH.md LDS table), i.e. 16 ROWS at one chunk
//    parity per group.  A swizzle of row & 7 alone leaves only 8 distinct chunks for them - a 2-way conflict on every fragment read,
//    measured in round 3 as SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 24 % (forward) and 32 % (dQ).  In each of those groups the two rows with
//    equal row & 7 differ in bit 4 of the row, so with FLIP the chunk's low bit is XOR-ed with (row >> 4) & 1: 16 lanes -> 16 chunks.
// FLIP is used for tiles that are ONLY read as row fragments (the forward's K tile has its own kswz32 form, the dQ kernel's V tile uses
// off32<true>): there the flip is a per-lane constant.  For a tile that is also read through the transposing reads (dQ's K tile) bit 4 of
// the row is a compile-time term of the read address, every (slot, flip) pair wants its own address register, and attn_bwd_dq32_kernel -
// 254 VGPRs - starts to spill (measured: conflicts 32 % -> 0.3 %, kernel 4.5 % SLOWER); that tile keeps the plain layout.
template <bool FLIP>
__device__ __forceinline__ int off32(int row, int c16) {
  return row * 256 + (((((c16 >> 1) ^ vswz(row)) << 1) | ((FLIP ? (c16 ^ (row >> 4)) : c16) & 1)) << 4);
}

// the dQ kernel's V tile: with the flip its row-read offsets differ from the K tile's (which cannot take it, above) and the second set of
// eight address registers spills too (60 bytes of scratch per lane, measured slower) - both tiles keep ONE set of offsets, no flip
constexpr bool DQ32_VFLIP = false;
constexpr int DQ32_PF = 1;        // prefetch depth of the dQ kernel's K / V row fragments (k-steps ahead of their MFMAs)
template <bool FLIP = false>
__device__ __forceinline__ void tile_commit_v(char* dst, const TileRegs<128>& r, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * 256;
    const int row = idx >> 4, c16 = idx & 15;
    *(u32x4_t*)(dst + off32<FLIP>(row, c16)) = r.v[i];
  }
}
// K tile of the forward (row fragments only): chunk ^= ((row & 7) << 1) | ((row >> 4) & 1), conflict-free for the b128 lane groups above
__device__ __forceinline__ int kswz32(int row) { return ((row & 7) << 1) | ((row >> 4) & 1); }
__device__ __forceinline__ void tile_commit_k32(char* dst, const TileRegs<128>& r, int tid) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + i * 256;
    const int row = idx >> 4, c16 = idx & 15;
    *(u32x4_t*)(dst + row * 256 + ((c16 ^ kswz32(row)) << 4)) = r.v[i];
  }
}

#ifndef ATTN32_RESCALE_LOG2
#define ATTN32_RESCALE_LOG2 8.0f    // shipped since round 6: the reference maximum moves when some row's maximum grew by more than 2^8 (see the kernel); 0.0f = every change rescales (rounds 3-5)
#endif
#ifndef OPADPO_ATTN_ABL
#define OPADPO_ATTN_ABL 0      // ablation builds (results WRONG, timing only): 1 no max / exp / row sums, 2 no P V MFMAs, 4 no S^T MFMAs, 8 no barrier, 16 no tile staging, 32 no K fragment reads, 64 no V fragment reads
#endif                         // 128 / 256 / 512: the tile loop of the 32-row forward runs at most 0 / 1 / 4 tiles (what a workgroup costs besides its tiles)
#ifndef OPADPO_ATTN32_PRIO
#define OPADPO_ATTN32_PRIO 0     // experiment: s_setprio 1 around the S^T (bit 0) / P V (bit 1) MFMA blocks of the 32-row forward (two waves per SIMD)
#endif
#ifndef OPADPO_ATTN32_DIAG
#define OPADPO_ATTN32_DIAG 0
#endif
#if OPADPO_ATTN32_DIAG
// diagnostics build (DIAG_SRC=attention tools/build_diag.sh a32diag:"-DOPADPO_ATTN32_DIAG=1"; tools/attn32_diag.py): shader cycles of wave 0 of every workgroup of the
// 32-row forward spent [0] before its first tile barrier (geometry, Q, first K / V tile), [1] in the tile loop, [2] from the loop's end to the last store issued,
// [3] workgroups, [4] tiles walked - summed over the launches since the last read
__device__ unsigned long long g_attn32_diag[5];
extern "C" int opadpo_debug_attn32_read(unsigned long long* out5, int reset) {
  hipError_t e = hipMemcpyFromSymbol(out5, HIP_SYMBOL(g_attn32_diag), 40);
  if (reset) { unsigned long long z[5] = {0, 0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_attn32_diag), z, 40); }
  return (int)e;
}
#endif
__global__ __launch_bounds__(256, 2) void attn_fwd32_kernel(AttnArgs p) {
#if OPADPO_ATTN32_DIAG
  const unsigned long long dg_t0 = __builtin_readcyclecounter();
  unsigned long long dg_t1 = dg_t0, dg_tiles = 0;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HD = 128, TILE = 64 * HD * 2;            // 16 KiB per K or V tile; ring: [K0 | V0 | K1 | V1]
  uint8_t* const ms_base = (uint8_t*)(smem + 4 * TILE);  // 2 x 80 mask bytes
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int n_qt = (p.L + 127) / 128;
  int s, h, qi;
  attn_block_map(p, n_qt, qi, h, s);
  const int qt = n_qt - 1 - qi;                           // heavier (later) causal tiles first
  const int q0 = qt * 128;
  const Geo ge = load_geo(p, s);
  const int L = ge.L;
  if (q0 >= L) return;
  // Prologue order (round 5; profiles/r05l_attn32_anatomy.txt: a workgroup spent 7.2 k cycles before its first tile, 1.7 tiles' worth, in three
  // dependent memory round trips - padding checks, then Q, then K, then V): everything the first tile needs is REQUESTED first - the padding bytes of
  // the q rows, Q, the first K tile and the first V tile (a second 16-register image that lives in the prologue only) - and only then waited for.
  const int qlo = q0 + w * 32, qhi = min(qlo + 31, L - 1);            // this wave's rows
  const int qpos = qlo + ql;
  const int qrow = min(qpos, L - 1);
  const bool pad_check = (p.causal & 2) && p.key_mask;
  uint8_t pm0 = 1, pm1 = 1, pmq = 1;
  if (pad_check) {
    const int a_ = q0 + lane, b_ = q0 + 64 + lane;
    pm0 = a_ < L ? p.key_mask[ge.row0 + a_] : (uint8_t)0; pm1 = b_ < L ? p.key_mask[ge.row0 + b_] : (uint8_t)0;
    pmq = qpos < L ? p.key_mask[ge.row0 + qpos] : (uint8_t)0;
  }
  bf16x8_t qf[8];
  {
    const bf16_t* qp = p.q + (ge.row0 + qrow) * p.ld + h * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const uint4 v = *(const uint4*)(qp + ks * 16);
      qf[ks] = *(const bf16x8_t*)&v;
    }
  }
  const int n_kt_all = p.causal ? (min(L, q0 + 128) + 63) / 64 : (L + 63) / 64;
  const int n_kt = (OPADPO_ATTN_ABL & 128) ? 0 : (OPADPO_ATTN_ABL & 256) ? min(n_kt_all, 1) : (OPADPO_ATTN_ABL & 512) ? min(n_kt_all, 4) : n_kt_all;
  const SegSkip sk(ge, q0, n_kt);
  const TileSrc<HD> ksrc(p.k, p.ld, ge.row0, L, h, tid), vsrc(p.v, p.ld, ge.row0, L, h, tid);
  // ONE 16-register staging image used twice per iteration (round 4): next K tile fetched at the top and written to the other ring buffer
  // after the S^T phase, next V tile fetched there and written at the end (two images in flight held 32 registers across the iteration)
  TileRegs<HD> kreg, vreg0;
  tile_fetch<HD>(kreg, ksrc, p.ld, sk.first() * 64);
  tile_fetch<HD>(vreg0, vsrc, p.ld, sk.first() * 64);
  // all-padding q tile (OPADPO_ATTN_SKIP_MASKED_Q): zeros, as the 16-row kernel
  if (pad_check) {
    if (__ballot((pm0 | pm1) != 0) == 0) {
      for (int i = tid; i < 128 * 16; i += 256) {
        const int row = q0 + (i >> 4);
        if (row < L) *(uint4*)(p.o + (ge.row0 + row) * p.ldo + h * HD + (i & 15) * 8) = make_uint4(0u, 0u, 0u, 0u);
      }
      if (tid < 128 && q0 + tid < L && p.lse) p.lse[stat_idx(p, ge, s, h, q0 + tid)] = NEG_BIG;
      return;
    }
  }
  bool wave_live = qlo < L, wave_pad = false;
  if (pad_check && wave_live) {                                       // 32 rows that are all masked as keys are padding too: zeros (o = 0, l = 0 below)
    wave_live = __ballot(pmq != 0) != 0;
    wave_pad = !wave_live;
  }
  f32x16_t o[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
  float m_run = NEG_BIG, l_run = 0.f;                     // m in RAW score units; scale2 enters in the exponent's FMA
  const int xlo = seg_xlo(ge), xhi = seg_qstart(ge, qpos);
  const int xhi_first = seg_on(ge) ? seg_qstart(ge, min(qlo, L - 1)) : 0;      // excluded key range [xlo, .) of the wave's first / last row
  const int xhi_last = seg_on(ge) ? seg_qstart(ge, qhi) : 0;
  const float scale2 = p.scale * 1.4426950408889634f;
  tile_commit_k32(smem, kreg, tid);
  tile_commit_v(smem + TILE, vreg0, tid);
  stage_mask(ms_base, p.key_mask, ge.row0, L, sk.first() * 64, tid);
  int cur = 0;
  const int krow_off = ql * 256;                          // K fragment: row kb*32 + ql, 16-byte chunk ks*2 + hi, swizzle kswz32(row) (bit 4 of the row = bit 4 of ql)
  const int kswz = kswz32(ql);
  const int a4 = lane & 15, vgrp = (lane >> 4) & 1;       // V fragment (transpose read): lane a of a 16-lane group addresses row a >> 2, columns (a & 3) * 4
  for (int kt = sk.first(), nxt; kt < n_kt; kt = nxt) {
    nxt = sk.next(kt);
    const int k0 = kt * 64;
    const char* const Ks = smem + cur * 2 * TILE;
    const char* const Vs = Ks + TILE;
    const uint8_t* const Ms = ms_base + cur * 80;
    if (!(OPADPO_ATTN_ABL & 8)) __syncthreads();          // tile kt is in LDS for everyone; everyone has left the other buffer
#if OPADPO_ATTN32_DIAG
    if (dg_tiles++ == 0) dg_t1 = __builtin_readcyclecounter();
#endif
    char* const nb = smem + (cur ^ 1) * 2 * TILE;
    if (nxt < n_kt && !(OPADPO_ATTN_ABL & 16)) tile_fetch<HD>(kreg, ksrc, p.ld, nxt * 64);
    // tiles none of this wave's rows can see: beyond its causal diagonal, or wholly inside the responses its rows exclude
    const bool dead = !wave_live || (p.causal && k0 > qhi) || (k0 >= xlo && k0 + 63 < xhi_first);
    f32x16_t sc[2];
    if (!dead) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
      {
        // S^T = K . Q^T with the K fragments of BOTH 32-key blocks requested FWD32_PF k-steps ahead of the MFMAs that consume them and the
        // two accumulator chains interleaved (round 4).  Left to itself the compiler issued read -> s_waitcnt -> MFMA eight times in a row
        // for the first block (every MFMA behind the full LDS latency of the read in front of it) and then eight dependent MFMAs for the second.
        constexpr int PF = 2;
#if OPADPO_ATTN32_PRIO & 1
        __builtin_amdgcn_s_setprio(1);
#endif
        bf16x8_t kfr[PF + 1][2];
        auto kread = [&](int ks, int kb) {
          if (OPADPO_ATTN_ABL & 32) return qf[ks];
          return *(const bf16x8_t*)(Ks + kb * 32 * 256 + krow_off + (((ks * 2 + hi) ^ kswz) << 4));
        };
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) { kfr[ks][0] = kread(ks, 0); kfr[ks][1] = kread(ks, 1); }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + PF < 8) { kfr[(ks + PF) % (PF + 1)][0] = kread(ks + PF, 0); kfr[(ks + PF) % (PF + 1)][1] = kread(ks + PF, 1); }
          __builtin_amdgcn_sched_barrier(0);
          if (!(OPADPO_ATTN_ABL & 4)) {
          sc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks % (PF + 1)][0], qf[ks], sc[0], 0, 0, 0);
          sc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks % (PF + 1)][1], qf[ks], sc[1], 0, 0, 0);
          } else { asm volatile("" :: "v"(kfr[ks % (PF + 1)][0]), "v"(kfr[ks % (PF + 1)][1])); }
          __builtin_amdgcn_sched_barrier(0);
        }
#if OPADPO_ATTN32_PRIO & 1
        __builtin_amdgcn_s_setprio(0);
#endif
      }
    }
    if (nxt < n_kt && !(OPADPO_ATTN_ABL & 16)) {          // the other buffer is free since this iteration's barrier
      tile_commit_k32(nb, kreg, tid);
      tile_fetch<HD>(kreg, vsrc, p.ld, nxt * 64);
    }
    if (!dead) {
      // sc[kb][r] = S^T[key = k0 + kb*32 + 8*(r >> 2) + 4*hi + (r & 3)][q = qpos], raw (unscaled) scores
      if (!(OPADPO_ATTN_ABL & 1)) {
      const bool clean = !Ms[64] && (!p.causal || k0 + 63 <= qlo) && (k0 + 63 < xlo || k0 >= xhi_last);
      float mx = -INFINITY;
      if (clean) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kb][r]);
      } else {
        // the lane's 64 visibility bits of this tile, built ONCE (key mask of the tile & causal limit of the row & not inside a response
        // the row excludes), shifted so that the lane's key kb*32 + 8b + 4hi + e sits at the compile-time bit kb*32 + 8b + e: three
        // VALU per score (and / compare / select) instead of eleven
        uint64_t vis = *(const uint64_t*)(Ms + 72);
        if (p.causal) {
          const int lim = qpos - k0;                      // keys 0..lim of the tile are at or before the row
          vis &= lim >= 63 ? ~0ull : lim < 0 ? 0ull : ((2ull << lim) - 1ull);
        }
        {
          const int lo = min(max(xlo - k0, 0), 64), hx = min(max(xhi - k0, 0), 64);      // excluded keys [lo, hx) of the tile (empty without segments: xlo = INT_MAX)
          if (hx > lo) {
            const uint64_t below_hx = hx >= 64 ? ~0ull : ((1ull << hx) - 1ull), below_lo = lo >= 64 ? ~0ull : ((1ull << lo) - 1ull);
            vis &= ~(below_hx & ~below_lo);
          }
        }
        vis >>= 4 * hi;
        const uint32_t vw[2] = {(uint32_t)vis, (uint32_t)(vis >> 32)};
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = (vw[kb] & (1u << (8 * (r >> 2) + (r & 3)))) ? sc[kb][r] : -INFINITY;
            sc[kb][r] = v;
            mx = fmaxf(mx, v);
          }
      }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));             // the other 32 keys of the tile for this q row
      const float m_new = fmaxf(m_run, mx);
      // lazy rescale, THRESHOLD form (shipped since round 6): the reference maximum is kept until some row's maximum has grown by more than 2^8 in the
      // exponent's units (exact: O / l and lse do not depend on the reference; P <= 2^8 keeps bf16's relative precision, the row sums are fp32).  With random
      // inputs the plain rule (every change rescales, -DATTN32_RESCALE_LOG2=0.0f: rounds 3-5) fires in most tiles: -2.5 % per launch (profiles/r05h_ab_attn_rescale.txt).
      // Round 5 withheld it because it moved the one-layer parity statistic from 9.87e-4 to 1.011e-3, across north_star's literal 1e-3 - but at that
      // scale the ORACLE's own bf16 emulation reads 9.99e-4 / 1.009e-3 (its two realisations): the statistic is the arithmetic's floor.  Judged against that
      // floor (tests/test_bench_config_parity_gpu.py::test_north_star_1e3_literal..., round 6) this form reads 1.002 x the floor at std 0.02 and 0.96 x at
      // std 0.01 (3.1e-4, where the literal 1e-3 is asserted), the plain form 0.98 x / 0.99 x (profiles/r06a_literal_floor.json).
      // The decision is PER ROW (the two lanes of a row agree: mx was exchanged above): a row whose own maximum did not cross the threshold keeps its
      // reference (alpha = exp2(0) = 1 exactly, its bits untouched) even when the wave enters the branch for another row - so a row's bits depend on
      // its own keys only, never on the later rows that share its wave (tests/test_fullsize_gpu.py P5: an edited token leaves every earlier log-prob BIT-equal).
      const bool moves = (m_new - m_run) * scale2 > ATTN32_RESCALE_LOG2;
      if (__ballot(moves)) {
        const float m_to = moves ? m_new : m_run;
        const float alpha = fast_exp2((m_run - m_to) * scale2);
        l_run *= alpha;
#pragma unroll
        for (int db = 0; db < 4; ++db)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
        m_run = m_to;
      }
      // exponent arguments and row sums two values per instruction (v_pk_fma_f32 / v_pk_add_f32: full rate on CDNA3+); the exp itself
      // is one quarter-rate v_exp_f32 per score.  exp2(-inf) = 0 for masked scores (m stays finite: NEG_BIG floor)
      typedef __attribute__((ext_vector_type(2))) float f32x2v_t;
      const f32x2v_t sc2 = {scale2, scale2}, mb2 = {-m_run * scale2, -m_run * scale2};
      f32x2v_t ps2 = {0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2v_t a = __builtin_elementwise_fma(f32x2v_t{sc[kb][r], sc[kb][r + 1]}, sc2, mb2);
          const f32x2v_t e = {fast_exp2(a[0]), fast_exp2(a[1])};
          sc[kb][r] = e[0]; sc[kb][r + 1] = e[1];
          ps2 += e;
        }
      l_run += ps2[0] + ps2[1];
      }
#if OPADPO_ATTN32_PRIO & 2
      __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          union { bf16x8_t v; uint32_t u[4]; } pf;
          pf.u[0] = pack_bf2(sc[kb][8 * s2 + 0], sc[kb][8 * s2 + 1]); pf.u[1] = pack_bf2(sc[kb][8 * s2 + 2], sc[kb][8 * s2 + 3]);
          pf.u[2] = pack_bf2(sc[kb][8 * s2 + 4], sc[kb][8 * s2 + 5]); pf.u[3] = pack_bf2(sc[kb][8 * s2 + 6], sc[kb][8 * s2 + 7]);
          const int r0 = kb * 32 + s2 * 16 + hi * 4 + (a4 >> 2), r1 = r0 + 8;
#pragma unroll
          for (int db = 0; db < 4; ++db) {
            const int c16 = db * 4 + vgrp * 2 + ((a4 & 3) >> 1), sub = ((a4 & 3) & 1) * 8;      // 16-byte chunk of column db*32 + vgrp*16 + (a & 3)*4, byte inside it
            union { bf16x8_t v; s16x4_t hh[2]; } vf;
            if (!(OPADPO_ATTN_ABL & 64)) {
            vf.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, Vs + off32<false>(r0, c16) + sub));
            vf.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, Vs + off32<false>(r1, c16) + sub));
            } else vf.v = pf.v;
            if (!(OPADPO_ATTN_ABL & 2)) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, o[db], 0, 0, 0);
            else asm volatile("" :: "v"(vf.v), "v"(pf.v));
          }
        }
#if OPADPO_ATTN32_PRIO & 2
      __builtin_amdgcn_s_setprio(0);
#endif
    }
    if (nxt < n_kt && !(OPADPO_ATTN_ABL & 16)) {
      tile_commit_v(nb + TILE, kreg, tid);
      stage_mask(ms_base + (cur ^ 1) * 80, p.key_mask, ge.row0, L, nxt * 64, tid);
    }
    cur ^= 1;
  }
  l_run += __shfl_xor(l_run, 32, 64);
#if OPADPO_ATTN32_DIAG
  const unsigned long long dg_t2 = __builtin_readcyclecounter();
  if (dg_tiles == 0) dg_t1 = dg_t2;
#endif
  __syncthreads();                                        // every wave has left the ring: it becomes the O staging area
  // O^T[d][q]: lane (q, hi) holds d = db*32 + 8b + 4hi + (0..3).  Through LDS as bf16 rows ([32][128] per wave, 16-byte chunks XOR
  // (row & 15)) and out as whole 256-byte rows.
  char* const stg = smem + w * 8192;
  const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      uint2 v;
      v.x = pack_bf2(o[db][b * 4 + 0] * inv, o[db][b * 4 + 1] * inv);
      v.y = pack_bf2(o[db][b * 4 + 2] * inv, o[db][b * 4 + 3] * inv);
      const int d = db * 32 + b * 8 + hi * 4;             // first of the lane's 4 consecutive columns
      *(uint2*)(stg + ql * 256 + ((((d >> 3) ^ (ql & 15))) << 4) + (d & 7) * 2) = v;
    }
  // rows of a skipped 32-row group sit inside q tiles the backward still walks (its skip granularity is 64 rows): their log-sum-exp is
  // +BIG so that the recomputed P = exp(s - lse) is exactly 0 there (NEG_BIG would give inf * 0 on the clean-tile path)
  if (hi == 0 && qpos < L && p.lse)
    p.lse[stat_idx(p, ge, s, h, qpos)] = l_run > 0.f ? (m_run * scale2 + log2f(l_run)) * 0.6931471805599453f : (wave_pad ? -NEG_BIG : NEG_BIG);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // own rows only: no barrier needed
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 4 + (lane >> 4), c16 = lane & 15;
    const uint4 v = *(const uint4*)(stg + row * 256 + ((c16 ^ (row & 15)) << 4));
    if (qlo + row < L) *(uint4*)(p.o + (ge.row0 + qlo + row) * p.ldo + h * HD + c16 * 8) = v;
  }
#if OPADPO_ATTN32_DIAG
  if (tid == 0) {
    const unsigned long long dg_t3 = __builtin_readcyclecounter();
    atomicAdd(&g_attn32_diag[0], dg_t1 - dg_t0); atomicAdd(&g_attn32_diag[1], dg_t2 - dg_t1); atomicAdd(&g_attn32_diag[2], dg_t3 - dg_t2);
    atomicAdd(&g_attn32_diag[3], 1ull); atomicAdd(&g_attn32_diag[4], dg_tiles);
  }
#endif
}

// (Round 5 built a 64-rows-per-wave forward here - one wave per SIMD, 512-register budget, S in arch VGPRs, Q / O in the accumulator file through asm MFMAs,
// K / V by asm LDS-DMA into a three-stage ring - to parity; compiler-scheduled it ran 39 % slower than the 32-row kernel above (0.581 vs 0.417 ms; per-phase
// cycle anatomy: profiles/r05g_attn_fwd64.txt) and was removed in round 6: the structure pays only as generated text with a register plan.)

// delta[s,h,pos] = sum_d dO * O.  HD/8 lanes per (row, head), 16 bytes of dO and of O per lane (a wave covers 64 / (HD/8) heads of
// one row: 512 contiguous bytes per operand at HD = 128), shuffle reduction inside the lane group.
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(AttnArgs p) {
  constexpr int LPH = HD / 8;                               // lanes per head: 16 (hd 128) or 8 (hd 64)
  constexpr int HPW = 64 / LPH;                             // heads per wave
  const int lane = threadIdx.x & 63;
  const size_t unit = ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * HPW + lane / LPH;      // (row, head) index
  const size_t total = (p.seq_meta ? (size_t)p.rows_total : (size_t)p.S * p.L) * p.nh;
  const bool live = unit < total;
  const size_t row = live ? unit / p.nh : 0;
  const int h = live ? (int)(unit % p.nh) : 0, i0 = (lane % LPH) * 8;
  float acc = 0.f;
  if (live) {
    float a[8], b[8];
    unpack8(*(const uint4*)(p.dout + row * p.ldo + h * HD + i0), a);
    unpack8(*(const uint4*)(p.o + row * p.ldo + h * HD + i0), b);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += a[j] * b[j];
  }
#pragma unroll
  for (int off = LPH / 2; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
  if (live && lane % LPH == 0) {
    if (p.seq_meta) {
      p.delta[(size_t)h * p.rows_total + row] = acc;        // ragged: [nh, rows_total]
    } else {
      const size_t sq = row / p.L, pos = row % p.L;
      p.delta[(sq * p.nh + h) * p.L + pos] = acc;
    }
  }
}

// ---- backward part 1: dK, dV (KV-outer) ---------------------------------------------------------
template <int HD, bool TR>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkdv_kernel(AttnArgs p) {
  constexpr int KK = HD / 32, DF = HD / 16;
  constexpr int TILE = 64 * HD * 2;
  // Q / dO tiles: direct-to-LDS double-buffered ring (the kernel sits at 2 blocks per CU for its registers anyway, so the
  // 64 KiB of LDS are free): the next q tile is in flight while this one is consumed, one barrier per tile.
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const ls_base = (float*)(smem + 4 * TILE);        // [2][lse 64 | delta 64]
  uint8_t* Ms = (uint8_t*)(ls_base + 256);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  int s, h, kt;
  attn_block_map(p, (p.L + 63) / 64, kt, h, s);
  const Geo ge = load_geo(p, s);
  const int L = ge.L;
  const int k0 = kt * 64;
  if (k0 >= L) return;
  const int kpos = k0 + w * 16 + c;          // this lane's key (as fragment row / output row)
  const int krow = min(kpos, L - 1);

  stage_mask(Ms, p.key_mask, ge.row0, L, k0, tid);
  bf16x8_t kf[KK], vf[KK];
  {
    const bf16_t* kp_ = p.k + (ge.row0 + krow) * p.ld + h * HD;
    const bf16_t* vp_ = p.v + (ge.row0 + krow) * p.ld + h * HD;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      uint4 a = *(const uint4*)(kp_ + kk * 32 + g * 8);
      uint4 b = *(const uint4*)(vp_ + kk * 32 + g * 8);
      kf[kk] = *(bf16x8_t*)&a;
      vf[kk] = *(bf16x8_t*)&b;
    }
  }
  f32x4_t dk[DF], dv[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) { dk[d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[d] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
  // causal bit 1 (OPADPO_ATTN_SKIP_MASKED_Q): q tiles whose 64 positions are all masked as keys are padding, their dO is exactly
  // zero -> they contribute nothing to dK / dV and are left out of the loop.  Bit t of `live` = q tile t has a valid position
  // (wave w scans tiles w, w+4, ...; one 64-byte read per tile).
  unsigned long long* const live_s = (unsigned long long*)(smem + 4 * TILE + 1024 + 80);
  const bool skip_q = (p.causal & 2) && p.key_mask && (L + 63) / 64 <= 64;
  if (skip_q) {
    unsigned long long mine = 0;
    for (int t = w; t < (L + 63) / 64; t += 4) {
      const int qp_ = t * 64 + lane;
      const uint8_t mq = qp_ < L ? p.key_mask[ge.row0 + qp_] : (uint8_t)0;
      if (__ballot(mq != 0) != 0) mine |= 1ull << t;
    }
    if (lane == 0) live_s[w] = mine;
  }
  __syncthreads();
  const unsigned long long live = skip_q ? (live_s[0] | live_s[1] | live_s[2] | live_s[3]) : ~0ull;
  const bool key_ok = Ms[w * 16 + c] != 0;
  const bool keys_clean = !Ms[64];
  const float scale2 = p.scale * 1.4426950408889634f;

  int n_qt = (L + 63) / 64;
  int kend = 0x7fffffff;                     // packed responses: a key in segment a is visible to queries < end of segment a
  int kend_min = 0x7fffffff;                 // smallest kend over the keys of the tile (block-uniform)
  if (seg_on(ge) && k0 + 63 >= ge.pfx) kend_min = seg_end(ge, max(k0, ge.pfx));
  if (seg_on(ge)) {
    if (kpos >= ge.pfx) kend = seg_end(ge, min(kpos, L - 1));
    if (k0 >= ge.pfx) {                      // whole tile inside the response area: later segments never see it
      const int q_cut = seg_end(ge, min(k0 + 63, L - 1));
      n_qt = min(n_qt, (q_cut + 63) / 64);
    }
  }
  const TileDma<HD> qdma(p.q, p.ld, ge.row0, L, h, tid), dodma(p.dout, p.ldo, ge.row0, L, h, tid);
  float lse_r = 0.f, dlt_r = 0.f;      // wave 0: lse / delta of the tile in flight (written to LDS one iteration later, so
                                       // that nobody waits on these loads right behind the DMA issue)
  auto stage = [&](int buf, int q0) {
    qdma.issue(smem + buf * 2 * TILE, p.ld, q0);
    dodma.issue(smem + buf * 2 * TILE + TILE, p.ldo, q0);
    if (tid < 64) {
      const size_t li = stat_idx(p, ge, s, h, min(q0 + tid, L - 1));
      lse_r = p.lse[li] * 1.4426950408889634f;      // log2 units
      dlt_r = p.delta[li];
    }
  };
  auto next_live = [&](int t) {              // first live q tile >= t (n_qt when there is none)
    if (t >= n_qt) return n_qt;
    if (!skip_q) return t;
    const unsigned long long m = live >> t;
    return m ? min(n_qt, t + (int)__builtin_ctzll(m)) : n_qt;
  };
  const int qt0 = next_live(p.causal ? kt : 0);
  if (qt0 < n_qt) stage(0, qt0 * 64);
  int cur = 0;
  for (int qt = qt0, qn; qt < n_qt; qt = qn) {
    qn = next_live(qt + 1);
    const int q0 = qt * 64;
    const char* Qs = smem + cur * 2 * TILE;
    const char* dOs = Qs + TILE;
    const float* lse_s = ls_base + cur * 128;
    const float* dlt_s = lse_s + 64;
    if (tid < 64) {
      ls_base[cur * 128 + tid] = lse_r;
      ls_base[cur * 128 + 64 + tid] = dlt_r;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();          // tile qt landed for everyone; everyone is done with the other buffer
    if (qn < n_qt) stage(cur ^ 1, qn * 64);
    cur ^= 1;

    f32x4_t sc[4], dp[4];
#pragma unroll
    for (int qf = 0; qf < 4; ++qf) {
      sc[qf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      dp[qf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        // D[i = q][j = key]: lane holds key = c, q = qf*16 + g*4 + r
        sc[qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<HD>(Qs, qf * 16 + c, kk, g), kf[kk], sc[qf], 0, 0, 0);
        dp[qf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<HD>(dOs, qf * 16 + c, kk, g), vf[kk], dp[qf], 0, 0, 0);
      }
    }
    // block-uniform: every (q, key) of this tile pair is visible -> no mask arithmetic
    const bool clean = keys_clean && q0 + 63 < L && (!p.causal || k0 + 63 <= q0) && q0 + 63 < kend_min;
    if (clean) {          // straight-line: 16 independent exp chains the scheduler can interleave with the MFMAs
#pragma unroll
      for (int qf = 0; qf < 4; ++qf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = qf * 16 + g * 4 + r;
          const float pv = fast_exp2(sc[qf][r] * scale2 - lse_s[ql]);
          sc[qf][r] = pv;
          dp[qf][r] = pv * (dp[qf][r] - dlt_s[ql]) * p.scale;
        }
    } else {              // masked pairs: exp2(-inf) = 0 through a select, not a branch per element
#pragma unroll
      for (int qf = 0; qf < 4; ++qf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ql = qf * 16 + g * 4 + r;
          const int qp = q0 + ql;
          const bool ok = key_ok & (qp < L) & (!p.causal | (kpos <= qp)) & (qp < kend);      // bitwise: no branch per term
          const float arg = sc[qf][r] * scale2 - lse_s[ql];                                   // LDS read outside the select: no branch
          const float pv = fast_exp2(ok ? arg : -INFINITY);
          sc[qf][r] = pv;
          dp[qf][r] = pv * (dp[qf][r] - dlt_s[ql]) * p.scale;
        }
    }
    // dV^T[d][key] += dO^T . P ; dK^T[d][key] += Q^T . dS   (contraction over the 64 q rows)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t pf = pack_frag(sc[2 * ks], sc[2 * ks + 1]);
      const bf16x8_t dsf = pack_frag(dp[2 * ks], dp[2 * ks + 1]);
      const int r0 = (2 * ks) * 16 + g * 4, r1 = (2 * ks + 1) * 16 + g * 4;
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        dv[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_col<HD, TR>(dOs, r0, r1, d * 16, lane), pf, dv[d], 0, 0, 0);
        dk[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_col<HD, TR>(Qs, r0, r1, d * 16, lane), dsf, dk[d], 0, 0, 0);
      }
    }
  }
  if (p.rope_pos) rope_inverse_rows<HD>(dk, g, p.rope_pos[ge.row0 + krow], p.rope_l2theta);      // dK back through the rotary embedding of k
  if (kpos < L) {
    bf16_t* dkp = p.dk + (ge.row0 + kpos) * p.ld + h * HD;
    bf16_t* dvp = p.dv + (ge.row0 + kpos) * p.ld + h * HD;
#pragma unroll
    for (int d = 0; d < DF; ++d) {
      uint2 a, b;
      a.x = pack_bf2(dk[d][0], dk[d][1]); a.y = pack_bf2(dk[d][2], dk[d][3]);
      b.x = pack_bf2(dv[d][0], dv[d][1]); b.y = pack_bf2(dv[d][2], dv[d][3]);
      *(uint2*)(dkp + d * 16 + g * 4) = a;
      *(uint2*)(dvp + d * 16 + g * 4) = b;
    }
  }
}

// ---- backward part 2: dQ (Q-outer, forward geometry, no atomics) ---------------------------------
template <int HD, bool TR>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(AttnArgs p) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 64 * HD * 2 + 80];
  char* Ks = smem;
  char* Vs = smem + 64 * HD * 2;
  uint8_t* Ms = (uint8_t*)(smem + 2 * 64 * HD * 2);
  constexpr int KK = HD / 32, DF = HD / 16;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int n_qt = (p.L + 63) / 64;
  int s, h, qi;
  attn_block_map(p, n_qt, qi, h, s);
  const int qt = n_qt - 1 - qi;
  const int q0 = qt * 64;
  const Geo ge = load_geo(p, s);
  const int L = ge.L;
  if (q0 >= L) return;
  const int qpos = q0 + w * 16 + c;
  const int qrow = min(qpos, L - 1);
  if ((p.causal & 2) && p.key_mask) {          // all-padding q tile (see attn_fwd_kernel): its dQ is exactly zero
    const int qp_ = q0 + lane;
    const uint8_t mq = qp_ < L ? p.key_mask[ge.row0 + qp_] : (uint8_t)0;
    if (__ballot(mq != 0) == 0) {
      if (qpos < L) {
        if (p.dq) {
          bf16_t* dqp = p.dq + (ge.row0 + qpos) * p.ld + h * HD;
#pragma unroll
          for (int d = 0; d < DF; ++d) *(uint2*)(dqp + d * 16 + g * 4) = make_uint2(0u, 0u);
        }
        if (p.dq_acc) {
          float* dqa = p.dq_acc + (ge.row0 + qpos) * (size_t)(p.nh * HD) + h * HD;
#pragma unroll
          for (int d = 0; d < DF; ++d) *(float4*)(dqa + d * 16 + g * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      return;
    }
  }

  bf16x8_t qf[KK], dof[KK];
  {
    const bf16_t* qp = p.q + (ge.row0 + qrow) * p.ld + h * HD;
    const bf16_t* dp_ = p.dout + (ge.row0 + qrow) * p.ldo + h * HD;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
      uint4 a = *(const uint4*)(qp + kk * 32 + g * 8);
      uint4 b = *(const uint4*)(dp_ + kk * 32 + g * 8);
      qf[kk] = *(bf16x8_t*)&a;
      dof[kk] = *(bf16x8_t*)&b;
    }
  }
  const size_t li = stat_idx(p, ge, s, h, qrow);
  const float lse2 = p.lse[li] * 1.4426950408889634f, dlt = p.delta[li];     // log2 units -> bare v_exp_f32
  const float scale2 = p.scale * 1.4426950408889634f;
  f32x4_t dq[DF];
#pragma unroll
  for (int d = 0; d < DF; ++d) dq[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int n_kt = p.causal ? (min(L, q0 + 64) + 63) / 64 : (L + 63) / 64;
  const SegSkip sk(ge, q0, n_kt);
  const int xlo = seg_xlo(ge), xhi = seg_qstart(ge, qpos);
  const int xhi_blk = seg_on(ge) ? seg_qstart(ge, q0 + 63) : 0;
  const TileSrc<HD> ksrc(p.k, p.ld, ge.row0, L, h, tid), vsrc(p.v, p.ld, ge.row0, L, h, tid);
  TileRegs<HD> kreg, vreg;
  tile_fetch<HD>(kreg, ksrc, p.ld, sk.first() * 64);
  tile_fetch<HD>(vreg, vsrc, p.ld, sk.first() * 64);
  for (int kt = sk.first(), nxt; kt < n_kt; kt = nxt) {
    nxt = sk.next(kt);
    const int k0 = kt * 64;
    tile_commit<HD>(Ks, kreg, tid);
    tile_commit<HD>(Vs, vreg, tid);
    stage_mask(Ms, p.key_mask, ge.row0, L, k0, tid);
    __syncthreads();
    if (nxt < n_kt) {
      tile_fetch<HD>(kreg, ksrc, p.ld, nxt * 64);
      tile_fetch<HD>(vreg, vsrc, p.ld, nxt * 64);
    }
    f32x4_t sc[4], dp[4];
#pragma unroll
    for (int kf = 0; kf < 4; ++kf) {
      sc[kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      dp[kf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < KK; ++kk) {
        // D[i = key][j = q]: lane holds q = c, key = kf*16 + g*4 + r
        sc[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<HD>(Ks, kf * 16 + c, kk, g), qf[kk], sc[kf], 0, 0, 0);
        dp[kf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_row<HD>(Vs, kf * 16 + c, kk, g), dof[kk], dp[kf], 0, 0, 0);
      }
    }
    const bool clean = !Ms[64] && q0 + 63 < L && (!p.causal || k0 + 63 <= q0) && (k0 + 63 < xlo || k0 >= xhi_blk);
    if (clean) {
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = fast_exp2(sc[kf][r] * scale2 - lse2);
          dp[kf][r] = pv * (dp[kf][r] - dlt) * p.scale;
        }
    } else {
      uint32_t mw[4];
#pragma unroll
      for (int kf = 0; kf < 4; ++kf) mw[kf] = *(const uint32_t*)(Ms + kf * 16 + g * 4);
#pragma unroll
      for (int kf = 0; kf < 4; ++kf)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int kl = kf * 16 + g * 4 + r;
          const bool ok = (((mw[kf] >> (8 * r)) & 0xffu) != 0) & (qpos < L) & (!p.causal | ((k0 + kl) <= qpos)) & (((k0 + kl) < xlo) | ((k0 + kl) >= xhi));
          const float pv = fast_exp2(ok ? sc[kf][r] * scale2 - lse2 : -INFINITY);
          dp[kf][r] = pv * (dp[kf][r] - dlt) * p.scale;
        }
    }
    // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const bf16x8_t dsf = pack_frag(dp[2 * ks], dp[2 * ks + 1]);
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        const bf16x8_t kfr = frag_col<HD, TR>(Ks, (2 * ks) * 16 + g * 4, (2 * ks + 1) * 16 + g * 4, d * 16, lane);
        dq[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr, dsf, dq[d], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (p.rope_pos) rope_inverse_rows<HD>(dq, g, p.rope_pos[ge.row0 + qrow], p.rope_l2theta);       // dQ back through the rotary embedding of q
  if (qpos < L) {
    if (p.dq) {
      bf16_t* dst = p.dq + (ge.row0 + qpos) * p.ld + h * HD;
#pragma unroll
      for (int d = 0; d < DF; ++d) {
        uint2 v;
        v.x = pack_bf2(dq[d][0], dq[d][1]);
        v.y = pack_bf2(dq[d][2], dq[d][3]);
        *(uint2*)(dst + d * 16 + g * 4) = v;
      }
    }
    if (p.dq_acc) {   // optional fp32 copy (diagnostics / tests)
      float* dst = p.dq_acc + (ge.row0 + qpos) * (p.nh * HD) + h * HD;
#pragma unroll
      for (int d = 0; d < DF; ++d) *(float4*)(dst + d * 16 + g * 4) = make_float4(dq[d][0], dq[d][1], dq[d][2], dq[d][3]);
    }
  }
}

// ---- backward part 2, round 3 (head_dim 128): dQ on 32 q rows per wave, the geometry of attn_fwd32_kernel ----------------------------
//   S^T[key][q]  = K . Q^T      A = K row fragment (ds_read_b128),  B = Q in registers
//   dP^T[key][q] = V . dO^T     A = V row fragment (ds_read_b128),  B = dO in registers
//   dS^T = P^T * (dP^T - delta) * scale, P^T = exp2(S^T * scale2 - lse2)      (lane (q, hi) holds keys kb*32 + 8b + 4hi + (0..3))
//   dQ^T[d][q] += K^T[d][key] . dS^T[key][q]     A = K fragment through two ds_read_b64_tr_b16 at exactly those key rows, B = dS^T from
//                                                 the accumulators (v_cvt_pk_bf16_f32) - the forward's P . V trick
// K tile in the off32<false> layout (transposing reads conflict-free, row reads 2-way), V tile in off32<true> (row reads only: conflict-free), double-buffered ring,
// register-staged, one barrier per tile; one 32-key block at a time so that S and dP cost 32 registers, not 64.
__global__ __launch_bounds__(256, 2) void attn_bwd_dq32_kernel(AttnArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int HD = 128, TILE = 64 * HD * 2;
  uint8_t* const ms_base = (uint8_t*)(smem + 4 * TILE);
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ql = lane & 31, hi = lane >> 5;
  const int n_qt = (p.L + 127) / 128;
  int s, h, qi;
  attn_block_map(p, n_qt, qi, h, s);
  const int qt = n_qt - 1 - qi;
  const int q0 = qt * 128;
  const Geo ge = load_geo(p, s);
  const int L = ge.L;
  if (q0 >= L) return;
  const int qlo = q0 + w * 32, qhi = min(qlo + 31, L - 1);
  bool wave_live = qlo < L;
  if ((p.causal & 2) && p.key_mask && wave_live) {      // 32 rows that are all padding: dQ = 0 (their dO is zero)
    const int a_ = qlo + ql;
    const uint8_t mq = a_ < L ? p.key_mask[ge.row0 + a_] : (uint8_t)0;
    wave_live = __ballot(mq != 0) != 0;
  }
  const int qpos = qlo + ql;
  const int qrow = min(qpos, L - 1);
  bf16x8_t qf[8], dof[8];
  {
    const bf16_t* qp = p.q + (ge.row0 + qrow) * p.ld + h * HD + hi * 8;
    const bf16_t* dp_ = p.dout + (ge.row0 + qrow) * p.ldo + h * HD + hi * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const uint4 a = *(const uint4*)(qp + ks * 16), b = *(const uint4*)(dp_ + ks * 16);
      qf[ks] = *(const bf16x8_t*)&a;
      dof[ks] = *(const bf16x8_t*)&b;
    }
  }
  const size_t li = stat_idx(p, ge, s, h, qrow);
  const float lse2 = p.lse[li] * 1.4426950408889634f, dlt = p.delta[li];
  const float scale2 = p.scale * 1.4426950408889634f;
  f32x16_t dq[4];
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[db][r] = 0.f;
  const int n_kt = p.causal ? (min(L, q0 + 128) + 63) / 64 : (L + 63) / 64;
  const SegSkip sk(ge, q0, n_kt);
  const int xlo = seg_xlo(ge), xhi = seg_qstart(ge, qpos);
  const int xhi_first = seg_on(ge) ? seg_qstart(ge, min(qlo, L - 1)) : 0;
  const int xhi_last = seg_on(ge) ? seg_qstart(ge, qhi) : 0;
  const TileSrc<HD> ksrc(p.k, p.ld, ge.row0, L, h, tid), vsrc(p.v, p.ld, ge.row0, L, h, tid);
  // staging registers: ONE 16-register tile image, used twice per iteration (round 4) - the next K tile is fetched at the top of the
  // iteration and written to the other ring buffer after the first 32-key block, the next V tile is fetched there and written at the
  // end.  Two images in flight (round 3) held 32 registers across the whole iteration; the 16 freed ones let the K / V row fragments of
  // the S / dP products be requested DQ32_PF k-steps ahead of their MFMAs without spilling.
  TileRegs<HD> kreg;
  tile_fetch<HD>(kreg, ksrc, p.ld, sk.first() * 64);
  tile_commit_v<false>(smem, kreg, tid);
  tile_fetch<HD>(kreg, vsrc, p.ld, sk.first() * 64);
  tile_commit_v<DQ32_VFLIP>(smem + TILE, kreg, tid);
  stage_mask(ms_base, p.key_mask, ge.row0, L, sk.first() * 64, tid);
  int cur = 0;
  const int a4 = lane & 15, vgrp = (lane >> 4) & 1;
  auto row_off = [&](int row, int c16) { return off32<false>(row, c16); };      // K tile: row fragments AND transposing reads
  auto row_off_v = [&](int row, int c16) { return off32<DQ32_VFLIP>(row, c16); };     // V tile: row fragments only
  for (int kt = sk.first(), nxt; kt < n_kt; kt = nxt) {
    nxt = sk.next(kt);
    const int k0 = kt * 64;
    const char* const Ks = smem + cur * 2 * TILE;
    const char* const Vs = Ks + TILE;
    const uint8_t* const Ms = ms_base + cur * 80;
    char* const nb = smem + (cur ^ 1) * 2 * TILE;
    __syncthreads();
    if (nxt < n_kt) tile_fetch<HD>(kreg, ksrc, p.ld, nxt * 64);
    const bool dead = !wave_live || (p.causal && k0 > qhi) || (k0 >= xlo && k0 + 63 < xhi_first);
    const bool clean = !Ms[64] && qhi == qlo + 31 && (!p.causal || k0 + 63 <= qlo) && (k0 + 63 < xlo || k0 >= xhi_last);
    uint32_t vw[2] = {~0u, ~0u};
    if (!dead && !clean) {
      uint64_t vis = *(const uint64_t*)(Ms + 72);
      if (qpos >= L) vis = 0;
      if (p.causal) {
        const int lim = qpos - k0;
        vis &= lim >= 63 ? ~0ull : lim < 0 ? 0ull : ((2ull << lim) - 1ull);
      }
      {
        const int lo = min(max(xlo - k0, 0), 64), hx = min(max(xhi - k0, 0), 64);
        if (hx > lo) {
          const uint64_t below_hx = hx >= 64 ? ~0ull : ((1ull << hx) - 1ull), below_lo = lo >= 64 ? ~0ull : ((1ull << lo) - 1ull);
          vis &= ~(below_hx & ~below_lo);
        }
      }
      vis >>= 4 * hi;
      vw[0] = (uint32_t)vis; vw[1] = (uint32_t)(vis >> 32);
    }
    auto kb_block = [&](auto KB_) {
      constexpr int kb = decltype(KB_)::value;
      f32x16_t sc, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sc[r] = 0.f; dp[r] = 0.f; }
      const int krow = kb * 32 + ql;
      {
        // K / V row fragments requested DQ32_PF k-steps ahead of their MFMAs (the compiler's own order was read, read, wait, MFMA, wait,
        // MFMA per k-step: every pair of MFMAs behind the full LDS latency)
        constexpr int PF = DQ32_PF;
        bf16x8_t kfr[PF + 1], vfr[PF + 1];
#pragma unroll
        for (int ks = 0; ks < PF; ++ks) {
          kfr[ks] = *(const bf16x8_t*)(Ks + row_off(krow, ks * 2 + hi));
          vfr[ks] = *(const bf16x8_t*)(Vs + row_off_v(krow, ks * 2 + hi));
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + PF < 8) {
            kfr[(ks + PF) % (PF + 1)] = *(const bf16x8_t*)(Ks + row_off(krow, (ks + PF) * 2 + hi));
            vfr[(ks + PF) % (PF + 1)] = *(const bf16x8_t*)(Vs + row_off_v(krow, (ks + PF) * 2 + hi));
          }
          if (PF > 0) __builtin_amdgcn_sched_barrier(0);
          sc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks % (PF + 1)], qf[ks], sc, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[ks % (PF + 1)], dof[ks], dp, 0, 0, 0);
          if (PF > 0) __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float arg = __builtin_fmaf(sc[r], scale2, -lse2);
        const bool ok = clean || (vw[kb] & (1u << (8 * (r >> 2) + (r & 3))));
        const float pv = fast_exp2(ok ? arg : -INFINITY);
        dp[r] = pv * (dp[r] - dlt) * p.scale;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        union { bf16x8_t v; uint32_t u[4]; } df;
        df.u[0] = pack_bf2(dp[8 * s2 + 0], dp[8 * s2 + 1]); df.u[1] = pack_bf2(dp[8 * s2 + 2], dp[8 * s2 + 3]);
        df.u[2] = pack_bf2(dp[8 * s2 + 4], dp[8 * s2 + 5]); df.u[3] = pack_bf2(dp[8 * s2 + 6], dp[8 * s2 + 7]);
        const int r0 = kb * 32 + s2 * 16 + hi * 4 + (a4 >> 2), r1 = r0 + 8;
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          const int c16 = db * 4 + vgrp * 2 + ((a4 & 3) >> 1), sub = ((a4 & 3) & 1) * 8;
          union { bf16x8_t v; s16x4_t hh[2]; } kf2;
          kf2.hh[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, Ks + row_off(r0, c16) + sub));
          kf2.hh[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(LDS_PTR(s16x4_t, Ks + row_off(r1, c16) + sub));
          dq[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf2.v, df.v, dq[db], 0, 0, 0);
        }
      }
    };
    if (!dead) kb_block(std::integral_constant<int, 0>{});
    if (nxt < n_kt) {                                     // the other buffer is free since this iteration's barrier
      tile_commit_v<false>(nb, kreg, tid);
      tile_fetch<HD>(kreg, vsrc, p.ld, nxt * 64);
    }
    if (!dead) kb_block(std::integral_constant<int, 1>{});
    if (nxt < n_kt) {
      tile_commit_v<DQ32_VFLIP>(nb + TILE, kreg, tid);
      stage_mask(ms_base + (cur ^ 1) * 80, p.key_mask, ge.row0, L, nxt * 64, tid);
    }
    cur ^= 1;
  }
  // dQ^T[d][q]: lane (q, hi) holds d = db*32 + 8b + 4hi + (0..3); the rotary pair (d, d + 64) = (db, db + 2) sits in the same lane
  if (p.rope_pos) {
    const float posf = (float)p.rope_pos[ge.row0 + qrow];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = db * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
        const float frev = __builtin_amdgcn_exp2f(-(float)(2 * i) * (1.0f / HD) * p.rope_l2theta) * 0.15915494309189535f;
        const float x = __builtin_amdgcn_fractf(posf * frev);
        const float c = __builtin_amdgcn_cosf(x), sn = __builtin_amdgcn_sinf(x);
        const float x1 = dq[db][r], x2 = dq[db + 2][r];
        dq[db][r] = x1 * c + x2 * sn;
        dq[db + 2][r] = x2 * c - x1 * sn;
      }
  }
  __syncthreads();
  char* const stg = smem + w * 8192;
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      uint2 v;
      v.x = pack_bf2(dq[db][b * 4 + 0], dq[db][b * 4 + 1]);
      v.y = pack_bf2(dq[db][b * 4 + 2], dq[db][b * 4 + 3]);
      const int d = db * 32 + b * 8 + hi * 4;
      *(uint2*)(stg + ql * 256 + ((((d >> 3) ^ (ql & 15))) << 4) + (d & 7) * 2) = v;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int row = it * 4 + (lane >> 4), c16 = lane & 15;
    const uint4 v = *(const uint4*)(stg + row * 256 + ((c16 ^ (row & 15)) << 4));
    if (qlo + row < L) *(uint4*)(p.dq + (ge.row0 + qlo + row) * p.ld + h * HD + c16 * 8) = v;
  }
}

}  // namespace

static bool g_attn_dma = false;   // measured: the 64-KiB ring allows 2 blocks/CU, the 32-KiB register-staged kernel 3 -> 0.92 vs 1.01 ms
void opadpo_set_attn_dma(bool on) { g_attn_dma = on; }

static int g_attn32 = -1;         // 1 (default): head_dim-128 forward on the 32-rows-per-wave kernel; OPADPO_ATTN32=0 keeps the 16-row kernel (A/B)
hipError_t launch_attn_fwd(const AttnArgs& a, hipStream_t st) {
  if (a.S <= 0 || a.L <= 0) return hipSuccess;
  if (a.hd != 128 && a.hd != 64) return hipErrorInvalidValue;
  if (g_attn32 < 0) { const char* v = getenv("OPADPO_ATTN32"); g_attn32 = (v && v[0] == '0') ? 0 : 1; }
  const bool legacy = a.use_tr >= 0 && (a.use_tr & 256);      // per-call: context flag bit 8 = the 16-row forward kernel
  if (a.hd == 128 && g_attn32 && !legacy && (double)a.L * a.ld * 2 < 2.0e9) {
    static bool attr32 = false;
    if (!attr32) { (void)hipFuncSetAttribute((const void*)attn_fwd32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128 * 2 + 160); attr32 = true; }
    const dim3 grid32((unsigned)(((a.L + 127) / 128) * a.nh * a.S));
    hipLaunchKernelGGL(attn_fwd32_kernel, grid32, dim3(256), 4 * 64 * 128 * 2 + 160, st, a);
    return hipGetLastError();
  }
  const dim3 grid((unsigned)(((a.L + 63) / 64) * a.nh * a.S));
  const bool tr = a.use_tr >= 0 ? (a.use_tr & 1) != 0 : opadpo_flag_tr();
  const bool dma = g_attn_dma && (double)a.L * a.ld * 2 < 2.0e9;     // per-sequence extent must fit the 32-bit descriptor
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128 * 2 + 160);
    (void)hipFuncSetAttribute((const void*)attn_fwd_kernel<128, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128 * 2 + 160);
    attr_set = true;
  }
#define FWD(HD_, TR_, DMA_) hipLaunchKernelGGL((attn_fwd_kernel<HD_, TR_, DMA_>), grid, dim3(256), (DMA_ ? 4 : 2) * 64 * HD_ * 2 + (DMA_ ? 160 : 80), st, a)
  if (a.hd == 128) {
    if (tr) { if (dma) FWD(128, true, true); else FWD(128, true, false); }
    else    { if (dma) FWD(128, false, true); else FWD(128, false, false); }
  } else {
    if (tr) { if (dma) FWD(64, true, true); else FWD(64, true, false); }
    else    { if (dma) FWD(64, false, true); else FWD(64, false, false); }
  }
#undef FWD
  return hipGetLastError();
}

hipError_t launch_attn_bwd(const AttnArgs& a, hipStream_t st) {
  if (a.S <= 0 || a.L <= 0) return hipSuccess;
  if (a.hd != 128 && a.hd != 64) return hipErrorInvalidValue;
  const int total = (a.seq_meta ? a.rows_total : a.S * a.L) * a.nh;
  const dim3 grid((unsigned)(((a.L + 63) / 64) * a.nh * a.S));
  const bool tr = a.use_tr >= 0 ? (a.use_tr & 1) != 0 : opadpo_flag_tr();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128 * 2 + 1024 + 80 + 64);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkdv_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128 * 2 + 1024 + 80 + 64);
    attr_set = true;
  }
  if ((double)a.L * a.ld * 2 >= 2.0e9 || (double)a.L * a.ldo * 2 >= 2.0e9) return hipErrorInvalidValue;   // 32-bit buffer extents
  static int g_dq32 = -1;
  if (g_dq32 < 0) {
    const char* v = getenv("OPADPO_DQ32");
    g_dq32 = (v && v[0] == '0') ? 0 : 1;      // default: the 32-rows-per-wave dQ kernel at head_dim 128 (bf16 dQ, no fp32 copy); OPADPO_DQ32=0 / context flag bit 8 keep the 16-row kernel
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 64 * 128 * 2 + 160);
  }
  const bool dq32 = g_dq32 && a.dq && !a.dq_acc && !(a.use_tr >= 0 && (a.use_tr & 256));
#define LAUNCH_BWD(HD_, TR_)                                                                              \
  hipLaunchKernelGGL((attn_delta_kernel<HD_>), dim3((unsigned)((total + 4 * (512 / HD_) - 1) / (4 * (512 / HD_)))), dim3(256), 0, st, a); \
  hipLaunchKernelGGL((attn_bwd_dkdv_kernel<HD_, TR_>), grid, dim3(256), 4 * 64 * HD_ * 2 + 1024 + 80 + 64, st, a); \
  if (HD_ == 128 && dq32) hipLaunchKernelGGL(attn_bwd_dq32_kernel, dim3((unsigned)(((a.L + 127) / 128) * a.nh * a.S)), dim3(256), 4 * 64 * 128 * 2 + 160, st, a); \
  else hipLaunchKernelGGL((attn_bwd_dq_kernel<HD_, TR_>), grid, dim3(256), 0, st, a)
  if (a.hd == 128) {
    if (tr) { LAUNCH_BWD(128, true); } else { LAUNCH_BWD(128, false); }
  } else {
    if (tr) { LAUNCH_BWD(64, true); } else { LAUNCH_BWD(64, false); }
  }
#undef LAUNCH_BWD
  return hipGetLastError();
}
