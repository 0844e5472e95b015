// opadpo_ctx: the sequence-level entry points of libopadpo_hip.so (include/opadpo_hip.h, "context API").
//
// The seam the reference offers is ONE call - `self.base_model(**inputs)` inside AutoregressivePolicy.forward
// (opadpo/dpo_models/rl_models.py:114-120) - plus `accelerator.backward(loss)` (rl_trainer.py:162) and
// `policy.generate(...)` (rl_models.py:166-183, online_generator.py:292-309).  This file puts the whole schedule behind
// that seam below the C ABI: the context owns the workspace, the saved activations and the KV cache, borrows the weights,
// and sequences the gfx950 kernels of gemm.hip / attention.hip / elementwise.hip / head_optim.hip / decode.hip on the
// caller's stream.  Host code only (plus a handful of index-building kernels); no torch, no exceptions across the boundary.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "common.h"
#include "kernels.h"

namespace {

constexpr size_t ALIGN = 256;
inline size_t up(size_t n) { return (n + ALIGN - 1) / ALIGN * ALIGN; }

// ---- tiny index / bookkeeping kernels (what the Python host did with torch indexing) ---------------------------------------
// rows[k,s,t] = s*Lp + (t == 0 ? pfx - 1 : pfx + k*T + t - 1), labels[k,s,t] = ids[s, n_txt - K*T + k*T + t]
// (rl_models.py:121-123: logits[:, -T-1:-1] predict ids[:, -T:]; packed rows: token 0 of every response is predicted from
// the last prefix position).
__global__ void head_index_kernel(const int32_t* ids, int S, int n_txt, int Lp, int pfx, int K, int T, int32_t* rows, int32_t* labels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * S * T) return;
  const int t = i % T, s = (i / T) % S, k = i / (T * S);
  rows[i] = s * Lp + (t == 0 ? pfx - 1 : pfx + k * T + t - 1);
  labels[i] = ids[(size_t)s * n_txt + (n_txt - K * T) + k * T + t];
}
// idx[b*P + p] = b*(P+1) + p + 1: CLIP hidden state rows without the CLS token
__global__ void drop_cls_index_kernel(int32_t* idx, int B, int P) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * P) idx[i] = i + i / P + 1;
}
__global__ void affine_index_kernel(int32_t* idx, int n, int a, int b) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) idx[i] = a * i + b;
}
__global__ void add_i32_kernel(int32_t* p, int v) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += v; }
__global__ void set_column_f32_kernel(float* x, int ld, int rows, int col, float v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) x[(size_t)i * ld + col] = v;
}
// decode key mask [B, max_ctx]: the prefill's key mask, then ones (a future slot becomes visible when the device-side
// position counter reaches it)
__global__ void decode_mask_kernel(const uint8_t* km_prefill, uint8_t* km, int B, int Lp, int max_ctx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * max_ctx) return;
  const int b = i / max_ctx, p = i % max_ctx;
  km[i] = p < Lp ? km_prefill[(size_t)b * Lp + p] : (uint8_t)1;
}
// post-RoPE k, v of the prefill ([B, Lp, 3, nh, hd] inside the q|k|v projection output) -> head-major caches [B, nh, max_ctx, hd]
__global__ __launch_bounds__(256) void kv_fill_kernel(const bf16_t* qkv, bf16_t* kc, bf16_t* vc, int B, int Lp, int nh, int hd, int max_ctx) {
  const int per = hd / 8;                                  // 16-byte pieces per (position, head)
  const size_t total = (size_t)B * Lp * nh * per;
  const int H = nh * hd;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % per);
    const int h = (int)((i / per) % nh);
    const int p = (int)((i / ((size_t)per * nh)) % Lp);
    const int b = (int)(i / ((size_t)per * nh * Lp));
    const bf16_t* src = qkv + ((size_t)b * Lp + p) * 3 * H + h * hd + c * 8;
    const size_t dst = (((size_t)b * nh + h) * max_ctx + p) * hd + c * 8;
    *(uint4*)(kc + dst) = *(const uint4*)(src + H);
    *(uint4*)(vc + dst) = *(const uint4*)(src + 2 * H);
  }
}
// splice backward (OPA LoRA-SFT stage): d_feats[feat_row[s], p, :] += dX[s*Lp + img_pos(s) + p, :]
__global__ __launch_bounds__(256) void splice_grad_kernel(const float* dX, const int32_t* ids, const int32_t* feat_row, float* d_feats, int S,
                                                         int n_txt, int Lp, int P, int H, int image_token, const int32_t* meta, int stride, int K) {
  const int s = blockIdx.y, p = blockIdx.x;
  __shared__ int pos;
  if (threadIdx.x == 0) {
    int q = 0;
    for (int j = 0; j < n_txt; ++j) if (ids[(size_t)s * n_txt + j] == image_token) { q = j; break; }
    pos = q;
  }
  __syncthreads();
  // ragged rows: sequence s starts at row meta[0] and its first `lead` (meta[2 + K]) positions are not rows
  const size_t row = meta ? (size_t)meta[(size_t)s * stride] + (pos - meta[(size_t)s * stride + 2 + K]) + p : (size_t)s * Lp + pos + p;
  const float* src = dX + row * H;
  float* dst = d_feats + ((size_t)feat_row[s] * P + p) * H;
  for (int c = threadIdx.x; c < H; c += 256) atomicAdd(dst + c, src[c]);
}

// ---- ragged rows: the padding rows (left pad of the query, right pad of every response) are removed from the flat [rows, H]
// buffers of a pass, so no GEMM / norm / SwiGLU / RoPE row is spent on padding.  Per sequence (meta row, META_STRIDE(K) ints):
// [row_start, b_0 .. b_K, lead] - b_0 = end of the prefix, b_a = start of response a, b_K = rows of the sequence (relative), lead =
// dropped left-pad positions.  Padded position of compact row r: r < b_0 -> lead + r; response a, token t = r - b_a -> pfx + a*T + t.
// The meta table travels as by-value kernel arguments (no host staging buffer whose lifetime would have to outlive the launch):
// 960 ints per launch, as many launches as the table needs (one for S*(2K+4) <= 960, i.e. up to 120 packed pairs).
struct MetaBlob { int32_t v[960]; };
__global__ void write_meta_kernel(MetaBlob blob, int32_t* dst, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = blob.v[i];
}
__global__ __launch_bounds__(256) void embed_splice_ragged_kernel(const int32_t* ids, const uint8_t* text_mask, const bf16_t* embed, const bf16_t* feats,
                                                                  const int32_t* feat_row, const uint8_t* image_mask, float* x, uint8_t* key_mask,
                                                                  int32_t* row_pos, const int32_t* meta, int stride, int K, int T, int n_txt, int P, int H,
                                                                  int image_token) {
  __shared__ int img_pos;
  const int s = blockIdx.y, r = blockIdx.x;
  const int32_t* m = meta + (size_t)s * stride;
  const int n_rows = m[1 + K];
  if (r >= n_rows) return;
  if (threadIdx.x == 0) img_pos = n_txt;
  __syncthreads();
  for (int i = threadIdx.x; i < n_txt; i += 256)
    if (ids[(size_t)s * n_txt + i] == image_token) img_pos = i;
  __syncthreads();
  const int ip = img_pos, pfx = n_txt - K * T + P - 1, lead = m[2 + K];
  int pos, rpos;                                   // padded position of the row, RoPE position (responses restart at the prefix end)
  if (r < m[1]) { pos = lead + r; rpos = pos; }
  else {
    int a = 0;
    for (int j = 1; j < K; ++j) if (r >= m[1 + j]) a = j;
    pos = pfx + a * T + (r - m[1 + a]);
    rpos = pfx + (r - m[1 + a]);
  }
  const bf16_t* src;
  uint8_t mk;
  if (pos < ip) {
    src = embed + (size_t)max(ids[(size_t)s * n_txt + pos], 0) * H;
    mk = text_mask[(size_t)s * n_txt + pos];
  } else if (pos < ip + P && ip < n_txt) {
    src = feats + ((size_t)feat_row[s] * P + (pos - ip)) * H;
    mk = image_mask ? image_mask[(size_t)s * P + (pos - ip)] : (uint8_t)1;
  } else {
    const int t = min(pos - (ip < n_txt ? P - 1 : 0), n_txt - 1);
    src = embed + (size_t)max(ids[(size_t)s * n_txt + t], 0) * H;
    mk = text_mask[(size_t)s * n_txt + t];
  }
  const size_t row = (size_t)m[0] + r;
  for (int i = threadIdx.x * 8; i < H; i += 256 * 8) {
    float f[8];
    unpack8(*(const uint4*)(src + i), f);
    *(float4*)(x + row * H + i) = make_float4(f[0], f[1], f[2], f[3]);
    *(float4*)(x + row * H + i + 4) = make_float4(f[4], f[5], f[6], f[7]);
  }
  if (threadIdx.x == 0) { key_mask[row] = mk; row_pos[row] = rpos; }
}
// Head rows of a ragged pass are COMPACT as well: only the cells (k, s, t) with t < valid length of response k get a head row (the
// others are padding: their log-prob is -0.0 / entropy 0 by definition, so no lm_head row, no softmax, no gradient is spent on them -
// 44 % of the [K,S,T] cells of a synthetic seq512 pair).  Cell (k,s,t) -> compact index j = off[s][k] + t (off = exclusive prefix sum of
// the valid lengths in [k][s] order, meta[3 + K + k]); rows[j] = the compact activation row that predicts the token (token 0: the last
// prefix row; token t: row t-1 of the response), labels[j] = the token, cell[j] = the flat [K,S,T] index the result is scattered to.
// urow != nullptr ("compact last layer"): the rows the head reads - per sequence the last prefix row and rows 0 .. v_k - 2 of every
// response - are the ONLY rows of the top decoder layer whose o-projection / MLP output anything reads (their K / V still need every
// row's q|k|v).  They form the list U (ascending; sequence s starts at meta[3 + 2K]); urow[u] = activation row of entry u, and rows[j]
// then holds the U index instead of the activation row: the top layer's o-projection and MLP run on the gathered U rows only.
__global__ void head_index_ragged_kernel(const int32_t* ids, const int32_t* meta, int stride, int S, int n_txt, int K, int T, int32_t* rows,
                                         int32_t* labels, int32_t* cell, int32_t* urow) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= K * S * T) return;
  const int t = i % T, s = (i / T) % S, k = i / (T * S);
  const int32_t* m = meta + (size_t)s * stride;
  const int b0 = m[1], bk = m[1 + k], vk = m[2 + k] - bk;
  if (t >= vk) return;
  const int j = m[3 + K + k] + t;
  const int row = m[0] + (t == 0 ? b0 - 1 : bk + t - 1);
  if (urow) {
    int u = m[3 + 2 * K];
    if (t > 0) {
      u += t;                                            // 1 (the last prefix row) + (t - 1)
      for (int a = 0; a < k; ++a) u += max(m[2 + a] - m[1 + a] - 1, 0);
    }
    urow[u] = row;                                       // token 0 of every response writes the same value
    rows[j] = u;
  } else {
    rows[j] = row;
  }
  labels[j] = ids[(size_t)s * n_txt + (n_txt - K * T) + k * T + t];
  cell[j] = i;
}
// out[i] = fill for every cell, then out[cell[j]] = src[j] (two launches: the fill must be complete first)
__global__ void fill_f32_kernel(float* a, float va, float* b, float vb, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { a[i] = va; b[i] = vb; }
}
__global__ void scatter_cells_kernel(const float* a_c, const float* b_c, const int32_t* cell, float* a, float* b, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) { a[cell[j]] = a_c[j]; b[cell[j]] = b_c[j]; }
}
__global__ void gather_cells_kernel(const float* a, const float* b, const int32_t* cell, float* a_c, float* b_c, int n) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) { a_c[j] = a[cell[j]]; if (b) b_c[j] = b[cell[j]]; }
}

inline dim3 g1(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------------
struct opadpo_saved {
  void* arena = nullptr; size_t bytes = 0;
  int S = 0, L = 0, T = 0, K = 1, M = 0, R = 0, n_txt = 0, train = 0, adapter = -1;
  int seg_prefix = 0, seg_len = 0;
  float temperature = 1.f;
  int nb = 1;                       // per-layer activation slots (n_layers when train, 1 otherwise)
  float* x = nullptr;               // [(nl+1) or 2][M,H] fp32 residual stream
  bf16_t *n1, *qkv, *t_qkv, *attn, *t_o, *n2, *t_gu, *gu, *act, *t_d;
  float *rstd1, *rstd2, *lse, *h;
  uint8_t* key_mask; float* hs; bf16_t* hn; float* rstd_f; float* logits; float* lse_head; float* ent;
  int32_t *rows, *labels;
  int Rc = 0;                       // head rows actually computed: R (padded layout) or the number of valid response cells (ragged)
  int head_chunk = 0;               // > 0: chunked head - the lm_head GEMM, the online log-sum-exp and (backward) the recomputed logits run over
                                    // `head_chunk` vocabulary columns at a time; `logits` is [R, head_chunk], nothing of size [R, vocab] exists
  float *h_s = nullptr, *h_zl = nullptr;      // chunked head: running sum exp / label logit (running max -> lse_head, sum z exp -> ent)
  int32_t* cell = nullptr; float* logp_c = nullptr;      // ragged: flat [K,S,T] index of every compact head row; compact log-probs
  int Uc = 0;                       // > 0: compact last layer - rows of the top layer's o-projection / MLP (head_index_ragged_kernel)
  int32_t* urow = nullptr; bf16_t* attn_u = nullptr;     // U list; attention output gathered on U (saved for the o-projection wgrads)
  const int32_t* ids = nullptr; const int32_t* feat_row = nullptr;   // borrowed (SFT splice backward only)
  // ragged rows (padding removed): M = valid rows of the batch, L = longest sequence; meta / row_pos live in the arena
  int ragged = 0, meta_stride = 0, S_pad_rows = 0;
  int32_t *meta = nullptr, *row_pos = nullptr;
};

struct opadpo_ctx {
  opadpo_dims d;
  int device = 0;
  std::string err;
  int wgrad_det = -1;                             // how the last backward flushed its LoRA wgrads: 1 ordered reduce (bit-reproducible), 0 fp32 atomics, -1 no backward yet
  // borrowed weights
  const bf16_t *embed = nullptr, *norm = nullptr, *lm_head = nullptr, *lm_head_t = nullptr;
  std::vector<opadpo_layer_weights> layers;
  opadpo_vision_weights vis{};
  std::vector<opadpo_vision_layer_weights> vlayers;
  struct Adapter {
    int kind = 0;                                 // 0 = none (bare base model), 1 = LoRA flat buffers, 2 = merged weight copy
    const bf16_t* work = nullptr; const bf16_t* work_t = nullptr; float* grad = nullptr;
    std::vector<opadpo_layer_weights> merged; int swiglu_pair = 0;
  };
  Adapter adapters[OPADPO_MAX_ADAPTERS];
  // rotary tables [pos][hd/2] fp32
  const float *cosb = nullptr, *sinb = nullptr; int rope_len = 0; void* rope_own = nullptr;
  // memory
  opadpo_alloc_fn alloc = nullptr; opadpo_free_fn dealloc = nullptr; void* alloc_user = nullptr;
  struct Block { void* p; size_t bytes; };
  std::vector<Block> cache;                       // released arenas kept for reuse (default allocator only)
  void* ws = nullptr; size_t ws_bytes = 0;        // scratch of vision / backward / decode
  size_t bytes_live = 0, bytes_peak = 0;
  size_t arena_hint[2][16] = {};                  // largest activation arena requested so far, per (no-grad / training pass, responses per row K):
                                                  // CoPO's K = 2 masked pass does not inherit the K = 3 pass's worst case
  std::vector<opadpo_saved*> live;
  // dispatch
  int gemm_variant = -1, use_tr = -1;             // -1: process default (opadpo_set_flags)
  // live profile of the gemm_nt launches (opadpo_ctx_profile)
  bool prof = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
  std::vector<double> prof_flops;
  std::vector<hipEvent_t> prof_pool;
  // decode state (opadpo_decode_begin .. opadpo_decode_step)
  struct Decode {
    bool active = false; void* arena = nullptr; size_t bytes = 0;
    int B = 0, Lp = 0, max_ctx = 0, adapter = 0, max_new = 0;
    bf16_t *kc, *vc; uint8_t* key_mask; float *x, *x2, *hs; bf16_t *hn, *n1, *qkv, *t_qkv, *att, *t_o, *n2, *t_gu, *gu, *act, *t_d, *emb;
    float *hb, *rstd, *logits; int32_t *cur_tok, *step_d, *pos_d; uint8_t* finished; void* ws; size_t ws_bytes;
    float *part_o, *part_d; int split_o = 1, split_d = 1; bool use64 = false;      // K-split partial tiles of the 9..64-token decode GEMMs
    int32_t* history; float temperature; int top_k; float top_p; uint64_t seed; int eos_id, pad_id, suppress_eos;
    hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; hipStream_t cap_stream = nullptr; hipEvent_t ev_in = nullptr, ev_out = nullptr;
  } dec;
};

namespace {

int cfail(opadpo_ctx* c, hipError_t e, const char* where) {
  char buf[512];
  snprintf(buf, sizeof(buf), "%s: %s (%d)", where, hipGetErrorString(e), (int)e);
  if (c) c->err = buf;
  return (int)e;
}
int cbad(opadpo_ctx* c, const char* where, const char* what) {
  if (c) c->err = std::string(where) + ": " + what;
  return (int)hipErrorInvalidValue;
}
#define CK(expr)                                                   \
  do {                                                             \
    hipError_t e_ = (expr);                                        \
    if (e_ != hipSuccess) return cfail(c, e_, __func__);           \
  } while (0)

void* ctx_alloc(opadpo_ctx* c, size_t bytes, hipStream_t st) {
  if (bytes == 0) bytes = ALIGN;
  void* p = nullptr;
  if (c->alloc) {
    p = c->alloc(bytes, (void*)st, c->alloc_user);
  } else {
    int best = -1;
    for (int i = 0; i < (int)c->cache.size(); ++i)
      if (c->cache[i].bytes >= bytes && c->cache[i].bytes <= bytes + bytes / 4 + (1 << 20) && (best < 0 || c->cache[i].bytes < c->cache[best].bytes)) best = i;
    if (best >= 0) {
      p = c->cache[best].p;
      bytes = c->cache[best].bytes;
      c->cache.erase(c->cache.begin() + best);
    } else if (hipMalloc(&p, bytes) != hipSuccess) {
      (void)hipGetLastError();
      for (auto& b : c->cache) (void)hipFree(b.p);        // give cached arenas back and retry once
      c->cache.clear();
      if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
    }
  }
  if (p) { c->bytes_live += bytes; c->bytes_peak = std::max(c->bytes_peak, c->bytes_live); }
  return p;
}
void ctx_free(opadpo_ctx* c, void* p, size_t bytes) {
  if (!p) return;
  c->bytes_live -= std::min(c->bytes_live, bytes);
  if (c->alloc) c->dealloc(p, c->alloc_user);
  else c->cache.push_back({p, bytes});            // same-stream reuse is ordered; opadpo_ctx_trim() returns it to the driver
}
void* ctx_ws(opadpo_ctx* c, size_t bytes, hipStream_t st) {
  if (bytes <= c->ws_bytes) return c->ws;
  if (c->ws) { if (c->alloc) c->dealloc(c->ws, c->alloc_user); else (void)hipFree(c->ws); c->ws = nullptr; c->ws_bytes = 0; }
  if (c->alloc) c->ws = c->alloc(bytes, (void*)st, c->alloc_user);
  else if (hipMalloc(&c->ws, bytes) != hipSuccess) { (void)hipGetLastError(); c->ws = nullptr; }
  c->ws_bytes = c->ws ? bytes : 0;
  return c->ws;
}

// bump carving of one arena
struct Carve {
  char* base; size_t off = 0;
  explicit Carve(void* p) : base((char*)p) {}
  template <typename T> T* take(size_t n) { T* r = base ? (T*)(base + off) : nullptr; off += up(n * sizeof(T)); return r; }
};

GemmNTArgs gemm(const opadpo_ctx* c, const bf16_t* A1, int lda1, const bf16_t* B1, int ldb1, int K1, void* C, int ldc, int out_f32, int M, int N) {
  GemmNTArgs a;
  a.A1 = A1; a.B1 = B1; a.A2 = nullptr; a.B2 = nullptr; a.C = C; a.R = nullptr; a.bias = nullptr;
  a.M = M; a.N = N; a.K1 = K1; a.K2 = 0;
  a.lda1 = lda1; a.ldb1 = ldb1; a.lda2 = 0; a.ldb2 = 0; a.ldc = ldc; a.ldr = 0;
  a.a2_group_n = 0; a.a2_group_stride = 0; a.a1_group_n = 0; a.a1_group_stride = 0;
  a.alpha = 1.f; a.act = 0; a.out_f32 = out_f32; a.r_f32 = 0;
  a.variant = c->gemm_variant;
  return a;
}
inline GemmNTArgs& tail(GemmNTArgs& a, const bf16_t* A2, int lda2, const bf16_t* B2, int ldb2, int K2, int gn = 0, int gs = 0) {
  a.A2 = A2; a.lda2 = lda2; a.B2 = B2; a.ldb2 = ldb2; a.K2 = K2; a.a2_group_n = gn; a.a2_group_stride = gs; return a;
}
inline GemmNTArgs& resid(GemmNTArgs& a, const void* R, int ldr, int r_f32) { a.R = R; a.ldr = ldr; a.r_f32 = r_f32; return a; }

// every gemm_nt launch of the context goes through here: with opadpo_ctx_profile(ctx, 1) each launch is bracketed by HIP events on
// the launch stream (bench.py's live measurement of the dominant kernel: algorithmic FLOPs / measured duration)
hipError_t run_gemm(opadpo_ctx* c, const GemmNTArgs& g, hipStream_t st) {
  if (!c->prof) return launch_gemm_nt(g, st);
  hipEvent_t e0, e1;
  hipError_t e;
  if (c->prof_pool.size() >= 2) {
    e0 = c->prof_pool.back(); c->prof_pool.pop_back();
    e1 = c->prof_pool.back(); c->prof_pool.pop_back();
  } else {
    if ((e = hipEventCreate(&e0)) != hipSuccess) return e;
    if ((e = hipEventCreate(&e1)) != hipSuccess) return e;
  }
  if ((e = hipEventRecord(e0, st)) != hipSuccess) return e;
  e = launch_gemm_nt(g, st);
  (void)hipEventRecord(e1, st);
  c->prof_ev.push_back({e0, e1});
  c->prof_flops.push_back(2.0 * (double)g.M * (double)g.N * (double)(g.K1 + g.K2));
  return e;
}

// flat LoRA buffer: element offsets of the fused blocks of one decoder layer (model.py lora_blocks() order)
struct LoraOff { size_t a_qkv, b_qkv, a_o, b_o, a_gu, b_gu, a_d, b_d, layer; };
LoraOff lora_off(const opadpo_dims& d) {
  const size_t H = d.hidden, F = d.ffn, r = d.lora_r;
  LoraOff o; size_t p = 0;
  o.a_qkv = p; p += 3 * r * H;  o.b_qkv = p; p += 3 * H * r;
  o.a_o = p;   p += r * H;      o.b_o = p;   p += H * r;
  o.a_gu = p;  p += 2 * r * H;  o.b_gu = p;  p += 2 * F * r;
  o.a_d = p;   p += r * F;      o.b_d = p;   p += H * r;
  o.layer = p;
  return o;
}

hipError_t ensure_rope(opadpo_ctx* c, int len, hipStream_t st) {
  if (c->cosb && c->rope_len >= len) return hipSuccess;
  if (c->cosb && !c->rope_own) return hipErrorInvalidValue;      // caller-provided tables are too short
  // HF Llama rotary (modeling_llama.py LlamaRotaryEmbedding): inv_freq = theta^(-2i/hd) (fp32), angle = pos * inv_freq (fp32)
  const int half = c->d.head_dim / 2;
  const int n = std::max(len, 2048);
  std::vector<float> hc((size_t)n * half), hs((size_t)n * half);
  for (int i = 0; i < half; ++i) {
    const float inv = 1.0f / powf(c->d.rope_theta, (float)(2 * i) / (float)c->d.head_dim);
    for (int p = 0; p < n; ++p) {
      const float ang = (float)p * inv;
      hc[(size_t)p * half + i] = (float)cos((double)ang);
      hs[(size_t)p * half + i] = (float)sin((double)ang);
    }
  }
  if (c->rope_own) (void)hipFree(c->rope_own);
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, 2 * hc.size() * sizeof(float));
  if (e != hipSuccess) return e;
  e = hipMemcpy(p, hc.data(), hc.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy((float*)p + hc.size(), hs.data(), hs.size() * sizeof(float), hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(p); return e; }
  c->rope_own = p; c->cosb = (float*)p; c->sinb = (float*)p + hc.size(); c->rope_len = n;
  (void)st;
  return hipSuccess;
}

// ---- one Llama decoder layer over M = S*Lp rows (model.py layer_fwd / mlp_fwd) ------------------------------------------------
struct LayerBufs { bf16_t *n1, *qkv, *t_qkv, *attn, *t_o, *n2, *t_gu, *gu, *act, *t_d; float *rstd1, *rstd2, *lse, *h; };

// full-sequence passes: residual adds in the o / down projections' epilogues (default) or deferred to the RMSNorm that follows (context flag bit 13;
// OPADPO_FUSE_RESID=0 makes that the process default - A/B runs)
static bool fuse_resid(const opadpo_ctx* c) {
  static const int env = getenv("OPADPO_FUSE_RESID") ? atoi(getenv("OPADPO_FUSE_RESID")) : 1;
  return c->use_tr >= 0 ? !(c->use_tr & 8192) && env != 0 : env != 0;
}
static bool fuse_swiglu_bwd(const opadpo_ctx* c) {
  static const int env = getenv("OPADPO_FUSE_SWIGLU_BWD") ? atoi(getenv("OPADPO_FUSE_SWIGLU_BWD")) : 1;
  return c->use_tr >= 0 ? !(c->use_tr & 16384) && env != 0 : env != 0;
}
// ydef != nullptr ("deferred residual", the full-sequence passes): the down projection writes its fp32 product to ydef WITHOUT the
// residual - the RMSNorm that follows adds it (rmsnorm_sum_fwd: x = h + y in the same fp32 arithmetic, so h / x keep their bits).
// Rounds 2-4 ran the full-sequence passes this way because a residual operand went through the LDS-STAGED epilogue of the 256x256 tile, every load
// behind the read-back of its row: +0.21-0.28 ms on a 0.6-1.6 ms GEMM (tools/resid_probe.py), while the norm kernel takes the same bytes at
// 6 TB/s (+0.13 ms).  Round 5: the DIRECT epilogue requests the residual rows one row block ahead of the accumulator read-out (+0.05-0.065 ms per
// GEMM whatever K, profiles/r05m_resid_probe_k.txt), so the add lives in the projections again (fuse_resid above) and this form is the A/B switch.
// ydef == nullptr (decode steps: a few rows; fused-residual full-sequence passes): residual in the epilogue, result in xo.
hipError_t mlp_fwd(opadpo_ctx* c, int i, const opadpo_ctx::Adapter& ad, const float* h, float* xo, const LayerBufs& b, int M, int stream_hint, hipStream_t st,
                   float* ydef = nullptr, bool norm_done = false) {
  const opadpo_dims& d = c->d;
  const int H = d.hidden, F = d.ffn, r = d.lora_r;
  const float s = d.lora_alpha / d.lora_r;
  const opadpo_layer_weights& w0 = c->layers[i];
  const bool merged = ad.kind == 2;
  const opadpo_layer_weights& w = merged ? ad.merged[i] : w0;
  const LoraOff o = lora_off(d);
  const bf16_t* lw = ad.kind == 1 ? ad.work + (size_t)i * o.layer : nullptr;
  hipError_t e;
  if (!norm_done && (e = launch_rmsnorm_fwd(h, 1, w0.ln2, b.n2, b.rstd2, M, H, d.rms_eps, st)) != hipSuccess) return e;
  bool have_gu = true;
  if (lw) {
    GemmNTArgs g1_ = gemm(c, b.n2, H, lw + o.a_gu, H, H, b.t_gu, 2 * r, 0, M, 2 * r); g1_.alpha = s; g1_.act |= stream_hint;
    if ((e = run_gemm(c, g1_, st)) != hipSuccess) return e;
    GemmNTArgs g2 = gemm(c, b.n2, H, w.wgu, H, H, b.gu, 2 * F, 0, M, 2 * F); tail(g2, b.t_gu, 2 * r, lw + o.b_gu, r, r, F, r); g2.act |= stream_hint;
    if ((e = run_gemm(c, g2, st)) != hipSuccess) return e;
  } else if (merged && ad.swiglu_pair) {        // SwiGLU fused into the projection's epilogue (rows per 128 = [64 gate | 64 up])
    GemmNTArgs g2 = gemm(c, b.n2, H, w.wgu, H, H, b.act, F, 0, M, 2 * F); g2.act = OPADPO_ACT_SWIGLU_PAIR | stream_hint;
    if ((e = run_gemm(c, g2, st)) != hipSuccess) return e;
    have_gu = false;
  } else {
    GemmNTArgs g2 = gemm(c, b.n2, H, w.wgu, H, H, b.gu, 2 * F, 0, M, 2 * F); g2.act |= stream_hint;
    if ((e = run_gemm(c, g2, st)) != hipSuccess) return e;
  }
  if (have_gu && (e = launch_silu_mul_fwd(b.gu, b.act, M, F, st)) != hipSuccess) return e;
  if (lw) {
    GemmNTArgs g3 = gemm(c, b.act, F, lw + o.a_d, F, F, b.t_d, r, 0, M, r); g3.alpha = s; g3.act |= stream_hint;
    if ((e = run_gemm(c, g3, st)) != hipSuccess) return e;
    GemmNTArgs g4 = gemm(c, b.act, F, w.wd, F, F, ydef ? ydef : xo, H, 1, M, H); tail(g4, b.t_d, r, lw + o.b_d, r, r); g4.act |= stream_hint;
    if (!ydef) resid(g4, h, H, 1);
    return run_gemm(c, g4, st);
  }
  GemmNTArgs g4 = gemm(c, b.act, F, w.wd, F, F, ydef ? ydef : xo, H, 1, M, H); g4.act |= stream_hint;
  if (!ydef) resid(g4, h, H, 1);
  return run_gemm(c, g4, st);
}

struct Rag { const int32_t* meta; int stride, n_seg, rows; const int32_t* row_pos; };      // ragged rows of a pass (nullptr = padded)
// top decoder layer on the rows the head reads only (opadpo_saved::Uc): U list, gathered attention output (kept), gathered input (scratch)
struct TopRows { const int32_t* urow; int n; bf16_t* attn_u; float* x_u; };

// Input of the layer: x = res (+ yin when yin != nullptr: the previous layer's down-projection product, residual deferred).  x is written
// to `x` (may alias res when yin is null), the layer leaves h = x + attention branch in b.h and its own down-projection product in Y:
// the caller hands (b.h, Y) to the next layer / the final norm.
hipError_t layer_fwd(opadpo_ctx* c, int i, const opadpo_ctx::Adapter& ad, const float* res, const float* yin, float* x, float* Y, const LayerBufs& b, int S, int Lp,
                     const uint8_t* key_mask, int seg0, int seg1, bf16_t* kc, bf16_t* vc, int max_ctx, hipStream_t st, const Rag* rg = nullptr,
                     const TopRows* top = nullptr) {
  const opadpo_dims& d = c->d;
  const int H = d.hidden, r = d.lora_r, nh = d.n_heads, hd = d.head_dim;
  const int M = rg ? rg->rows : S * Lp;
  const float s = d.lora_alpha / d.lora_r;
  const opadpo_layer_weights& w0 = c->layers[i];
  const opadpo_layer_weights& w = ad.kind == 2 ? ad.merged[i] : w0;
  const LoraOff o = lora_off(d);
  const bf16_t* lw = ad.kind == 1 ? ad.work + (size_t)i * o.layer : nullptr;
  hipError_t e;
  const size_t MH = (size_t)M * H;
  if (yin) e = launch_rmsnorm_sum_fwd(res, 1, yin, 1, MH, w0.ln1, x, b.n1, b.rstd1, M, H, d.rms_eps, st);
  else e = (res == x) ? launch_rmsnorm_fwd(x, 1, w0.ln1, b.n1, b.rstd1, M, H, d.rms_eps, st) : hipErrorInvalidValue;
  if (e != hipSuccess) return e;
  // rotary embedding: on ragged rows (per-row positions at hand) inside the q|k|v projection's epilogue, table-free - one 130-us pass over
  // q and k less per layer and pass; context flag bit 11 keeps the separate in-place kernel (also: padded rows, head_dim 64, small problems
  // that the 256x256 kernel does not take)
  const bool fuse_rope = rg && hd == 128 && (3 * H) % 256 == 0 && !(c->use_tr >= 0 && (c->use_tr & 2048)) &&
                         (double)M * H * 2 < 4.0e9 && ((M + 255) / 256) * (3 * H / 256) >= 150;
  auto with_rope = [&](GemmNTArgs& g) { if (fuse_rope) { g.rope_pos = rg->row_pos; g.rope_l2theta = log2f(d.rope_theta); g.rope_cols = 2 * H; } };
  if (lw) {
    GemmNTArgs g1_ = gemm(c, b.n1, H, lw + o.a_qkv, H, H, b.t_qkv, 3 * r, 0, M, 3 * r); g1_.alpha = s;
    if ((e = run_gemm(c, g1_, st)) != hipSuccess) return e;
    GemmNTArgs g2 = gemm(c, b.n1, H, w.wqkv, H, H, b.qkv, 3 * H, 0, M, 3 * H); tail(g2, b.t_qkv, 3 * r, lw + o.b_qkv, r, r, H, r);
    with_rope(g2);
    if ((e = run_gemm(c, g2, st)) != hipSuccess) return e;
  } else {
    GemmNTArgs g2 = gemm(c, b.n1, H, w.wqkv, H, H, b.qkv, 3 * H, 0, M, 3 * H);
    with_rope(g2);
    if ((e = run_gemm(c, g2, st)) != hipSuccess) return e;
  }
  // ragged passes that are too small (or too odd) for the fused epilogue rotate with the SAME table-free angles and arithmetic: the bits of a
  // row's q / k do not depend on the size of the batch around it, and the backward (always table-free on ragged rows) is the exact transpose
  if (!fuse_rope && (e = launch_rope(b.qkv, 3 * H, c->cosb, c->sinb, M, Lp, 2 * nh, hd, 0, nullptr, seg0, seg1, st, rg ? rg->row_pos : nullptr,
                                     rg ? log2f(d.rope_theta) : 0.f)) != hipSuccess) return e;
  if (kc) {                                        // rollout prefill: post-RoPE k, v -> head-major KV cache
    hipLaunchKernelGGL(kv_fill_kernel, dim3(std::min<size_t>(4096, ((size_t)M * nh * (hd / 8) + 255) / 256)), dim3(256), 0, st, b.qkv, kc, vc, S, Lp, nh, hd, max_ctx);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  AttnArgs a;
  memset(&a, 0, sizeof(a));
  a.q = b.qkv; a.k = b.qkv + H; a.v = b.qkv + 2 * H; a.o = b.attn; a.lse = b.lse; a.key_mask = key_mask;
  a.S = S; a.L = Lp; a.nh = nh; a.hd = hd; a.ld = 3 * H; a.ldo = H; a.causal = 1 | OPADPO_ATTN_SKIP_MASKED_Q; a.scale = 1.0f / sqrtf((float)hd);
  a.seg_prefix = seg0; a.seg_len = seg1; a.use_tr = c->use_tr;
  if (rg) { a.seq_meta = rg->meta; a.meta_stride = rg->stride; a.n_seg = rg->n_seg; a.rows_total = rg->rows; a.seg_prefix = 0; a.seg_len = 0; }
  if ((e = launch_attn_fwd(a, st)) != hipSuccess) return e;
  // top layer with `top`: from here on only the rows the head reads exist (attention needed every row's k, v; nothing reads the
  // o-projection / MLP output of the other rows).  b.t_o / b.h / b.n2 / ... and Y then hold top->n COMPACT rows.
  int Mo = M;
  const bf16_t* attn = b.attn;
  const float* xr = x;
  if (top) {
    if ((e = launch_gather_rows(b.attn, H, top->urow, top->attn_u, top->n, H, st)) != hipSuccess) return e;
    if ((e = launch_gather_rows((const bf16_t*)x, 2 * H, top->urow, (bf16_t*)top->x_u, top->n, 2 * H, st)) != hipSuccess) return e;      // fp32 rows = 2H bf16 units
    Mo = top->n; attn = top->attn_u; xr = top->x_u;
  }
  // fused residual (round 5, default; context flag bit 13 keeps the deferred form): h = x + attn . Wo leaves the o projection's DIRECT epilogue
  // (fp32 residual rows requested one row block ahead of the accumulator read-out, gemm.hip w4_direct_epilogue) and the layer's output
  // x' = h + down(..) the down projection's, written to Y - which the caller makes the next layer's x.  Same fp32 additions on the same
  // operands as rmsnorm_sum_fwd's: h / x keep their bits; the norm passes read 6 instead of 14 bytes per element.
  const bool fr = fuse_resid(c);
  float* const o_dst = fr ? b.h : Y;
  if (lw) {
    GemmNTArgs g3 = gemm(c, attn, H, lw + o.a_o, H, H, b.t_o, r, 0, Mo, r); g3.alpha = s;
    if ((e = run_gemm(c, g3, st)) != hipSuccess) return e;
    GemmNTArgs g4 = gemm(c, attn, H, w.wo, H, H, o_dst, H, 1, Mo, H); tail(g4, b.t_o, r, lw + o.b_o, r, r);
    if (fr) resid(g4, xr, H, 1);
    if ((e = run_gemm(c, g4, st)) != hipSuccess) return e;
  } else {
    GemmNTArgs g4 = gemm(c, attn, H, w.wo, H, H, o_dst, H, 1, Mo, H);
    if (fr) resid(g4, xr, H, 1);
    if ((e = run_gemm(c, g4, st)) != hipSuccess) return e;
  }
  if (fr) {
    if ((e = launch_rmsnorm_fwd(b.h, 1, w0.ln2, b.n2, b.rstd2, Mo, H, d.rms_eps, st)) != hipSuccess) return e;      // n2 = norm(h)
    return mlp_fwd(c, i, ad, b.h, Y, b, Mo, 0, st, nullptr, true);                                                   // Y = h + down(..)
  }
  if ((e = launch_rmsnorm_sum_fwd(xr, 1, Y, 1, (size_t)Mo * H, w0.ln2, b.h, b.n2, b.rstd2, Mo, H, d.rms_eps, st)) != hipSuccess) return e;      // h = x + attn . Wo, n2 = norm(h)
  return mlp_fwd(c, i, ad, b.h, nullptr, b, Mo, 0, st, Y, true);
}

size_t saved_layout(const opadpo_dims& d, opadpo_saved* sv, void* base) {
  const size_t M = sv->M, H = d.hidden, F = d.ffn, r = d.lora_r, nl = d.n_layers, nb = sv->nb, R = sv->R;
  Carve cv(base);
  sv->x = cv.take<float>((sv->train ? nl + 1 : 3) * M * H);      // x_0 .. x_{nl-1} (train) or two alternating slots, + ONE slot for the branch product Y
  sv->n1 = cv.take<bf16_t>(nb * M * H);
  sv->rstd1 = cv.take<float>(nb * M);
  sv->qkv = cv.take<bf16_t>(nb * M * 3 * H);
  sv->t_qkv = cv.take<bf16_t>(nb * M * 3 * r);
  sv->attn = cv.take<bf16_t>(nb * M * H);
  sv->lse = cv.take<float>(nb * (size_t)sv->S * d.n_heads * sv->L);
  sv->t_o = cv.take<bf16_t>(nb * M * r);
  sv->h = cv.take<float>(nb * M * H);
  sv->n2 = cv.take<bf16_t>(nb * M * H);
  sv->rstd2 = cv.take<float>(nb * M);
  sv->t_gu = cv.take<bf16_t>(nb * M * 2 * r);
  sv->gu = cv.take<bf16_t>(nb * M * 2 * F);
  sv->act = cv.take<bf16_t>(nb * M * F);
  sv->t_d = cv.take<bf16_t>(nb * M * r);
  sv->key_mask = cv.take<uint8_t>((size_t)sv->S * sv->L);
  sv->hs = cv.take<float>(R * H);
  sv->hn = cv.take<bf16_t>(R * H);
  sv->rstd_f = cv.take<float>(R);
  sv->logits = cv.take<float>(R * (size_t)std::max(sv->head_chunk > 0 ? sv->head_chunk : d.vocab, d.hidden));      // also parks the head rows of the deferred branch product ([R,H]) before the lm_head GEMM writes it
  sv->h_s = cv.take<float>(sv->head_chunk > 0 ? R : 1);
  sv->h_zl = cv.take<float>(sv->head_chunk > 0 ? R : 1);
  sv->lse_head = cv.take<float>(R);
  sv->ent = cv.take<float>(R);
  sv->rows = cv.take<int32_t>(R);
  sv->labels = cv.take<int32_t>(R);
  sv->cell = cv.take<int32_t>(sv->ragged ? R : 1);
  sv->logp_c = cv.take<float>(sv->ragged ? R : 1);
  sv->urow = cv.take<int32_t>(sv->ragged ? R : 1);                  // |U| <= valid tokens <= R
  sv->attn_u = cv.take<bf16_t>(sv->ragged ? std::min(R, M) * H : 1);
  sv->meta = cv.take<int32_t>(sv->ragged ? (size_t)sv->S * sv->meta_stride : 1);
  sv->row_pos = cv.take<int32_t>(sv->ragged ? M : 1);
  return cv.off;
}

LayerBufs slot(const opadpo_dims& d, const opadpo_saved* sv, int k) {
  const size_t M = sv->M, H = d.hidden, F = d.ffn, r = d.lora_r;
  LayerBufs b;
  b.n1 = sv->n1 + k * M * H; b.rstd1 = sv->rstd1 + k * M; b.qkv = sv->qkv + k * M * 3 * H; b.t_qkv = sv->t_qkv + k * M * 3 * r;
  b.attn = sv->attn + k * M * H; b.lse = sv->lse + (size_t)k * sv->S * d.n_heads * sv->L; b.t_o = sv->t_o + k * M * r;
  b.h = sv->h + k * M * H; b.n2 = sv->n2 + k * M * H; b.rstd2 = sv->rstd2 + k * M; b.t_gu = sv->t_gu + k * M * 2 * r;
  b.gu = sv->gu + k * M * 2 * F; b.act = sv->act + k * M * F; b.t_d = sv->t_d + k * M * r;
  return b;
}

void saved_destroy(opadpo_ctx* c, opadpo_saved* sv) {
  if (!sv) return;
  ctx_free(c, sv->arena, sv->bytes);
  c->live.erase(std::remove(c->live.begin(), c->live.end(), sv), c->live.end());
  delete sv;
}

bool weights_ready(const opadpo_ctx* c) { return c->embed && c->norm && c->lm_head && (int)c->layers.size() == c->d.n_layers; }

}  // namespace

// ===========================================================================================================================
extern "C" {

int opadpo_ctx_create(const opadpo_dims* dims, int device, opadpo_ctx** out) {
  if (!dims || !out) return (int)hipErrorInvalidValue;
  const opadpo_dims& d = *dims;
  if (d.hidden != d.n_heads * d.head_dim || (d.head_dim != 64 && d.head_dim != 128) || d.hidden % 128 || d.ffn % 128 || d.vocab % 128 ||
      d.lora_r % 128 || d.n_layers <= 0)
    return (int)hipErrorInvalidValue;
  opadpo_ctx* c = new (std::nothrow) opadpo_ctx();
  if (!c) return (int)hipErrorOutOfMemory;
  c->d = d; c->device = device;
  *out = c;
  return 0;
}

void opadpo_ctx_destroy(opadpo_ctx* c) {
  if (!c) return;
  opadpo_decode_end(c);
  while (!c->live.empty()) saved_destroy(c, c->live.back());
  for (auto& b : c->cache) (void)hipFree(b.p);
  if (c->ws) { if (c->alloc) c->dealloc(c->ws, c->alloc_user); else (void)hipFree(c->ws); }
  if (c->rope_own) (void)hipFree(c->rope_own);
  for (auto& pr : c->prof_ev) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
  for (auto& ev : c->prof_pool) (void)hipEventDestroy(ev);
  if (c->dec.cap_stream) (void)hipStreamDestroy(c->dec.cap_stream);
  if (c->dec.ev_in) (void)hipEventDestroy(c->dec.ev_in);
  if (c->dec.ev_out) (void)hipEventDestroy(c->dec.ev_out);
  delete c;
}

const char* opadpo_ctx_last_error(const opadpo_ctx* c) { return c ? c->err.c_str() : "null context"; }

int opadpo_ctx_set_allocator(opadpo_ctx* c, opadpo_alloc_fn alloc, opadpo_free_fn dealloc, void* user) {
  if (!c) return (int)hipErrorInvalidValue;
  if ((alloc == nullptr) != (dealloc == nullptr)) return cbad(c, __func__, "alloc and free must be given together");
  if (c->bytes_live || c->ws) return cbad(c, __func__, "the allocator can only change while the context owns no memory");
  for (auto& b : c->cache) (void)hipFree(b.p);
  c->cache.clear();
  c->alloc = alloc; c->dealloc = dealloc; c->alloc_user = user;
  return 0;
}

int opadpo_ctx_set_flags(opadpo_ctx* c, int gemm_variant, int use_tr) {
  if (!c) return (int)hipErrorInvalidValue;
  c->gemm_variant = gemm_variant; c->use_tr = use_tr;
  return 0;
}

int opadpo_ctx_trim(opadpo_ctx* c) {
  if (!c) return (int)hipErrorInvalidValue;
  for (auto& b : c->cache) (void)hipFree(b.p);
  c->cache.clear();
  if (c->ws) { if (c->alloc) c->dealloc(c->ws, c->alloc_user); else (void)hipFree(c->ws); c->ws = nullptr; c->ws_bytes = 0; }
  // the "largest activation arena seen so far" hints go too: after a trim the next pass sizes its arena for ITS batch shape (round 6: a 22-pair
  // training forward followed by a 2-pair one kept asking for the 22-pair arena - 155 GB at 7B - next to whatever else the host had allocated since)
  for (auto& row : c->arena_hint) for (auto& h : row) h = 0;
  return 0;
}

size_t opadpo_ctx_bytes_peak(const opadpo_ctx* c) { return c ? c->bytes_peak : 0; }
int opadpo_ctx_wgrad_deterministic(const opadpo_ctx* c) { return c ? c->wgrad_det : -1; }

int opadpo_ctx_profile(opadpo_ctx* c, int enable) {
  if (!c) return (int)hipErrorInvalidValue;
  c->prof = enable != 0;
  return 0;
}

int opadpo_ctx_profile_read(opadpo_ctx* c, double* flops, double* ms, int64_t* launches) {
  if (!c) return (int)hipErrorInvalidValue;
  double f = 0, t = 0;
  for (size_t i = 0; i < c->prof_ev.size(); ++i) {
    CK(hipEventSynchronize(c->prof_ev[i].second));
    float dt = 0.f;
    CK(hipEventElapsedTime(&dt, c->prof_ev[i].first, c->prof_ev[i].second));
    t += dt; f += c->prof_flops[i];
    c->prof_pool.push_back(c->prof_ev[i].first);
    c->prof_pool.push_back(c->prof_ev[i].second);
  }
  if (flops) *flops = f;
  if (ms) *ms = t;
  if (launches) *launches = (int64_t)c->prof_ev.size();
  c->prof_ev.clear();
  c->prof_flops.clear();
  return 0;
}

int opadpo_ctx_set_llm_weights(opadpo_ctx* c, const uint16_t* embed, const uint16_t* final_norm, const uint16_t* lm_head,
                               const uint16_t* lm_head_t, const opadpo_layer_weights* layers, int n_layers) {
  if (!c) return (int)hipErrorInvalidValue;
  if (!embed || !final_norm || !lm_head || !layers || n_layers != c->d.n_layers) return cbad(c, __func__, "null pointer or layer count != dims.n_layers");
  for (int i = 0; i < n_layers; ++i)
    if (!layers[i].wqkv || !layers[i].wo || !layers[i].wgu || !layers[i].wd || !layers[i].ln1 || !layers[i].ln2) return cbad(c, __func__, "null layer weight");
  c->embed = embed; c->norm = final_norm; c->lm_head = lm_head; c->lm_head_t = lm_head_t;
  c->layers.assign(layers, layers + n_layers);
  return 0;
}

int opadpo_ctx_set_vision_weights(opadpo_ctx* c, const opadpo_vision_weights* w, const opadpo_vision_layer_weights* layers, int n_layers) {
  if (!c) return (int)hipErrorInvalidValue;
  if (!w || !layers || n_layers != c->d.v_used_layers) return cbad(c, __func__, "null pointer or layer count != dims.v_used_layers");
  c->vis = *w;
  c->vlayers.assign(layers, layers + n_layers);
  return 0;
}

int opadpo_ctx_set_rope_tables(opadpo_ctx* c, const float* cos_tab, const float* sin_tab, int n_pos) {
  if (!c) return (int)hipErrorInvalidValue;
  if (c->rope_own) { (void)hipFree(c->rope_own); c->rope_own = nullptr; }
  c->cosb = cos_tab; c->sinb = sin_tab; c->rope_len = cos_tab ? n_pos : 0;
  return 0;
}

int opadpo_ctx_set_adapter(opadpo_ctx* c, int id, const uint16_t* work, const uint16_t* work_t, float* grad) {
  if (!c || id < 0 || id >= OPADPO_MAX_ADAPTERS) return c ? cbad(c, __func__, "adapter id out of range") : (int)hipErrorInvalidValue;
  opadpo_ctx::Adapter& a = c->adapters[id];
  a.kind = work ? 1 : 0; a.work = work; a.work_t = work_t; a.grad = grad; a.merged.clear(); a.swiglu_pair = 0;
  return 0;
}

int opadpo_ctx_set_merged_adapter(opadpo_ctx* c, int id, const opadpo_layer_weights* merged, int n_layers, int swiglu_pair) {
  if (!c || id < 0 || id >= OPADPO_MAX_ADAPTERS) return c ? cbad(c, __func__, "adapter id out of range") : (int)hipErrorInvalidValue;
  if (!merged || n_layers != c->d.n_layers) return cbad(c, __func__, "null pointer or layer count != dims.n_layers");
  for (int i = 0; i < n_layers; ++i) if (!merged[i].wgu) return cbad(c, __func__, "merged layer needs at least the gate|up weight");
  if (swiglu_pair && c->d.ffn % 128) return cbad(c, __func__, "SwiGLU-pair layout needs ffn % 128 == 0");
  opadpo_ctx::Adapter& a = c->adapters[id];
  a.kind = 2; a.work = a.work_t = nullptr; a.grad = nullptr; a.swiglu_pair = swiglu_pair;
  a.merged.assign(merged, merged + n_layers);
  for (int i = 0; i < n_layers; ++i) {            // projections a merged copy does not carry fall back to the base weights
    opadpo_layer_weights& m = a.merged[i];
    const opadpo_layer_weights& w = c->layers.empty() ? m : c->layers[i];
    if (!m.wqkv) m.wqkv = w.wqkv;
    if (!m.wo) m.wo = w.wo;
    if (!m.wd) m.wd = w.wd;
  }
  return 0;
}

// ---- vision: CLIP-ViT hidden_states[-2][:, 1:] -> mlp2x_gelu (model.py encode_images) -----------------------------------------
int opadpo_vision_encode(opadpo_ctx* c, const uint16_t* pixels, int B, uint16_t* feats, void* stream) {
  if (!c) return (int)hipErrorInvalidValue;
  if (!pixels || !feats || B <= 0) return cbad(c, __func__, "null operand or empty batch");
  if ((int)c->vlayers.size() != c->d.v_used_layers || !c->vis.patch_w) return cbad(c, __func__, "vision weights not set");
  hipStream_t st = (hipStream_t)stream;
  const opadpo_dims& d = c->d;
  const int side = d.image_size / d.patch, P = side * side, vh = d.v_hidden, vf = d.v_ffn, T = P + 1, M = B * T;
  const int kpad = (3 * d.patch * d.patch + 63) / 64 * 64, hd = vh / d.v_heads, H = d.hidden;
  size_t need = 0;
  { Carve cv(nullptr); cv.take<bf16_t>((size_t)B * P * kpad); cv.take<bf16_t>((size_t)B * P * vh); cv.take<float>((size_t)M * vh); cv.take<float>((size_t)M * vh);
    cv.take<bf16_t>((size_t)M * vh); cv.take<bf16_t>((size_t)M * 3 * vh); cv.take<bf16_t>((size_t)M * vh); cv.take<bf16_t>((size_t)M * vf);
    cv.take<int32_t>((size_t)B * P); cv.take<bf16_t>((size_t)B * P * vh); cv.take<bf16_t>((size_t)B * P * H); need = cv.off; }
  void* base = ctx_ws(c, need, st);
  if (!base) return cfail(c, hipErrorOutOfMemory, __func__);
  Carve cv(base);
  bf16_t* cols = cv.take<bf16_t>((size_t)B * P * kpad); bf16_t* patches = cv.take<bf16_t>((size_t)B * P * vh);
  // the tower's residual stream x is fp32 (round 4): every block adds two bf16-rounded branch outputs to it, and rounding x itself to
  // bf16 at each of the 46 adds carried a quarter of the 8-layer log-prob error (profiles/r03_bf16_ablation.json) for 0.5 % of the FLOPs
  float* x = cv.take<float>((size_t)M * vh); float* x2 = cv.take<float>((size_t)M * vh); bf16_t* n = cv.take<bf16_t>((size_t)M * vh);
  bf16_t* qkv = cv.take<bf16_t>((size_t)M * 3 * vh); bf16_t* att = cv.take<bf16_t>((size_t)M * vh); bf16_t* f1 = cv.take<bf16_t>((size_t)M * vf);
  int32_t* idx = cv.take<int32_t>((size_t)B * P); bf16_t* tok = cv.take<bf16_t>((size_t)B * P * vh); bf16_t* h0 = cv.take<bf16_t>((size_t)B * P * H);
  CK(launch_im2col(pixels, cols, B, d.image_size, d.patch, kpad, st));
  { GemmNTArgs g = gemm(c, cols, kpad, c->vis.patch_w, kpad, kpad, patches, vh, 0, B * P, vh); CK(run_gemm(c, g, st)); }
  CK(launch_vision_embed(patches, c->vis.cls, c->vis.pos, x, B, P, vh, st, 1));
  CK(launch_layernorm_fwd(x, c->vis.pre_ln_w, c->vis.pre_ln_b, x2, M, vh, d.v_eps, st, 1, 1));
  std::swap(x, x2);
  for (const opadpo_vision_layer_weights& w : c->vlayers) {
    CK(launch_layernorm_fwd(x, w.ln1_w, w.ln1_b, n, M, vh, d.v_eps, st, 1, 0));
    { GemmNTArgs g = gemm(c, n, vh, w.wqkv, vh, vh, qkv, 3 * vh, 0, M, 3 * vh); g.bias = w.bqkv; CK(run_gemm(c, g, st)); }
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = qkv; a.k = qkv + vh; a.v = qkv + 2 * vh; a.o = att; a.S = B; a.L = T; a.nh = d.v_heads; a.hd = hd; a.ld = 3 * vh; a.ldo = vh;
    a.causal = 0; a.scale = 1.0f / sqrtf((float)hd); a.use_tr = c->use_tr;
    CK(launch_attn_fwd(a, st));
    { GemmNTArgs g = gemm(c, att, vh, w.wo, vh, vh, x2, vh, 1, M, vh); g.bias = w.bo; resid(g, x, vh, 1); CK(run_gemm(c, g, st)); }
    CK(launch_layernorm_fwd(x2, w.ln2_w, w.ln2_b, n, M, vh, d.v_eps, st, 1, 0));
    { GemmNTArgs g = gemm(c, n, vh, w.fc1, vh, vh, f1, vf, 0, M, vf); g.bias = w.b1; g.act = OPADPO_ACT_QUICK_GELU; CK(run_gemm(c, g, st)); }
    { GemmNTArgs g = gemm(c, f1, vf, w.fc2, vf, vf, x, vh, 1, M, vh); g.bias = w.b2; resid(g, x2, vh, 1); CK(run_gemm(c, g, st)); }
  }
  hipLaunchKernelGGL(drop_cls_index_kernel, g1((size_t)B * P), dim3(256), 0, st, idx, B, P);
  CK(hipGetLastError());
  CK(launch_f32_to_bf16(x, n, (size_t)M * vh, st));          // one rounding of hidden_states[-2]: the bf16 operand of the projector
  CK(launch_gather_rows(n, vh, idx, tok, B * P, vh, st));
  { GemmNTArgs g = gemm(c, tok, vh, c->vis.proj0, vh, vh, h0, H, 0, B * P, H); g.bias = c->vis.proj0_b; g.act = OPADPO_ACT_GELU; CK(run_gemm(c, g, st)); }
  { GemmNTArgs g = gemm(c, h0, H, c->vis.proj2, H, H, feats, H, 0, B * P, H); g.bias = c->vis.proj2_b; CK(run_gemm(c, g, st)); }
  return 0;
}

// ---- sequence log-probs forward (model.py seq_logprobs_fwd; rl_models.py:114-132) ---------------------------------------------
int opadpo_seq_logprobs_fwd(opadpo_ctx* c, int adapter_id, const int32_t* ids, const uint8_t* text_mask, const int32_t* feat_row,
                            const uint8_t* image_mask, const uint16_t* feats, int S, int n_txt, int T, int K, float temperature, int train,
                            float* logp, float* ent, opadpo_saved** saved_out, const int32_t* row_plan, void* stream) {
  if (!c) return (int)hipErrorInvalidValue;
  if (!weights_ready(c)) return cbad(c, __func__, "LLM weights not set");
  if (adapter_id < 0 || adapter_id >= OPADPO_MAX_ADAPTERS) return cbad(c, __func__, "adapter id out of range");
  if (!ids || !text_mask || !feat_row || !feats || !logp || !ent) return cbad(c, __func__, "null operand");
  if (S <= 0 || T <= 0 || K <= 0 || n_txt <= K * T || temperature <= 0.f) return cbad(c, __func__, "bad shape (need n_txt > K*T, temperature > 0)");
  const opadpo_ctx::Adapter& ad = c->adapters[adapter_id];
  if (train && (ad.kind != 1 || !ad.grad || !ad.work_t || !c->lm_head_t)) return cbad(c, __func__, "train=1 needs a LoRA adapter with grad / transposed buffers and lm_head_t");
  if (train && !saved_out) return cbad(c, __func__, "train=1 needs saved_out");
  hipStream_t st = (hipStream_t)stream;
  const opadpo_dims& d = c->d;
  const int side = d.image_size / d.patch, P = side * side, H = d.hidden;
  const int Lp = n_txt + P - 1;
  CK(ensure_rope(c, Lp, st));
  opadpo_saved* sv = new (std::nothrow) opadpo_saved();
  if (!sv) return cfail(c, hipErrorOutOfMemory, __func__);
  sv->S = S; sv->L = Lp; sv->T = T; sv->K = K; sv->M = S * Lp; sv->R = K * S * T; sv->n_txt = n_txt; sv->train = train ? 1 : 0;
  sv->adapter = adapter_id; sv->temperature = temperature; sv->nb = train ? d.n_layers : 1;
  const int pfx = Lp - K * T;
  sv->seg_prefix = K > 1 ? pfx : 0; sv->seg_len = K > 1 ? T : 0;
  sv->ids = ids; sv->feat_row = feat_row;
  // ragged rows: row_plan (HOST, [S][K+1]) = per sequence the number of dropped left-pad positions and the valid length of every
  // response (trailing pad dropped); the valid rows of the batch become the M of every row-wise kernel of the pass
  std::vector<int32_t> metav;
  if (row_plan) {
    const int stride = 2 * K + 4;
    metav.assign((size_t)S * stride, 0);
    int row = 0, lmax = 0;
    for (int q = 0; q < S; ++q) {
      const int32_t* rp = row_plan + (size_t)q * (K + 1);
      int32_t* m = metav.data() + (size_t)q * stride;
      if (rp[0] < 0 || rp[0] >= pfx) { delete sv; return cbad(c, __func__, "row_plan: dropped left padding must leave a non-empty prefix"); }
      m[0] = row; m[1] = pfx - rp[0]; m[2 + K] = rp[0];
      for (int k = 0; k < K; ++k) {
        if (rp[1 + k] < 0 || rp[1 + k] > T) { delete sv; return cbad(c, __func__, "row_plan: response length outside [0, T]"); }
        m[2 + k] = m[1 + k] + rp[1 + k];
      }
      row += m[1 + K];
      lmax = std::max(lmax, (int)m[1 + K]);
    }
    int cells = 0;                                  // compact head rows, in the [k][s] order of the outputs
    for (int k = 0; k < K; ++k)
      for (int q = 0; q < S; ++q) { metav[(size_t)q * stride + 3 + K + k] = cells; cells += row_plan[(size_t)q * (K + 1) + 1 + k]; }
    int nu = 0;                                     // rows of the compact top layer (U), sequence by sequence
    for (int q = 0; q < S; ++q) {
      const int32_t* rp = row_plan + (size_t)q * (K + 1);
      metav[(size_t)q * stride + 3 + 2 * K] = nu;
      int any = 0;
      for (int k = 0; k < K; ++k) { any |= rp[1 + k] > 0; nu += std::max(rp[1 + k] - 1, 0); }
      nu += any;
    }
    sv->ragged = 1; sv->meta_stride = stride; sv->M = row; sv->L = lmax; sv->Rc = cells;
    sv->Uc = (c->use_tr >= 0 && (c->use_tr & 128)) ? 0 : nu;      // context flag bit 7: top layer on every row
  } else {
    sv->Rc = sv->R;
  }
  // chunked head: automatically when the fp32 logits of the batch SHAPE (S * K * T head rows, padding included) reach 4 GiB (K = 3 responses at
  // T = 896, large rollout batches); context flag (use_tr) bit 9 forces it for any size, bit 10 switches it off.  The decision is a function
  // of the shape of the pass, not of how many of its cells are padding (round 3 compared the ragged row count Rc with 2 GiB: batches of one
  // shape whose valid-token counts straddled the threshold alternated between two numerically different heads from step to step, and the
  // reference and policy passes of one step could differ).  Same-box A/B at the benchmark's 22 pairs (2.16 GB in this measure, 1.27 GB of
  // valid rows): chunked 977.2 ms / 217.9 GB, un-chunked 972.9 ms / 220.2 GB per step - below the threshold the one-buffer form is kept (the
  // recomputed lm_head GEMM and 8 short launches cost 0.4 %)
  {
    const int ut = c->use_tr >= 0 ? c->use_tr : 0;
    const bool want = (ut & 512) || (!(ut & 1024) && (size_t)std::max(S * K * T, 1) * d.vocab * sizeof(float) >= ((size_t)4 << 30));
    sv->head_chunk = (want && d.vocab > 4096) ? 4096 : 0;
  }
  sv->bytes = saved_layout(d, sv, nullptr);
  // ragged batches differ in size: ask for the largest arena seen so far for this kind of pass, so that the allocator hands the
  // same block back every time instead of growing (and fragmenting) its pool
  size_t& hint = c->arena_hint[train ? 1 : 0][std::min(K, 15)];
  if (sv->ragged) {                               // size for the padded row count of this shape: a ragged batch can never need more
    opadpo_saved worst = *sv;
    worst.M = S * Lp; worst.L = Lp;
    hint = std::max(hint, saved_layout(d, &worst, nullptr));
  }
  hint = std::max(hint, sv->bytes);
  sv->bytes = hint;
  sv->arena = ctx_alloc(c, sv->bytes, st);
  if (!sv->arena) { delete sv; return cfail(c, hipErrorOutOfMemory, "opadpo_seq_logprobs_fwd (activation arena)"); }
  saved_layout(d, sv, sv->arena);
  c->live.push_back(sv);
  const size_t MH = (size_t)sv->M * H;
#define CKS(expr)                                                                   \
  do {                                                                              \
    hipError_t e_ = (expr);                                                         \
    if (e_ != hipSuccess) { saved_destroy(c, sv); return cfail(c, e_, __func__); }  \
  } while (0)
  Rag rag{sv->meta, sv->meta_stride, K, sv->M, sv->row_pos};
  const Rag* rg = sv->ragged ? &rag : nullptr;
  if (rg) {
    for (size_t o = 0; o < metav.size(); o += sizeof(MetaBlob) / sizeof(int32_t)) {
      MetaBlob blob;
      const int n = (int)std::min(metav.size() - o, sizeof(MetaBlob) / sizeof(int32_t));
      memcpy(blob.v, metav.data() + o, (size_t)n * sizeof(int32_t));
      hipLaunchKernelGGL(write_meta_kernel, dim3(1), dim3(256), 0, st, blob, sv->meta + o, n);
      CKS(hipGetLastError());
    }
    hipLaunchKernelGGL(embed_splice_ragged_kernel, dim3(sv->L, S), dim3(256), 0, st, ids, text_mask, c->embed, feats, feat_row, image_mask, sv->x,
                       sv->key_mask, sv->row_pos, sv->meta, sv->meta_stride, K, T, n_txt, P, H, OPADPO_IMAGE_TOKEN);
    CKS(hipGetLastError());
  } else {
    CKS(launch_embed_splice(ids, text_mask, c->embed, feats, feat_row, image_mask, sv->x, 1, sv->key_mask, S, n_txt, P, H, OPADPO_IMAGE_TOKEN, st));
  }
  const int Rall = sv->R, R = sv->Rc;                // cells of the [K,S,T] outputs; head rows computed
  if (rg) hipLaunchKernelGGL(head_index_ragged_kernel, g1(Rall), dim3(256), 0, st, ids, sv->meta, sv->meta_stride, S, n_txt, K, T, sv->rows, sv->labels, sv->cell,
                             sv->Uc > 0 ? sv->urow : nullptr);
  else hipLaunchKernelGGL(head_index_kernel, g1(Rall), dim3(256), 0, st, ids, S, n_txt, Lp, pfx, K, T, sv->rows, sv->labels);
  CKS(hipGetLastError());
  if (R == 0) {       // every response of the batch is empty: all cells are padding (-0.0 / 0), there is no head row and no gradient
    hipLaunchKernelGGL(fill_f32_kernel, g1(Rall), dim3(256), 0, st, logp, -0.0f, ent, 0.0f, Rall);
    CKS(hipGetLastError());
    if (train) *saved_out = sv; else { saved_destroy(c, sv); if (saved_out) *saved_out = nullptr; }
    return 0;
  }
  float* const Y = sv->x + (size_t)(train ? d.n_layers : 2) * MH;          // branch product of the o / down projections (residual deferred)
  const float* res = sv->x;
  const float* yin = nullptr;
  const TopRows top{sv->urow, sv->Uc, sv->attn_u, sv->hs};                 // hs is free until the head (and |U| <= R rows fit)
  const bool fr = fuse_resid(c);
  for (int i = 0; i < d.n_layers; ++i) {
    const LayerBufs lb = slot(d, sv, train ? i : 0);
    float* x = sv->x + (size_t)(train ? i : (i & 1)) * MH;
    // fused residual: the layer writes its OUTPUT x_{i+1} = h + down(..) into the next layer's slot (training: slot i + 1, the last one being the
    // old branch-product slot; no-grad: the other of the two alternating slots)
    float* const xo = fr ? sv->x + (size_t)(train ? i + 1 : ((i + 1) & 1)) * MH : Y;
    CKS(layer_fwd(c, i, ad, res, yin, x, xo, lb, S, sv->L, sv->key_mask, sv->seg_prefix, sv->seg_len, nullptr, nullptr, 0, st, rg,
                  (sv->Uc > 0 && i == d.n_layers - 1) ? &top : nullptr));
    if (fr) { res = xo; yin = nullptr; } else { res = lb.h; yin = Y; }
  }
  if (fr) {
    // final hidden state of the HEAD rows only: x = the last layer's output rows (fp32 rows = 2H bf16 units)
    CKS(launch_gather_rows((const bf16_t*)res, 2 * H, sv->rows, (bf16_t*)sv->hs, R, 2 * H, st));
    CKS(launch_rmsnorm_fwd(sv->hs, 1, c->norm, sv->hn, sv->rstd_f, R, H, d.rms_eps, st));
  } else {
  // final hidden state x = h + y of the HEAD rows only (fp32 rows = 2H bf16 units; the y rows park in the logits buffer, written later)
  CKS(launch_gather_rows((const bf16_t*)res, 2 * H, sv->rows, (bf16_t*)sv->hs, R, 2 * H, st));
  CKS(launch_gather_rows((const bf16_t*)Y, 2 * H, sv->rows, (bf16_t*)sv->logits, R, 2 * H, st));
  CKS(launch_rmsnorm_sum_fwd(sv->hs, 1, sv->logits, 1, (size_t)R * H, c->norm, sv->hs, sv->hn, sv->rstd_f, R, H, d.rms_eps, st));
  }
  float* const head_logp = rg ? sv->logp_c : logp;
  if (sv->head_chunk > 0) {
    // lm_head + online log-sum-exp + label gather + entropy, one vocabulary chunk at a time (rl_models.py:121-132, common_utils.py:112-118):
    // running max in lse_head, sum z exp(z - m) in ent, sum exp in h_s, label logit in h_zl; the finish kernel turns them into the outputs
    const int VC = sv->head_chunk;
    for (int c0 = 0; c0 < d.vocab; c0 += VC) {
      const int n = std::min(VC, d.vocab - c0);
      GemmNTArgs g = gemm(c, sv->hn, H, c->lm_head + (size_t)c0 * H, H, H, sv->logits, VC, 1, R, n);
      CKS(run_gemm(c, g, st));
      CKS(launch_head_fwd_chunk(sv->logits, VC, sv->labels, 1.0f / temperature, c0, n, c0 == 0, sv->lse_head, sv->h_s, sv->ent, sv->h_zl, R, st));
    }
    CKS(launch_head_fwd_finish(sv->labels, sv->lse_head, sv->h_s, sv->ent, sv->h_zl, head_logp, sv->ent, sv->lse_head, R, st));
  } else {
    GemmNTArgs g = gemm(c, sv->hn, H, c->lm_head, H, H, sv->logits, d.vocab, 1, R, d.vocab);
    CKS(run_gemm(c, g, st));
    CKS(launch_head_fwd(sv->logits, d.vocab, sv->labels, 1.0f / temperature, head_logp, sv->ent, sv->lse_head, R, d.vocab, st));
  }
  if (rg) {       // compact head rows -> the rectangular outputs (padding cells: -0.0 / 0, Quirk Q4)
    hipLaunchKernelGGL(fill_f32_kernel, g1(Rall), dim3(256), 0, st, logp, -0.0f, ent, 0.0f, Rall);
    CKS(hipGetLastError());
    if (R > 0) {
      hipLaunchKernelGGL(scatter_cells_kernel, g1(R), dim3(256), 0, st, sv->logp_c, sv->ent, sv->cell, logp, ent, R);
      CKS(hipGetLastError());
    }
  } else {
    CKS(hipMemcpyAsync(ent, sv->ent, (size_t)R * sizeof(float), hipMemcpyDeviceToDevice, st));
  }
#undef CKS
  if (train) {
    *saved_out = sv;
  } else {
    saved_destroy(c, sv);                           // stream-ordered: the arena is only reused by later work on this stream
    if (saved_out) *saved_out = nullptr;
  }
  return 0;
}

int opadpo_saved_release(opadpo_ctx* c, opadpo_saved* saved) {
  if (!c) return (int)hipErrorInvalidValue;
  if (saved && std::find(c->live.begin(), c->live.end(), saved) == c->live.end()) return cbad(c, __func__, "unknown activation handle");
  saved_destroy(c, saved);
  return 0;
}

int opadpo_saved_residual(opadpo_ctx* c, const opadpo_saved* sv, int layer, float* dst, int* rows, void* stream) {
  if (!c) return (int)hipErrorInvalidValue;
  if (!sv || std::find(c->live.begin(), c->live.end(), sv) == c->live.end()) return cbad(c, __func__, "unknown activation handle");
  if (!sv->train || layer < 0 || layer >= c->d.n_layers) return cbad(c, __func__, "needs a training forward and 0 <= layer < n_layers");
  if (rows) *rows = sv->M;
  if (dst) CK(hipMemcpyAsync(dst, sv->x + (size_t)layer * sv->M * c->d.hidden, (size_t)sv->M * c->d.hidden * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

// ---- LoRA backward (model.py seq_logprobs_bwd; rl_trainer.py:162 accelerator.backward) -----------------------------------------
// Layers [layer_lo, layer_hi] are processed top-down; the first call of a backward must start at layer_hi = n_layers - 1 (it runs
// the head backward), later calls continue where the previous one stopped (the data-parallel host launches the exchange of a
// finished bucket of layers between two calls).
int opadpo_seq_logprobs_bwd(opadpo_ctx* c, opadpo_saved* sv, const float* dlogp, const float* dent, float* d_feats, int layer_hi, int layer_lo,
                            void* stream) {
  if (!c) return (int)hipErrorInvalidValue;
  if (!sv || std::find(c->live.begin(), c->live.end(), sv) == c->live.end()) return cbad(c, __func__, "unknown activation handle");
  if (!sv->train) return cbad(c, __func__, "activations were not saved for backward (train = 0)");
  const opadpo_dims& d = c->d;
  if (layer_hi >= d.n_layers || layer_lo < 0 || layer_lo > layer_hi) return cbad(c, __func__, "bad layer range");
  const opadpo_ctx::Adapter& ad = c->adapters[sv->adapter];
  if (ad.kind != 1 || !ad.grad || !ad.work_t) return cbad(c, __func__, "adapter lost its gradient / transposed buffers");
  hipStream_t st = (hipStream_t)stream;
  const int S = sv->S, Lp = sv->L, M = sv->M, R = sv->Rc, H = d.hidden, F = d.ffn, r = d.lora_r, nh = d.n_heads, hd = d.head_dim, V = d.vocab;
  const float s = d.lora_alpha / d.lora_r;
  const size_t MH = (size_t)M * H;
  // workspace of the LoRA wgrads' deterministic flush (gemm_tn_w4_kernel writes a run's partial tiles there, gemm_tn_reduce_kernel adds them in a
  // fixed order: no fp32 atomics, gradients bit-reproducible run to run): the worst case over the layer's 8 problems as one group, or one
  // by one (compact top layer: two row counts); context flag (use_tr) bit 12 keeps the atomic flush (A/B)
  size_t tn_ws_floats = 1;
  if (!(c->use_tr >= 0 && (c->use_tr & 4096))) {
    GemmTNArgs shp[8];
    const int n1s[8] = {H, r, 2 * F, 2 * r, H, r, 3 * H, 3 * r}, n2s[8] = {r, F, r, H, r, H, r, H}, qg[8] = {0, 0, F, 0, 0, 0, H, 0};
    for (int rows_case = 0; rows_case < 2; ++rows_case) {
      for (int q = 0; q < 8; ++q) {
        memset(&shp[q], 0, sizeof(GemmTNArgs));
        shp[q].M = (rows_case == 1 && sv->Uc > 0 && q < 6) ? sv->Uc : M;
        shp[q].N1 = n1s[q]; shp[q].N2 = n2s[q]; shp[q].ldp = n1s[q]; shp[q].ldq = n2s[q] * (qg[q] ? n1s[q] / qg[q] : 1); shp[q].q_group_n1 = qg[q]; shp[q].use_tr = c->use_tr;
      }
      tn_ws_floats = std::max(tn_ws_floats, gemm_tn_group_workspace_bytes(shp, 8) / sizeof(float));
    }
  }
  // workspace (persists between the ranged calls of one backward)
  size_t need = 0;
  const int VC = sv->head_chunk;                      // chunked head: dz and the recomputed logits exist for one vocabulary chunk at a time
  const size_t dz_cols = VC > 0 ? (size_t)VC : (size_t)V;
  { Carve cv(nullptr); cv.take<bf16_t>((size_t)R * dz_cols); cv.take<float>(VC > 0 ? (size_t)R * H : 1); cv.take<bf16_t>((size_t)R * H); cv.take<float>((size_t)R * H); cv.take<float>(MH); cv.take<bf16_t>(MH);
    cv.take<float>(MH); cv.take<bf16_t>(MH); cv.take<bf16_t>(MH); cv.take<bf16_t>((size_t)M * F); cv.take<bf16_t>((size_t)M * 2 * F); cv.take<bf16_t>(MH);
    cv.take<bf16_t>((size_t)M * 3 * H); cv.take<float>((size_t)S * nh * Lp); cv.take<bf16_t>((size_t)M * r); cv.take<bf16_t>((size_t)M * 2 * r);
    cv.take<bf16_t>((size_t)M * 3 * r); cv.take<bf16_t>((size_t)M * r); cv.take<float>(sv->Uc > 0 ? MH : 1); cv.take<float>(tn_ws_floats); need = cv.off; }
  const bool first = layer_hi == d.n_layers - 1;
  if (first) c->wgrad_det = 1;                        // of THIS backward: cleared below by any grouped wgrad launch that fell back to fp32 atomics
  if (R == 0) return 0;                               // no valid response token: every gradient of this pass is exactly zero (nothing flushed: trivially reproducible)
  if (!first && (c->ws_bytes < need || !c->ws)) return cbad(c, __func__, "ranged backward must start at the top layer");
  void* base = ctx_ws(c, need, st);
  if (!base) return cfail(c, hipErrorOutOfMemory, __func__);
  Carve cv(base);
  bf16_t* dz = cv.take<bf16_t>((size_t)R * dz_cols); float* d_hn32 = cv.take<float>(VC > 0 ? (size_t)R * H : 1);
  bf16_t* d_hn = cv.take<bf16_t>((size_t)R * H); float* d_hs = cv.take<float>((size_t)R * H);
  float* dX = cv.take<float>(MH); bf16_t* dXb = cv.take<bf16_t>(MH); float* d_h = cv.take<float>(MH); bf16_t* d_hb = cv.take<bf16_t>(MH);
  bf16_t* d_n = cv.take<bf16_t>(MH); bf16_t* d_act = cv.take<bf16_t>((size_t)M * F); bf16_t* d_gu = cv.take<bf16_t>((size_t)M * 2 * F);
  bf16_t* d_attn = cv.take<bf16_t>(MH); bf16_t* dqkv = cv.take<bf16_t>((size_t)M * 3 * H); float* delta = cv.take<float>((size_t)S * nh * Lp);
  bf16_t* dt_r = cv.take<bf16_t>((size_t)M * r); bf16_t* dt_2r = cv.take<bf16_t>((size_t)M * 2 * r); bf16_t* dt_3r = cv.take<bf16_t>((size_t)M * 3 * r);
  bf16_t* dt_ra = cv.take<bf16_t>((size_t)M * r);        // dT of the o projection (dt_r keeps the down projection's until the grouped wgrad)
  float* d_full = cv.take<float>(sv->Uc > 0 ? MH : 1);   // compact top layer: its residual gradient scattered back to every row
  float* tn_ws = cv.take<float>(tn_ws_floats);
  const LoraOff o = lora_off(d);
  if (first) {
    if (!dlogp) return cbad(c, __func__, "null dlogp");
    if (sv->ragged && R > 0) {      // gradients of the compact head rows: gathered from the rectangular [K,S,T] gradients (into d_hs, free until the norm backward)
      float* dl_c = d_hs; float* de_c = d_hs + R;
      hipLaunchKernelGGL(gather_cells_kernel, g1(R), dim3(256), 0, st, dlogp, dent, sv->cell, dl_c, de_c, R);
      CK(hipGetLastError());
      dlogp = dl_c; if (dent) dent = de_c;
    }
    if (VC > 0) {
      // per vocabulary chunk: recompute the logits (hn . W_c^T), dz_c from the row statistics of the forward, d_hn += dz_c . W_c (fp32
      // accumulation across the chunks, ONE rounding to bf16 at the end - what the single K = vocab GEMM does inside its accumulators)
      for (int c0 = 0; c0 < V; c0 += VC) {
        const int n = std::min(VC, V - c0);
        { GemmNTArgs g = gemm(c, sv->hn, H, c->lm_head + (size_t)c0 * H, H, H, sv->logits, VC, 1, R, n); CK(run_gemm(c, g, st)); }
        CK(launch_head_bwd(sv->logits, VC, sv->labels, sv->lse_head, dlogp, dent ? sv->ent : nullptr, dent, 1.0f / sv->temperature, dz, VC, R, n, st, c0));
        GemmNTArgs g = gemm(c, dz, VC, c->lm_head_t + c0, V, n, d_hn32, H, 1, R, H);
        if (c0 > 0) resid(g, d_hn32, H, 1);
        CK(run_gemm(c, g, st));
      }
      CK(launch_f32_to_bf16(d_hn32, d_hn, (size_t)R * H, st));
    } else {
      CK(launch_head_bwd(sv->logits, V, sv->labels, sv->lse_head, dlogp, dent ? sv->ent : nullptr, dent, 1.0f / sv->temperature, dz, V, R, V, st));
      GemmNTArgs g = gemm(c, dz, V, c->lm_head_t, V, V, d_hn, H, 0, R, H);
      CK(run_gemm(c, g, st));
    }
    CK(launch_rmsnorm_bwd(d_hn, sv->hs, 1, c->norm, sv->rstd_f, nullptr, 0, d_hs, nullptr, R, H, st));
    CK(hipMemsetAsync(dX, 0, MH * sizeof(float), st));
    if (sv->K > 1 || sv->ragged) CK(launch_scatter_add_rows_f32(d_hs, sv->rows, dX, H, R, H, st));     // rows repeat: the last prefix row feeds token 0 of every response (ragged: compact head rows; with the compact top layer `rows` index the U rows that dX holds for that layer)
    else CK(launch_scatter_rows((const bf16_t*)d_hs, sv->rows, (bf16_t*)dX, 2 * H, R, 2 * H, st));
    CK(launch_f32_to_bf16(dX, dXb, MH, st));
  }
  for (int i = layer_hi; i >= layer_lo; --i) {
    const opadpo_layer_weights& w = c->layers[i];
    if (!w.wqkv_t || !w.wo_t || !w.wgu_t || !w.wd_t) return cbad(c, __func__, "transposed base weights not set (needed by dgrad)");
    const bf16_t* wt = ad.work_t + (size_t)i * o.layer;
    float* gr = ad.grad + (size_t)i * o.layer;
    const LayerBufs b = slot(d, sv, i);
    // the 8 LoRA wgrads of the layer are collected and run as ONE grouped launch (launch_gemm_tn_group) once the last of their
    // operands exists: every operand stays untouched until then (dt_ra is the second dT buffer that makes that true)
    GemmTNArgs wg[8];
    int nwg = 0;
    // compact top layer (opadpo_saved::Uc): its MLP / o-projection activations and the incoming gradient are rows of U, not of the batch
    const bool top = sv->Uc > 0 && i == d.n_layers - 1;
    const int Mm = top ? sv->Uc : M;
    auto tn = [&](const bf16_t* Pm, int ldp, const bf16_t* Q, int ldq, float* Cg, int ldc, int N1, int N2, int qgn, int qgs, int rows) {
      GemmTNArgs& t = wg[nwg++]; t.P = Pm; t.Q = Q; t.C = Cg; t.M = rows; t.N1 = N1; t.N2 = N2; t.ldp = ldp; t.ldq = ldq; t.ldc = ldc;
      t.q_group_n1 = qgn; t.q_group_stride = qgs; t.alpha = 1.f; t.splits = 0; t.use_tr = c->use_tr;
      return hipSuccess;
    };
    // ---- MLP ----
    { GemmNTArgs g = gemm(c, dXb, H, wt + o.b_d, H, H, dt_r, r, 0, Mm, r); g.alpha = s; CK(run_gemm(c, g, st)); }
    CK(tn(dXb, H, b.t_d, r, gr + o.b_d, r, H, r, 0, 0, Mm));
    CK(tn(dt_r, r, b.act, F, gr + o.a_d, F, r, F, 0, 0, Mm));
    const bool sb_staged = c->use_tr >= 0 && (c->use_tr & 64);
    if (F % 256 == 0 && (sb_staged || fuse_swiglu_bwd(c))) {
      // SwiGLU backward in the dgrad's epilogue: d_act never reaches HBM.  Default since round 5 in the DIRECT form (gate / up operands requested one
      // row block ahead of the accumulator read-out, lane-local math, gemm.hip w4_direct_epilogue_swiglu_bwd; context flag bit 14 / OPADPO_FUSE_SWIGLU_BWD=0
      // keep the two-kernel form).  Flag bit 6 = the LDS-staged form of rounds 3-4 (every operand load's latency exposed: 1016.9 vs 1013.3 ms per step
      // for the two-kernel form then) - the cross-check of the direct one.
      GemmNTArgs g = gemm(c, dXb, H, w.wd_t, H, H, d_gu, 2 * F, 0, Mm, F); tail(g, dt_r, r, wt + o.a_d, r, r);
      g.act = OPADPO_ACT_SWIGLU_BWD; g.R = b.gu; g.ldr = 2 * F; g.r_f32 = 0; g.swiglu_bwd_staged = sb_staged ? 1 : 0;
      CK(run_gemm(c, g, st));
    } else {
      { GemmNTArgs g = gemm(c, dXb, H, w.wd_t, H, H, d_act, F, 0, Mm, F); tail(g, dt_r, r, wt + o.a_d, r, r); CK(run_gemm(c, g, st)); }
      CK(launch_silu_mul_bwd(d_act, b.gu, d_gu, Mm, F, st));
    }
    { GemmNTArgs g = gemm(c, d_gu, 2 * F, wt + o.b_gu, F, F, dt_2r, 2 * r, 0, Mm, 2 * r); g.alpha = s; g.a1_group_n = r; g.a1_group_stride = F; CK(run_gemm(c, g, st)); }
    CK(tn(d_gu, 2 * F, b.t_gu, 2 * r, gr + o.b_gu, r, 2 * F, r, F, r, Mm));
    CK(tn(dt_2r, 2 * r, b.n2, H, gr + o.a_gu, H, 2 * r, H, 0, 0, Mm));
    { GemmNTArgs g = gemm(c, d_gu, 2 * F, w.wgu_t, 2 * F, 2 * F, d_n, H, 0, Mm, H); tail(g, dt_2r, 2 * r, wt + o.a_gu, 2 * r, 2 * r); CK(run_gemm(c, g, st)); }
    CK(launch_rmsnorm_bwd(d_n, b.h, 1, w.ln2, b.rstd2, dX, 1, d_h, d_hb, Mm, H, st));
    // ---- attention ----
    { GemmNTArgs g = gemm(c, d_hb, H, wt + o.b_o, H, H, dt_ra, r, 0, Mm, r); g.alpha = s; CK(run_gemm(c, g, st)); }
    CK(tn(d_hb, H, b.t_o, r, gr + o.b_o, r, H, r, 0, 0, Mm));
    CK(tn(dt_ra, r, top ? sv->attn_u : b.attn, H, gr + o.a_o, H, r, H, 0, 0, Mm));
    {
      bf16_t* dst = top ? d_n : d_attn;            // d_n is free again: the norm backward above has consumed it
      GemmNTArgs g = gemm(c, d_hb, H, w.wo_t, H, H, dst, H, 0, Mm, H); tail(g, dt_ra, r, wt + o.a_o, r, r); CK(run_gemm(c, g, st));
      if (top) {                                   // gradient of the attention output: U rows carry one, every other row zero
        CK(hipMemsetAsync(d_attn, 0, MH * sizeof(bf16_t), st));
        CK(launch_scatter_rows(d_n, sv->urow, d_attn, H, Mm, H, st));
      }
    }
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = b.qkv; a.k = b.qkv + H; a.v = b.qkv + 2 * H; a.o = b.attn; a.lse = b.lse; a.key_mask = sv->key_mask;
    a.S = S; a.L = Lp; a.nh = nh; a.hd = hd; a.ld = 3 * H; a.ldo = H; a.causal = 1 | OPADPO_ATTN_SKIP_MASKED_Q; a.scale = 1.0f / sqrtf((float)hd);
    a.dout = d_attn; a.dq_acc = nullptr; a.dq = dqkv; a.dk = dqkv + H; a.dv = dqkv + 2 * H; a.delta = delta;
    a.seg_prefix = sv->seg_prefix; a.seg_len = sv->seg_len; a.use_tr = c->use_tr;
    if (sv->ragged) { a.seq_meta = sv->meta; a.meta_stride = sv->meta_stride; a.n_seg = sv->K; a.rows_total = M; a.seg_prefix = 0; a.seg_len = 0; }
    // ragged rows: dQ / dK leave the attention backward already rotated back (lane-local in its accumulator layout) - no pass over dq | dk;
    // context flag bit 11 (and padded rows) keep the separate inverse rope kernel
    const bool fuse_rope_bwd = sv->ragged && !(c->use_tr >= 0 && (c->use_tr & 2048));
    if (fuse_rope_bwd) { a.rope_pos = sv->row_pos; a.rope_l2theta = log2f(d.rope_theta); }
    CK(launch_attn_bwd(a, st));
    if (!fuse_rope_bwd) CK(launch_rope(dqkv, 3 * H, c->cosb, c->sinb, M, Lp, 2 * nh, hd, 1, nullptr, sv->seg_prefix, sv->seg_len, st, sv->ragged ? sv->row_pos : nullptr,
                                       sv->ragged ? log2f(d.rope_theta) : 0.f));
    { GemmNTArgs g = gemm(c, dqkv, 3 * H, wt + o.b_qkv, H, H, dt_3r, 3 * r, 0, M, 3 * r); g.alpha = s; g.a1_group_n = r; g.a1_group_stride = H; CK(run_gemm(c, g, st)); }
    CK(tn(dqkv, 3 * H, b.t_qkv, 3 * r, gr + o.b_qkv, r, 3 * H, r, H, r, M));
    CK(tn(dt_3r, 3 * r, b.n1, H, gr + o.a_qkv, H, 3 * r, H, 0, 0, M));
    { int ord = 1;      // what the launches of THIS list actually did (not a synthetic shape list): ordered reduce for every problem, or atomics somewhere
      CK(launch_gemm_tn_group(wg, nwg, st, tn_ws_floats > 1 ? tn_ws : nullptr, tn_ws_floats > 1 ? tn_ws_floats * sizeof(float) : 0, &ord));
      if (!ord) c->wgrad_det = 0; }
    if (i > 0 || d_feats) {          // layer-0 input is the frozen embedding / image features: no further dgrad in the DPO stage
      { GemmNTArgs g = gemm(c, dqkv, 3 * H, w.wqkv_t, 3 * H, 3 * H, d_n, H, 0, M, H); tail(g, dt_3r, 3 * r, wt + o.a_qkv, 3 * r, 3 * r); CK(run_gemm(c, g, st)); }
      const float* dres = d_h;
      if (top) {                                   // residual gradient of the block: d_h holds it for the U rows only
        CK(hipMemsetAsync(d_full, 0, MH * sizeof(float), st));
        CK(launch_scatter_rows((const bf16_t*)d_h, sv->urow, (bf16_t*)d_full, 2 * H, Mm, 2 * H, st));      // fp32 rows = 2H bf16 units
        dres = d_full;
      }
      CK(launch_rmsnorm_bwd(d_n, sv->x + (size_t)i * MH, 1, w.ln1, b.rstd1, dres, 1, dX, dXb, M, H, st));
    }
  }
  if (layer_lo == 0 && d_feats) {
    const int side = d.image_size / d.patch, P = side * side;
    hipLaunchKernelGGL(splice_grad_kernel, dim3(P, S), dim3(256), 0, st, dX, sv->ids, sv->feat_row, d_feats, S, sv->n_txt, Lp, P, H, OPADPO_IMAGE_TOKEN,
                       sv->ragged ? sv->meta : nullptr, sv->meta_stride, sv->K);
    CK(hipGetLastError());
  }
  return 0;
}

// ---- rollout: prefill + KV-cache decode (generate.py; online_generator.py:292-309) ----------------------------------------------
static hipError_t decode_head(opadpo_ctx* c, const float* src_f32, const float* partials, int n_partials, hipStream_t st) {
  opadpo_ctx::Decode& D = c->dec;
  const opadpo_dims& d = c->d;
  hipError_t e;
  if (n_partials > 0) {
    if ((e = launch_rmsnorm_sum_fwd(src_f32, 1, partials, n_partials, (size_t)D.B * d.hidden, c->norm, D.x2, D.hn, D.rstd, D.B, d.hidden, d.rms_eps, st)) != hipSuccess) return e;
  } else if ((e = launch_rmsnorm_fwd(src_f32, 1, c->norm, D.hn, D.rstd, D.B, d.hidden, d.rms_eps, st)) != hipSuccess) return e;
  GemmNTArgs g = gemm(c, D.hn, d.hidden, c->lm_head, d.hidden, d.hidden, D.logits, d.vocab, 1, D.B, d.vocab); g.act |= OPADPO_GEMM_STREAM;
  if (D.use64) e = launch_gemm_nt_dec64(g, 1, 1, st);
  else e = run_gemm(c, g, st);
  if (e != hipSuccess) return e;
  if (D.suppress_eos) {
    hipLaunchKernelGGL(set_column_f32_kernel, g1(D.B), dim3(256), 0, st, D.logits, d.vocab, D.B, D.eos_id, -INFINITY);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  if ((e = launch_sample(D.logits, d.vocab, D.B, d.vocab, D.temperature, D.top_k, D.top_p, D.seed, 0, D.step_d, D.finished, D.pad_id,
                         D.suppress_eos ? -1 : D.eos_id, D.cur_tok, D.history, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(64), 0, st, D.step_d, 1);
  return hipGetLastError();
}

static hipError_t decode_one(opadpo_ctx* c, hipStream_t st) {
  opadpo_ctx::Decode& D = c->dec;
  const opadpo_dims& d = c->d;
  const int B = D.B, H = d.hidden, r = d.lora_r, nh = d.n_heads, hd = d.head_dim;
  const float s = d.lora_alpha / d.lora_r;
  const opadpo_ctx::Adapter& ad = c->adapters[D.adapter];
  const LoraOff o = lora_off(d);
  hipError_t e;
  hipLaunchKernelGGL(add_i32_kernel, dim3(1), dim3(64), 0, st, D.pos_d, 1);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if ((e = launch_gather_rows(c->embed, H, D.cur_tok, D.emb, B, H, st)) != hipSuccess) return e;
  if (D.use64) {
    // 9..64 tokens: the decode GEMMs for <= 64 tokens (opadpo_gemm_nt_decode); x / hb = fp32 residual stream, part_o / part_d = K-slice partial tiles of the o / down
    // projections, added (in slice order) by the RMSNorm that follows them
    const int F = d.ffn;
    const size_t pstride = (size_t)B * H;
    auto dg = [&](const bf16_t* A, int lda, const bf16_t* W, int K, void* C, int ldc, int N, int mode, int splits) {
      GemmNTArgs g = gemm(c, A, lda, W, K, K, C, ldc, mode == 1, B, N);
      return launch_gemm_nt_dec64(g, mode, splits, st);
    };
    for (int i = 0; i < d.n_layers; ++i) {
      const opadpo_layer_weights& w0 = c->layers[i];
      const opadpo_layer_weights& w = ad.kind == 2 ? ad.merged[i] : w0;
      if (i == 0) e = launch_rmsnorm_sum_fwd(D.emb, 0, nullptr, 0, 0, w0.ln1, D.x, D.n1, D.rstd, B, H, d.rms_eps, st);
      else e = launch_rmsnorm_sum_fwd(D.hb, 1, D.part_d, D.split_d, pstride, w0.ln1, D.x, D.n1, D.rstd, B, H, d.rms_eps, st);
      if (e != hipSuccess) return e;
      if ((e = dg(D.n1, H, w.wqkv, H, D.qkv, 3 * H, 3 * H, 0, 1)) != hipSuccess) return e;
      const size_t per_layer = (size_t)B * nh * D.max_ctx * hd;
      if ((e = launch_attn_decode_fused(D.qkv, 3 * H, c->cosb, c->sinb, D.kc + i * per_layer, D.vc + i * per_layer, D.att, D.key_mask, B, nh, hd, D.pos_d,
                                        D.max_ctx, 1.0f / sqrtf((float)hd), D.ws, D.ws_bytes, st)) != hipSuccess) return e;
      if ((e = dg(D.att, H, w.wo, H, D.part_o, H, H, 1, D.split_o)) != hipSuccess) return e;
      if ((e = launch_rmsnorm_sum_fwd(D.x, 1, D.part_o, D.split_o, pstride, w0.ln2, D.hb, D.n2, D.rstd, B, H, d.rms_eps, st)) != hipSuccess) return e;
      if (ad.kind == 2 && ad.swiglu_pair) {
        if ((e = dg(D.n2, H, w.wgu, H, D.act, F, 2 * F, 2, 1)) != hipSuccess) return e;
      } else {
        if ((e = dg(D.n2, H, w.wgu, H, D.gu, 2 * F, 2 * F, 0, 1)) != hipSuccess) return e;
        if ((e = launch_silu_mul_fwd(D.gu, D.act, B, F, st)) != hipSuccess) return e;
      }
      if ((e = dg(D.act, F, w.wd, F, D.part_d, H, H, 1, D.split_d)) != hipSuccess) return e;
    }
    return decode_head(c, D.hb, D.part_d, D.split_d, st);
  }
  const void* cur = D.emb; int cur_f32 = 0;
  float* nx = D.x;
  LayerBufs b; b.n1 = D.n1; b.qkv = D.qkv; b.t_qkv = D.t_qkv; b.attn = D.att; b.t_o = D.t_o; b.n2 = D.n2; b.t_gu = D.t_gu; b.gu = D.gu; b.act = D.act;
  b.t_d = D.t_d; b.rstd1 = D.rstd; b.rstd2 = D.rstd; b.lse = nullptr; b.h = D.hb;
  for (int i = 0; i < d.n_layers; ++i) {
    const opadpo_layer_weights& w0 = c->layers[i];
    const opadpo_layer_weights& w = ad.kind == 2 ? ad.merged[i] : w0;
    const bf16_t* lw = ad.kind == 1 ? ad.work + (size_t)i * o.layer : nullptr;
    if ((e = launch_rmsnorm_fwd(cur, cur_f32, w0.ln1, D.n1, D.rstd, B, H, d.rms_eps, st)) != hipSuccess) return e;
    if (lw) {
      GemmNTArgs g1_ = gemm(c, D.n1, H, lw + o.a_qkv, H, H, D.t_qkv, 3 * r, 0, B, 3 * r); g1_.alpha = s; g1_.act |= OPADPO_GEMM_STREAM;
      if ((e = run_gemm(c, g1_, st)) != hipSuccess) return e;
      GemmNTArgs g2 = gemm(c, D.n1, H, w.wqkv, H, H, D.qkv, 3 * H, 0, B, 3 * H); tail(g2, D.t_qkv, 3 * r, lw + o.b_qkv, r, r, H, r); g2.act |= OPADPO_GEMM_STREAM;
      if ((e = run_gemm(c, g2, st)) != hipSuccess) return e;
    } else {
      GemmNTArgs g2 = gemm(c, D.n1, H, w.wqkv, H, H, D.qkv, 3 * H, 0, B, 3 * H); g2.act |= OPADPO_GEMM_STREAM;
      if ((e = run_gemm(c, g2, st)) != hipSuccess) return e;
    }
    const size_t per_layer = (size_t)B * nh * D.max_ctx * hd;
    if ((e = launch_attn_decode_fused(D.qkv, 3 * H, c->cosb, c->sinb, D.kc + i * per_layer, D.vc + i * per_layer, D.att, D.key_mask, B, nh, hd, D.pos_d,
                                      D.max_ctx, 1.0f / sqrtf((float)hd), D.ws, D.ws_bytes, st)) != hipSuccess) return e;
    if (lw) {
      GemmNTArgs g3 = gemm(c, D.att, H, lw + o.a_o, H, H, D.t_o, r, 0, B, r); g3.alpha = s; g3.act |= OPADPO_GEMM_STREAM;
      if ((e = run_gemm(c, g3, st)) != hipSuccess) return e;
      GemmNTArgs g4 = gemm(c, D.att, H, w.wo, H, H, D.hb, H, 1, B, H); tail(g4, D.t_o, r, lw + o.b_o, r, r); resid(g4, cur, H, cur_f32); g4.act |= OPADPO_GEMM_STREAM;
      if ((e = run_gemm(c, g4, st)) != hipSuccess) return e;
    } else {
      GemmNTArgs g4 = gemm(c, D.att, H, w.wo, H, H, D.hb, H, 1, B, H); resid(g4, cur, H, cur_f32); g4.act |= OPADPO_GEMM_STREAM;
      if ((e = run_gemm(c, g4, st)) != hipSuccess) return e;
    }
    if ((e = mlp_fwd(c, i, ad, D.hb, nx, b, B, OPADPO_GEMM_STREAM, st)) != hipSuccess) return e;
    cur = nx; cur_f32 = 1;
    nx = (nx == D.x) ? D.x2 : D.x;
  }
  return decode_head(c, (const float*)cur, nullptr, 0, st);
}

int opadpo_decode_begin(opadpo_ctx* c, int adapter_id, const int32_t* ids, const uint8_t* text_mask, const uint16_t* feats, int B, int Q,
                        int max_new_tokens, float temperature, int top_k, float top_p, uint64_t seed, int eos_id, int pad_id, int suppress_eos,
                        int32_t* history, void* stream) {
  if (!c) return (int)hipErrorInvalidValue;
  if (!weights_ready(c)) return cbad(c, __func__, "LLM weights not set");
  if (adapter_id < 0 || adapter_id >= OPADPO_MAX_ADAPTERS) return cbad(c, __func__, "adapter id out of range");
  if (!ids || !text_mask || !feats || !history || B <= 0 || Q <= 0 || max_new_tokens <= 0 || temperature <= 0.f) return cbad(c, __func__, "null operand or bad shape");
  if (B > 64) return cbad(c, __func__, "decode batch above 64 per device is not supported by the weight-streaming GEMMs");
  opadpo_decode_end(c);
  hipStream_t st = (hipStream_t)stream;
  const opadpo_dims& d = c->d;
  const opadpo_ctx::Adapter& ad = c->adapters[adapter_id];
  const int side = d.image_size / d.patch, P = side * side, H = d.hidden, F = d.ffn, r = d.lora_r, nh = d.n_heads, hd = d.head_dim, V = d.vocab;
  const int Lp = Q + P - 1, max_ctx = Lp + max_new_tokens, M = B * Lp;
  CK(ensure_rope(c, max_ctx, st));
  opadpo_ctx::Decode& D = c->dec;
  D.B = B; D.Lp = Lp; D.max_ctx = max_ctx; D.adapter = adapter_id; D.max_new = max_new_tokens;
  D.history = history; D.temperature = temperature; D.top_k = top_k; D.top_p = top_p; D.seed = seed; D.eos_id = eos_id; D.pad_id = pad_id;
  D.suppress_eos = suppress_eos;
  D.ws_bytes = attn_decode_workspace_bytes(B, nh, hd, max_ctx);
  // 9..64 sequences without LoRA tails (adapter-free or merged adapter): the decode GEMM for <= 64 tokens (gemm_nt_dec64x_kernel), the o / down
  // projections K-split into fp32 partial tiles that the following RMSNorm adds.  Measured per decode step against the 8/16-row streaming kernels
  // of the other path: B = 16 4.84 vs 5.15 ms, B = 8 4.12 vs 3.80 ms.  OPADPO_DEC64_MIN (diagnostics): smallest batch taking it.
  static const int dec64_min = getenv("OPADPO_DEC64_MIN") ? atoi(getenv("OPADPO_DEC64_MIN")) : 9;
  D.use64 = B >= dec64_min && B <= 64 && ad.kind != 1 && H <= 256 * 8 * 3 && F % 64 == 0 && !(c->use_tr >= 0 && (c->use_tr & 32));
  D.split_o = D.use64 ? gemm_nt_dec64_splits(H, H, 0) : 1;
  D.split_d = D.use64 ? gemm_nt_dec64_splits(H, F, 0) : 1;
  auto layout = [&](void* base) {
    Carve cv(base);
    const size_t kvn = (size_t)d.n_layers * B * nh * max_ctx * hd;
    D.kc = cv.take<bf16_t>(kvn); D.vc = cv.take<bf16_t>(kvn); D.key_mask = cv.take<uint8_t>((size_t)B * max_ctx);
    D.x = cv.take<float>((size_t)B * H); D.x2 = cv.take<float>((size_t)B * H); D.hs = cv.take<float>((size_t)B * H); D.hn = cv.take<bf16_t>((size_t)B * H);
    D.n1 = cv.take<bf16_t>((size_t)B * H); D.qkv = cv.take<bf16_t>((size_t)B * 3 * H); D.t_qkv = cv.take<bf16_t>((size_t)B * 3 * r);
    D.att = cv.take<bf16_t>((size_t)B * H); D.t_o = cv.take<bf16_t>((size_t)B * r); D.hb = cv.take<float>((size_t)B * H); D.n2 = cv.take<bf16_t>((size_t)B * H);
    D.t_gu = cv.take<bf16_t>((size_t)B * 2 * r); D.gu = cv.take<bf16_t>((size_t)B * 2 * F); D.act = cv.take<bf16_t>((size_t)B * F); D.t_d = cv.take<bf16_t>((size_t)B * r);
    D.rstd = cv.take<float>(B); D.emb = cv.take<bf16_t>((size_t)B * H); D.logits = cv.take<float>((size_t)B * V);
    D.cur_tok = cv.take<int32_t>(B); D.finished = cv.take<uint8_t>(B); D.step_d = cv.take<int32_t>(1); D.pos_d = cv.take<int32_t>(1);
    D.ws = cv.take<uint8_t>(std::max<size_t>(D.ws_bytes, 4));
    D.part_o = cv.take<float>(D.use64 ? (size_t)D.split_o * B * H : 1);
    D.part_d = cv.take<float>(D.use64 ? (size_t)D.split_d * B * H : 1);
    return cv.off;
  };
  D.bytes = layout(nullptr);
  D.arena = ctx_alloc(c, D.bytes, st);
  if (!D.arena) return cfail(c, hipErrorOutOfMemory, "opadpo_decode_begin (KV cache)");
  layout(D.arena);
  D.active = true;
  // ---- prefill through the full-sequence kernels (scratch arena released afterwards) ----
  opadpo_saved pf;
  pf.S = B; pf.L = Lp; pf.T = 1; pf.K = 1; pf.M = M; pf.R = B; pf.train = 0; pf.nb = 1;
  size_t pbytes = 0;
  auto playout = [&](void* base) {
    Carve cv(base);
    pf.x = cv.take<float>(3 * (size_t)M * H); pf.n1 = cv.take<bf16_t>((size_t)M * H); pf.rstd1 = cv.take<float>(M); pf.qkv = cv.take<bf16_t>((size_t)M * 3 * H);
    pf.t_qkv = cv.take<bf16_t>((size_t)M * 3 * r); pf.attn = cv.take<bf16_t>((size_t)M * H); pf.lse = cv.take<float>((size_t)B * nh * Lp);
    pf.t_o = cv.take<bf16_t>((size_t)M * r); pf.h = cv.take<float>((size_t)M * H); pf.n2 = cv.take<bf16_t>((size_t)M * H); pf.rstd2 = cv.take<float>(M);
    pf.t_gu = cv.take<bf16_t>((size_t)M * 2 * r); pf.gu = cv.take<bf16_t>((size_t)M * 2 * F); pf.act = cv.take<bf16_t>((size_t)M * F); pf.t_d = cv.take<bf16_t>((size_t)M * r);
    pf.key_mask = cv.take<uint8_t>((size_t)B * Lp); pf.rows = cv.take<int32_t>(B);
    pf.labels = cv.take<int32_t>(B);              // reused as feat_row = arange(B): every prompt carries its own image
    return cv.off;
  };
  pbytes = playout(nullptr);
  void* parena = ctx_alloc(c, pbytes, st);
  if (!parena) { opadpo_decode_end(c); return cfail(c, hipErrorOutOfMemory, "opadpo_decode_begin (prefill scratch)"); }
  playout(parena);
#define CKD(expr)                                                                                         \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) { ctx_free(c, parena, pbytes); opadpo_decode_end(c); return cfail(c, e_, __func__); } \
  } while (0)
  hipLaunchKernelGGL(affine_index_kernel, g1(B), dim3(256), 0, st, pf.labels, B, 1, 0);       // feat_row = arange(B)
  CKD(hipGetLastError());
  CKD(launch_embed_splice(ids, text_mask, c->embed, feats, pf.labels, nullptr, pf.x, 1, pf.key_mask, B, Q, P, H, OPADPO_IMAGE_TOKEN, st));
  hipLaunchKernelGGL(decode_mask_kernel, g1((size_t)B * max_ctx), dim3(256), 0, st, pf.key_mask, D.key_mask, B, Lp, max_ctx);
  CKD(hipGetLastError());
  const size_t MH = (size_t)M * H, per_layer = (size_t)B * nh * max_ctx * hd;
  const LayerBufs pb = slot(d, &pf, 0);
  float* const Yp = pf.x + 2 * MH;
  const float* yin = nullptr;
  const bool fr = fuse_resid(c);                    // residual adds in the projections' epilogues: a layer writes its output into the other x slot
  const float* xin = pf.x;
  for (int i = 0; i < d.n_layers; ++i) {
    float* const xo = fr ? pf.x + (size_t)((i + 1) & 1) * MH : Yp;
    CKD(layer_fwd(c, i, ad, fr ? xin : (i == 0 ? pf.x : pb.h), yin, pf.x + (size_t)(i & 1) * MH, xo, pb, B, Lp, pf.key_mask, 0, 0, D.kc + i * per_layer,
                  D.vc + i * per_layer, max_ctx, st));
    if (fr) xin = xo; else yin = Yp;
  }
  hipLaunchKernelGGL(affine_index_kernel, g1(B), dim3(256), 0, st, pf.rows, B, Lp, Lp - 1);
  CKD(hipGetLastError());
  if (fr) CKD(launch_gather_rows((const bf16_t*)xin, 2 * H, pf.rows, (bf16_t*)D.hs, B, 2 * H, st));   // last position of the last layer's output
  else {
  CKD(launch_gather_rows((const bf16_t*)pb.h, 2 * H, pf.rows, (bf16_t*)D.hs, B, 2 * H, st));          // last position: x = h + y, added by the head's norm
  CKD(launch_gather_rows((const bf16_t*)Yp, 2 * H, pf.rows, (bf16_t*)D.x, B, 2 * H, st));
  }
  CKD(hipMemsetAsync(D.finished, 0, B, st));
  CKD(hipMemsetAsync(D.step_d, 0, sizeof(int32_t), st));
  { const int32_t p0 = Lp - 1; CKD(hipMemcpyAsync(D.pos_d, &p0, sizeof(int32_t), hipMemcpyHostToDevice, st)); CKD(hipStreamSynchronize(st)); }
  // history [max_new, B] starts as pad
  {
    std::vector<int32_t> pad((size_t)max_new_tokens * B, pad_id);
    CKD(hipMemcpyAsync(history, pad.data(), pad.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    CKD(hipStreamSynchronize(st));
  }
  CKD(fr ? decode_head(c, D.hs, nullptr, 0, st) : decode_head(c, D.hs, D.x, 1, st));          // token 0 from the prefill logits
#undef CKD
  ctx_free(c, parena, pbytes);
  return 0;
}

int opadpo_decode_step(opadpo_ctx* c, void* stream) {
  if (!c) return (int)hipErrorInvalidValue;
  if (!c->dec.active) return cbad(c, __func__, "no active rollout (call opadpo_decode_begin)");
  CK(decode_one(c, (hipStream_t)stream));
  return 0;
}

// n further tokens.  use_graph = 0 (what the host side asks for by default): the ~230 launches of a step are issued one by one from this
// loop - nothing between them touches the host, the queue stays full, and on MI355X this is the FASTER form (7B: B = 4 3.45 vs 3.72 ms,
// B = 8 3.88 vs 4.09, B = 64 9.26 vs 9.52 ms per step: a graph node pays ~1 us more per kernel boundary than a queued launch).
// use_graph = 1: the launches of ONE step are captured once into a hipGraph (on an internal stream: the caller's may be the legacy
// default stream, which cannot be captured) and replayed per token - for hosts whose launch path cannot keep up.  Either way
// everything that changes between steps lives in device memory (position / step counters, current tokens, finished flags).
int opadpo_decode_run(opadpo_ctx* c, int n_steps, int use_graph, void* stream) {
  if (!c) return (int)hipErrorInvalidValue;
  opadpo_ctx::Decode& D = c->dec;
  if (!D.active) return cbad(c, __func__, "no active rollout (call opadpo_decode_begin)");
  if (n_steps <= 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (!use_graph || n_steps < 3) {
    for (int i = 0; i < n_steps; ++i) CK(decode_one(c, st));
    return 0;
  }
  int done = 0;
  if (!D.exec) {
    CK(decode_one(c, st));                          // warm-up of every kernel of the step (lazy attribute calls are not capturable)
    done = 1;
    if (!D.cap_stream) CK(hipStreamCreateWithFlags(&D.cap_stream, hipStreamNonBlocking));
    if (!D.ev_in) CK(hipEventCreateWithFlags(&D.ev_in, hipEventDisableTiming));
    if (!D.ev_out) CK(hipEventCreateWithFlags(&D.ev_out, hipEventDisableTiming));
    CK(hipStreamSynchronize(st));
    CK(hipStreamBeginCapture(D.cap_stream, hipStreamCaptureModeThreadLocal));
    hipError_t e = decode_one(c, D.cap_stream);
    hipError_t e2 = hipStreamEndCapture(D.cap_stream, &D.graph);
    if (e != hipSuccess) return cfail(c, e, "opadpo_decode_run (capture)");
    if (e2 != hipSuccess) return cfail(c, e2, "opadpo_decode_run (end capture)");
    CK(hipGraphInstantiate(&D.exec, D.graph, nullptr, nullptr, 0));
  }
  CK(hipEventRecord(D.ev_in, st));
  CK(hipStreamWaitEvent(D.cap_stream, D.ev_in, 0));
  for (; done < n_steps; ++done) CK(hipGraphLaunch(D.exec, D.cap_stream));
  CK(hipEventRecord(D.ev_out, D.cap_stream));
  CK(hipStreamWaitEvent(st, D.ev_out, 0));
  return 0;
}

int opadpo_decode_all_finished(opadpo_ctx* c, int* all_finished, void* stream) {
  if (!c) return (int)hipErrorInvalidValue;
  opadpo_ctx::Decode& D = c->dec;
  if (!D.active || !all_finished) return cbad(c, __func__, "no active rollout");
  std::vector<uint8_t> f(D.B);
  CK(hipMemcpyAsync(f.data(), D.finished, D.B, hipMemcpyDeviceToHost, (hipStream_t)stream));
  CK(hipStreamSynchronize((hipStream_t)stream));
  int all = 1;
  for (uint8_t v : f) all &= (v != 0);
  *all_finished = all;
  return 0;
}

int opadpo_decode_end(opadpo_ctx* c) {
  if (!c) return (int)hipErrorInvalidValue;
  opadpo_ctx::Decode& D = c->dec;
  if (D.exec) { (void)hipGraphExecDestroy(D.exec); D.exec = nullptr; }
  if (D.graph) { (void)hipGraphDestroy(D.graph); D.graph = nullptr; }
  if (D.cap_stream) { (void)hipStreamSynchronize(D.cap_stream); }
  if (D.active) { ctx_free(c, D.arena, D.bytes); D.arena = nullptr; D.active = false; }
  return 0;
}

}  // extern "C"
