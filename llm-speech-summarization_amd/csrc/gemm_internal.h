// gemm_internal.h — declarations shared by the GEMM translation units (gemm.hip, gemm_stream.hip).
#pragma once
#include "common.h"

struct GemmP {
  const void* A; int64_t lda, sA;
  const void* W; int64_t ldw, sW;
  void* C; int64_t ldc, sC;
  const void* bias; int64_t sBias;
  const void* res; int64_t ldr, sR;
  int M, N, K, out_f32;
  int tiles_m, tiles_n;
  int ta, tw;          // operand stored transposed: A as (K, M) rows lda; W as (K, N) rows ldw
  void* aux;           // optional: pre-activation values (after bias, before act), same layout/dtype as C
  int res_f32;         // residual is float (fp32 gradient accumulation: C = C_old + A.W^T with out_f32)
  const int64_t* grp;  // grouped (ragged) batch: per z {M, a_off, c_off, r_off} in elements; W/bias use z % w_mod
  int w_mod;
  int64_t cx, rx, wx;  // per-block extra offsets resolved from grp
  int grp_ext;         // records are {M, a_off, c_off, r_off, w_off, N, K, 0}
  int grp_kslab;       // groups_ext == 2: every group's K is a whole number of 128-byte slabs (LDS-DMA kernel allowed)
  int gm;              // 256-tile kernel: tile rows per XCD patch (SL_GEMM_GM, default 8)
  int direct_epi;      // tiled kernels: skip the LDS-staged row epilogue (SL_DIRECT_EPILOGUE=1, for A/B measurements)
  // LayerNorm folded into the surrounding Linears (sl_gemm_ex_args.ln_* / stats_out; rows epilogue of the LDS-DMA tiled kernels, bf16):
  const float* ln_mr;  // consumer: per row {mean, rstd} of the LayerNorm in front of this Linear; W carries the gain, ln_u[n] = sum_k W[n][k],
  const float* ln_u;   //           ln_c[n] = (W0 . beta)[n] + bias[n]:  out = rstd * (A . W^T - mean * ln_u) + ln_c
  const float* ln_c;
  float* stats_out;    // producer: per row and 64-column segment {sum, sum of squares} of the STORED (rounded) values, [row][N / 64][2]
  float* amax_val;     // fused row-wise top-1 (sl_gemm_ex_args.amax_*): per 64-column group g and row m the largest value of
  int* amax_idx;       // columns [64 g, 64 g + 64) at [g][m] and its column index; C is then not written at all
  // training-tape epilogue fusions (sl_gemm_ex_args.post_op ...): rows epilogue (generic form) / direct epilogue / split-K reduce pass
  int post;            // SL_POST_*
  uint32_t drop_thr24; // keep iff u24(hash) >= thr (0: no mask)
  float drop_scale;    // 1 / (1 - p)
  uint64_t drop_seed;
  int64_t drop_ld;     // element index of (row, col) in the mask = row * drop_ld + col
  const void* post_in; int64_t post_ld;
  float* colsum;       // fp32 column sums of the stored values (+=, atomics)
  int32_t* defer;      // host: split-K reduce pass left to the consumer (sl_gemm_ex_args.deferred_splits)
  int krun;            // phased 256-tile kernel, K runs of UNEVEN length on blockIdx.y: run z takes slabs [z krun, min((z + 1) krun, K / BK)) of the same A / W (sA = sW = 0) and
                       // writes its fp32 partial tile to C + z sC (gemm.hip splitk256_runs); 0 = blockIdx.y is a batch index
  uint32_t* stamp;     // instrumented build of the phased 256-tile kernel only: [block][half][32] cycle stamps (SL_GEMM_STAMP_PTR)
};

// resolve the per-batch descriptor: returns false when this block's tile lies outside batch z's rows
__device__ __forceinline__ bool resolve_group(GemmP& p, int z, int bm, int64_t& a_off, int& wz, int tile_rows = 128) {
  p.cx = 0; p.rx = 0; p.wx = 0;
  wz = z;
  a_off = (int64_t)z * p.sA;
  if (p.grp) {
    const int64_t* g = p.grp + (p.grp_ext ? 8 : 4) * (int64_t)z;
    p.M = (int)g[0];
    if (p.grp_ext) { p.wx = g[4]; p.N = (int)g[5]; p.K = (int)g[6]; }
    a_off = g[1];
    p.cx = g[2] - (int64_t)z * p.sC;   // the epilogue adds z*sC back
    p.rx = g[3] - (int64_t)z * p.sR;
    wz = z % p.w_mod;
    if (bm * tile_rows >= p.M) return false;
  }
  return true;
}

// ----------------------------------------------------------------------------------------------
// epilogue helper: +bias, act, +residual, store (T or float)
// ----------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void store_out(const GemmP& p, void* Cb, const void* Rb, int64_t row, int64_t col, float v) {
  if (Rb) v += p.res_f32 ? ((const float*)Rb)[row * p.ldr + col] : to_f32(((const T*)Rb)[row * p.ldr + col]);
  if (p.out_f32)
    ((float*)Cb)[row * p.ldc + col] = v;
  else
    ((T*)Cb)[row * p.ldc + col] = from_f32<T>(v);
}

// ----------------------------------------------------------------------------------------------
// training-tape post-ops on NV consecutive columns of one output row (v = A.W^T + bias, after act; before the residual add).
// Roundings follow the unfused launch sequence: GEMM store -> sl_dropout (in place) -> sl_gelu_bwd, each storing in T.
// ----------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float round_as(float v) { return to_f32(from_f32<T>(v)); }
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }

template <typename T, int NV>
__device__ __forceinline__ void post_drop(const GemmP& p, int64_t row, int col, float (&v)[NV]) {
  if (!p.drop_thr24) return;
  const int64_t i0 = row * p.drop_ld + col;
  const uint32_t inner = drop_inner((uint32_t)((uint64_t)i0 >> 32), p.drop_seed);
  const uint32_t inner2 = drop_inner((uint32_t)((uint64_t)(i0 + NV - 1) >> 32), p.drop_seed);     // the NV indices straddle a 2^32 boundary at most once
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int64_t i = i0 + j;
    const uint32_t in_ = ((uint64_t)i >> 32) == ((uint64_t)i0 >> 32) ? inner : inner2;
    v[j] = dropout_keep_lo((uint32_t)i, in_, p.drop_thr24) ? round_as<T>(v[j]) * p.drop_scale : 0.f;
  }
}

// SL_POST_DROPOUT / SL_POST_GELU_BWD on NV columns starting at `col` (NV = 4 rows epilogue, 1 direct epilogue)
template <typename T, int NV>
__device__ __forceinline__ void post_apply(const GemmP& p, int64_t row, int col, float (&v)[NV]) {
  if (p.post == SL_POST_DROPOUT) {
    post_drop<T, NV>(p, row, col, v);
  } else if (p.post == SL_POST_GELU_BWD) {
    if (p.drop_thr24) {
      post_drop<T, NV>(p, row, col, v);
    }
    const T* pre = (const T*)p.post_in + row * p.post_ld + col;
#pragma unroll
    for (int j = 0; j < NV; ++j) v[j] = round_as<T>(v[j]) * gelu_grad(to_f32(pre[j]));
  }
}

// SL_POST_SILU_MUL_BWD: d = (A.W^T)[row][col], gate / up pre-activations from gu (interleaved [16 gate | 16 up]) -> d gate, d up
template <typename T>
__device__ __forceinline__ void post_silu_bwd(float d, float g, float u, float& dg, float& du) {
  d = round_as<T>(d);
  const float sg = 1.0f / (1.0f + __expf(-g));
  dg = d * u * sg * (1.0f + g * (1.0f - sg));
  du = d * g * sg;
}

// ----------------------------------------------------------------------------------------------
// LDS tile geometry shared by the tiled and streaming kernels: rows of 128 bytes of K, XOR-swizzled
// ----------------------------------------------------------------------------------------------
constexpr int TBM = 128, TBN = 128, TROWB = 128;  // tile rows, tile cols, bytes of K per LDS row

// byte offset of 16-byte chunk `ch` (0..7) of tile row `row` in a [128][128 B] swizzled LDS tile
__device__ __forceinline__ int lds_off(int row, int ch) { return row * TROWB + ((ch ^ (row & 7)) << 4); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// 256 x 256 tile geometry and the stream-K workspace layout (kernels: gemm256.hip; launch decisions: gemm.hip)
constexpr int XBM = 256, XBN = 256;
constexpr size_t SK_FLAG_BYTES = 1024, SK_SLOT_BYTES = (size_t)XBM * XBN * 4;
constexpr int SK_WS_SLOTS = 512;      // workspace = flags + 512 slots (128 MiB): the stream-K form uses <= 256 of them, three fp32 partial products of 3 200 x 3 072 need 450

// gemm256.hip: launch one of the 256-tile kernels (explicitly instantiated for bf16 / fp32 x NONE / GELU / SILU_MUL)
enum : int { SL_T256_PHASED = 0, SL_T256_PHASED_SW = 1, SL_T256_PLAIN = 2, SL_T256_PLAIN_SW = 3, SL_T256_SK = 4, SL_T256_SK_SW = 5, SL_T256_DBG = 16 };
template <typename T, int ACT>
int sl_gemm256_launch(const GemmP& p, int kind, dim3 grid, void* sk_ws, hipStream_t st);
// gemm128.hip: the 128-tile kernel with a ring of `stages` K slabs (3 or 4), one block per CU; whole-slab untransposed ungrouped 2-byte products
template <typename T, int ACT>
int sl_gemm128_ring_launch(const GemmP& p, int stages, dim3 grid, hipStream_t st);
// gemm_tt.hip: the weight-gradient kernel on token-major operands, grid (tiles, K runs)
int sl_gemm_tt_kernel_launch(const GemmP& p, int nt, int S, int slabs_per_run, hipStream_t st, int batch = 1);

// decode-side fused inputs of the weight-streaming kernels (sl_gemm_fused on the device side)
struct SkinnyX {
  const float* cos; const float* sin;      // (rope_len, 64) tables
  const int32_t* pos; const int32_t* seq;  // per activation row: position / cache slot
  void* kc; void* vc;                      // this layer's caches (slots, n_kv, max_ctx, 128)
  int nh, nkv, max_ctx, fuse_rms;
  float eps;
  const float* rstd_in;   // per-row RMSNorm scale computed by the producer of A (replaces the in-kernel statistics)
  float* rstd_out;        // K-split reduce kernel: also emit rsqrt(mean(out_row^2) + eps) of the rows it stores
  void* norm_out;         // ... and the normalised rows themselves: gain * round(row * rstd), row stride N (sl_gemm_fused.norm_out)
  const void* norm_gain;
};

// gemm_stream.hip: packed-weight streaming GEMM for 16 < M <= 256 (large-batch decode)
int sl_gemm_stream_launch(GemmP& p, const SkinnyX& sx, int dtype, int act, void* split_ws, size_t split_ws_bytes, hipStream_t st);
size_t sl_gemm_stream_ws_bytes(int M, int N, int K, int dtype);
int sl_gemm_stream_splits(int M, int N, int K, int dtype);   // K splits the heuristic picks when a workspace is supplied
