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
// LDS tile geometry shared by the tiled and streaming kernels: rows of 128 bytes of K, XOR-swizzled
// ----------------------------------------------------------------------------------------------
constexpr int TBM = 128, TBN = 128, TROWB = 128;  // tile rows, tile cols, bytes of K per LDS row

// byte offset of 16-byte chunk `ch` (0..7) of tile row `row` in a [128][128 B] swizzled LDS tile
__device__ __forceinline__ int lds_off(int row, int ch) { return row * TROWB + ((ch ^ (row & 7)) << 4); }

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

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
