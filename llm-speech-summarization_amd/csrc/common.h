// common.h — shared device/host helpers for libspeechllm (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/speechllm.h"

// ----------------------------------------------------------------------------------------------
// error plumbing
// ----------------------------------------------------------------------------------------------
void sl_set_error(const char* fmt, ...);

#define SL_CHECK_ARG(cond, ...)      \
  do {                               \
    if (!(cond)) {                   \
      sl_set_error(__VA_ARGS__);     \
      return SL_ERR_ARG;             \
    }                                \
  } while (0)

#define SL_CHECK_LAUNCH(what)                                                        \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      sl_set_error("%s: %s", what, hipGetErrorString(e__));                          \
      return SL_ERR_LAUNCH;                                                          \
    }                                                                                \
  } while (0)

#define SL_HIP(call)                                                                 \
  do {                                                                               \
    hipError_t e__ = (call);                                                         \
    if (e__ != hipSuccess) {                                                         \
      sl_set_error("%s failed: %s", #call, hipGetErrorString(e__));                  \
      return SL_ERR_LAUNCH;                                                          \
    }                                                                                \
  } while (0)

#define SL_TRY(call)            \
  do {                          \
    int r__ = (call);           \
    if (r__ != 0) return r__;   \
  } while (0)

// ----------------------------------------------------------------------------------------------
// element types.  bf16 is carried as its 16 raw bits; conversion f32->bf16 is a plain cast so hipcc
// emits v_cvt_pk_bf16_f32 (round-to-nearest-even, NaN-preserving: MI355X_MICROARCH "Correctness").
// ----------------------------------------------------------------------------------------------
struct bf16_t {
  uint16_t bits;
};

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// 16-byte streaming load with the non-temporal hint (weights read once per token: guide nt-weights)
__device__ __forceinline__ uint4 ld_nt16(const void* p) {
#ifdef SL_W_PLAIN      // A/B builds only (tools/ab): default-policy weight loads
  return *(const uint4*)p;
#else
  u32x4_t v = __builtin_nontemporal_load((const u32x4_t*)p);
  return make_uint4(v.x, v.y, v.z, v.w);
#endif
}

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __builtin_bit_cast(float, b << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;
  return __builtin_bit_cast(uint16_t, h);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// one v_cvt_pk_bf16_f32 (converting the halves separately and OR-ing them costs four instructions per pair)
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
  const bf16x2_t v = __builtin_convertvector(f32x2_t{lo, hi}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, v);
}

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return bf16_bits_to_f32(v.bits); }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return bf16_t{f32_to_bf16_bits(v)}; }

// 16-byte vector of T: 4 floats or 8 bf16.  VEC = elements per 16 bytes.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void unpack(const uint4& u, float* f) {
    f[0] = __builtin_bit_cast(float, u.x); f[1] = __builtin_bit_cast(float, u.y);
    f[2] = __builtin_bit_cast(float, u.z); f[3] = __builtin_bit_cast(float, u.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(__builtin_bit_cast(uint32_t, f[0]), __builtin_bit_cast(uint32_t, f[1]),
                      __builtin_bit_cast(uint32_t, f[2]), __builtin_bit_cast(uint32_t, f[3]));
  }
};
template <> struct Vec16<bf16_t> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void unpack(const uint4& u, float* f) {
    f[0] = bf16_bits_to_f32(u.x & 0xffffu); f[1] = bf16_bits_to_f32(u.x >> 16);
    f[2] = bf16_bits_to_f32(u.y & 0xffffu); f[3] = bf16_bits_to_f32(u.y >> 16);
    f[4] = bf16_bits_to_f32(u.z & 0xffffu); f[5] = bf16_bits_to_f32(u.z >> 16);
    f[6] = bf16_bits_to_f32(u.w & 0xffffu); f[7] = bf16_bits_to_f32(u.w >> 16);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return make_uint4(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3]), pack2_bf16(f[4], f[5]),
                      pack2_bf16(f[6], f[7]));
  }
};

// ----------------------------------------------------------------------------------------------
// MFMA 16x16 tile step, dtype-generic.  One "k-step" consumes 64 bytes of K per operand row:
//   bf16: 32 k  -> one v_mfma_f32_16x16x32_bf16
//   f32 : 16 k  -> four v_mfma_f32_16x16x4_f32 (exact fp32 fma chain)
// Lane l = (r = l & 15, q = l >> 4) supplies, for BOTH operands, the 16 bytes at k-offset q*VEC of
// row r of its operand tile (A: output row, B: output column).  For f32 the four MFMAs take element
// s of each lane's float4, i.e. k = 4q + s: a k-permutation shared by A and B, so the sum is exact.
// Accumulator (C/D) layout: lane (c = l & 15, q = l >> 4) holds D[row 4q + i][col c], i = 0..3.
// ----------------------------------------------------------------------------------------------
template <typename T> struct MMA;
template <> struct MMA<bf16_t> {
  static constexpr int KSTEP = 32;
  static __device__ __forceinline__ void step(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                  acc, 0, 0, 0);
  }
};
template <> struct MMA<float> {
  static constexpr int KSTEP = 16;
  static __device__ __forceinline__ void step(f32x4& acc, const uint4& a, const uint4& b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
  }
};

// ----------------------------------------------------------------------------------------------
// math
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// GELU for the bf16 instantiations: erf by Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, far inside bf16's 4e-3 output
// rounding) — rcp + exp2 + 7 fma instead of libm's branchy erff; erf evaluations were ~11 % of an encoder pass
// (conv stack LayerNorm+GELU, 24 x FFN1).  gelu(x) = x - x*w for x >= 0 and x*w for x < 0, w = 0.5 erfc(|x|/sqrt 2).
__device__ __forceinline__ float gelu_fast(float x) {
  // 20 issue slots per value (two of them quarter rate) instead of 23: 0.5 folded into the polynomial, and the two branches x - x w (x >= 0) /
  // x w (x < 0) written as max(x, 0) - |x| w — one v_max and one v_fma with source modifiers instead of multiply, compare, select, subtract
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
  float pl = __builtin_fmaf(0.5f * 1.061405429f, t, 0.5f * -1.453152027f);
  pl = __builtin_fmaf(pl, t, 0.5f * 1.421413741f);
  pl = __builtin_fmaf(pl, t, 0.5f * -0.284496736f);
  pl = __builtin_fmaf(pl, t, 0.5f * 0.254829592f);
  const float w = pl * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);     // 0.5 erfc(|x| / sqrt 2)
  return __builtin_fmaf(-fabsf(x), w, fmaxf(x, 0.f));
}
// counter-based dropout mask shared by sl_dropout, the attention kernels and sl_attn_dropout_bwd (train_ops.hip has the story)
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ bool dropout_keep(int64_t i, uint64_t seed, uint32_t thr24) {
  const uint32_t h = lowbias32((uint32_t)i ^ lowbias32((uint32_t)((uint64_t)i >> 32) ^ (uint32_t)seed) ^ (uint32_t)(seed >> 32));
  return (h >> 8) >= thr24;
}
// the same mask in two steps, for loops in which the upper half of the index takes one or two values: inner = drop_inner(hi, seed) once,
// then dropout_keep_lo(lo, inner, thr) per element (a 32-bit integer multiply is a quarter-rate instruction: lowbias32 costs ~56 cycles)
__device__ __forceinline__ uint32_t drop_inner(uint32_t hi, uint64_t seed) { return lowbias32(hi ^ (uint32_t)seed) ^ (uint32_t)(seed >> 32); }
__device__ __forceinline__ bool dropout_keep_lo(uint32_t lo, uint32_t inner, uint32_t thr24) { return (lowbias32(lo ^ inner) >> 8) >= thr24; }
template <typename T> __device__ __forceinline__ float gelu_act(float x) {
  if constexpr (sizeof(T) == 2) return gelu_fast(x);
  else return gelu_erf(x);
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + __expf(-x)); }
// d gelu(u) / du (exact erf form, hf:activations.py GELUActivation): sl_gelu_bwd and the GEMM epilogue's SL_POST_GELU_BWD
__device__ __forceinline__ float gelu_grad(float u) {
  const float cdf = 0.5f * (1.0f + erff(u * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * u * u);
  return cdf + u * pdf;
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

constexpr int SL_MAX_DEVICES = 64;      // per-thread, per-device helper objects (capture streams, side streams) are kept in arrays of this size

static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// dtype dispatch on the host
#define SL_DISPATCH_DTYPE(dtype, T, ...)                      \
  do {                                                        \
    if ((dtype) == SL_F32) {                                  \
      using T = float;                                        \
      __VA_ARGS__;                                            \
    } else if ((dtype) == SL_BF16) {                          \
      using T = bf16_t;                                       \
      __VA_ARGS__;                                            \
    } else {                                                  \
      sl_set_error("unknown dtype %d", (int)(dtype));         \
      return SL_ERR_ARG;                                      \
    }                                                         \
  } while (0)

static inline size_t sl_dtype_size(int dtype) { return dtype == SL_F32 ? 4 : 2; }

// ----------------------------------------------------------------------------------------------
// tuning switches: environment variables read ONCE (first use) into this table, never on a dispatch path;
// sl_tuning_reload() (api.hip, exported for tools/tune_*.py) re-reads them for in-process A/B runs
// ----------------------------------------------------------------------------------------------
struct SlEnv {
  int attn_fwd_st;         // SL_ATTN_FWD_ST       (default 1; 2 measured equal, profiles/r06_l_attn_fwd_st_ab.txt) head_dim-64 attention forward: 64-key tiles per staged block
  int conv0_fold;          // SL_CONV0_FOLD        (default 1) conv0 backward of a batch: the block's four waves fold their sums through LDS and flush once, >= 4 strips per wave; 0 = a flush per wave of a 512-block grid (A/B)
  int enc_wt_ahead;        // SL_ENC_WT_AHEAD      (default 1) encoder tape: the layers' transposed weights for the data-gradient products are made in one batched launch on a side stream beside the forward; 0 = a transpose in front of each product
  int glds_ring;           // SL_GLDS_RING         (default 4) bf16 products of <= 256 tiles of 128 x 128 (one block per CU) run the ring form of the 128-tile kernel (gemm128.hip): 4 / 3 = stages of 32 KiB, 104 = four stages without the software-pipelined fragment reads, 0 = the two-stage kernel (A/B)
  int tt_batched;          // SL_TT_BATCHED        (default 1) batched both-transposed products of 64 or 128 k output rows (the positional conv's weight gradient) run on the token-major kernel, batch index on blockIdx.z (0: the register-staged loader — A/B)
  int wgrad_stream_min_tok; // SL_WGRAD_STREAM_MIN_TOK (default 0 = 512; 2 048 until round 6) encoder tape: from this many token rows the parameter-gradient products run on the library's side stream (tuning)
  int attn_bwd_both;       // SL_ATTN_BWD_BOTH     (default 1) attention backward: dK / dV and dQ blocks in ONE launch where neither pass fills the chip (<= 1 024 blocks together: the per-rank KD window); 0 = two launches (A/B, same bits)
  int ring_max_tiles;      // SL_GLDS_RING_MAX_TILES (default 0 = the CU count) most 128 x 128 tiles (x batch) of a product the ring form of the 128-tile kernel takes; above the CU count its blocks run in two rounds (tuning)
  int split_k256;          // SL_SPLIT_K256        (default 1) 256-tile products that fill 50-66 % of one round and hand their K runs to the consumer (deferred_splits) are cut into S uneven runs of whole slabs (gemm.hip splitk256_runs); 0 = off (A/B)
  int ln_colred_inkernel;  // SL_LN_COLRED_INKERNEL (default 0: measured SLOWER, per-rank KD window 30.8 against 29.7 ms — one block pulls every record through one CU behind an atomic round trip; profiles/r06_am_*) 1 = training tapes: the LayerNorm backward's last-arriving block sums the per-block dgamma / dbeta records itself (52 launches fewer per window, same bits)
  int tt_ring;             // SL_TT_RING           (default 1) token-major weight-gradient launches of <= 256 blocks run the ring form of gemm_tiled_tt_kernel (0: the two-stage kernel — A/B)
  int glds_dmab;           // SL_GLDS_DMAB         (default 0: measured equal to -6 % at two blocks per CU, profiles/r06_ah_*) 1 = the two-stage 128-tile kernel issues the next slab's DMA requests between the MFMAs of the current one (0: in a burst at the top of the iteration)
  int splitk_slots;        // SL_SPLITK_SLOTS      (default 0 = rule) block slots the plain split-K rule fills with K runs: 256 = one block per CU (ring form), 512 = two (two-stage kernel)
  int rms_bwd_lean;        // SL_RMSBWD_LEAN       (default 2) RMSNorm backward: x / dy kept as loaded 16-byte vectors (no float copies: ~130 registers instead of 256, 3 waves per SIMD instead of 1); 2 = and four rows per block at every row count, 3 = eight; 0 = the float-copy form
  int decode_prefetch;     // SL_DECODE_PREFETCH   (default 0: measured 2.2 x SLOWER, profiles/r06_j_decode_prefetch_ab.txt) small-batch decode graphs with a weight-prefetch branch two matrices ahead of the chain (runtime.hip DecodePrefetch)
  int attn_bwd_kf;         // SL_ATTN_BWD_KF       (default 0 = by shape) 16-row fragments per wave in the attention-backward kernels: 1 / 2 force a form
  int tape_fuse;           // SL_TAPE_FUSE         (default 1) training tapes: dropout / GELU' / SwiGLU' / bias-gradient passes inside the GEMM and norm-backward kernels (0: the unfused launch sequence, A/B + parity tests)
  int compact_pin;         // SL_COMPACT_PIN       (default 1) a compacting generation keeps the kernel family of its first batch (0: each rung picks its own — A/B only)
  int stream_min_m;        // SL_STREAM_MIN_M      (default 26) packed-weight products with more rows than this take the LDS-staged streaming kernels
  int disable_t256;        // SL_DISABLE_T256
  int t256_min_tiles;      // SL_T256_MIN_TILES    (default 512)
  int t256_min_k;          // SL_T256_MIN_K        (default 1024)
  int t256_by_rounds_pad;  // SL_T256_BY_ROUNDS_PAD 1 (default) = a product chosen for the 256-tile kernel by whole rounds of tiles may pad its rows to 256 freely (634 rows -> 768), 0 = round-4 rule (<= 1/8 more rows than the 128-row padding)
  int t256_phased;         // SL_T256_PHASED       1 (default) = the 256-tile GEMM runs the staggered two-phase main loop, 0 = the round-3 one-barrier-per-slab loop (A/B)
  int disable_glds;        // SL_DISABLE_GLDS      0 / 1 / 2
  int direct_epilogue;     // SL_DIRECT_EPILOGUE
  int gemm_gm;             // SL_GEMM_GM           (default 8)
  int attn_full_min;       // SL_ATTN_FULL_MIN     (default 32)
  int attn_force_split;    // SL_ATTN_FORCE_SPLIT
  int attn_decode_ks;      // SL_ATTN_DECODE_KS    keys per chunk of the single-pass decode attention: 128 (default) or 64 (a block small enough to share a CU with a 256-tile GEMM block of another stream)
  int attn_split_merge;    // SL_ATTN_SPLIT_MERGE    split attention merges its partial records inside the split launch: -1 (default) = where B * n_kv <= 32, 0 = never, 1 = always
  int attn_generic;        // SL_ATTN_GENERIC
  int attn_qt;             // SL_ATTN_QT
  int norm_single_row;     // SL_NORM_SINGLE_ROW
  int no_ln_fold;          // SL_NO_LN_FOLD        1 = the encoder runs its LayerNorm kernels even when folded weights are supplied (A/B)
  int no_wgrad_stream;     // SL_NO_WGRAD_STREAM   1 = the encoder backward keeps its parameter-gradient products on the caller's stream (A/B)
  int gemm_log;            // SL_GEMM_LOG          1 = every sl_gemm* call prints its shape and flags on stderr (shape census for tuning)
  int no_swap_epilogue;    // SL_NO_SWAP_EPILOGUE  1 = the 256-tile GEMM keeps the LDS-turned rows epilogue where the swapped-operand form applies (A/B)
  int decode_tiled;        // SL_DECODE_TILED      1 (default) = decode steps above ~900 rows run o and gate/up on the row-major 256-tile kernels, 0 = streaming forms
  int prefill_share_prefix; // SL_PREFILL_SHARE_PREFIX 1 (default) = prefill computes a shared prompt prefix (sl_kv_cache.shared_prefix) once per batch, 0 = per sequence (A/B)
  int skinny_alt;          // SL_SKINNY_ALT        1 = o / down at M <= 8 keep the two-steps-in-flight structure of the larger row counts (A/B)
  int tt_max_splits;       // SL_TT_MAX_SPLITS     8 (default): most K runs of a token-major weight-gradient product (tuning)
  int lnbwd_nw;            // SL_LNBWD_NW          16 (default) | 8 | 4: waves per block of the LayerNorm backward for rows <= 1 024 elements (tuning)
  int wgrad_tr;            // SL_WGRAD_TR          1 (default) = weight gradients read dY and X as stored (gemm_tiled_tt_kernel), 0 = through K-contiguous transposed copies (A/B)
  int split_k;             // SL_SPLIT_K           1 (default) = products of few tiles whose caller supplies a workspace (sk_ws) run as S batched K runs + a fixed-order reduce launch, 0 = off (A/B)
  int stream_k;            // SL_STREAM_K          stream-K form of the 256-tile GEMM when a workspace is supplied: 0 = never, 1 = by rule (default), 2 = whenever the form allows
  int gemm_ko;             // SL_GEMM_KO           debug builds (-DSL_GEMM_DEBUG): knock-out bits of the phased 256-tile GEMM (1 reads, 2 DMA, 4 MFMAs)
  unsigned long long gemm_stamp_ptr;   // SL_GEMM_STAMP_PTR  device address (hex) of a uint32 buffer: the phased 256-tile GEMM launches its instrumented build and
                                       // leaves in-kernel cycle stamps there (tools/gemm_stamps.py); 0 = off
  int stream_splits, stream_nwv, stream_mt;   // SL_STREAM_CFG "splits,nwv[,mt]" (0 = not set)
  int stream_wide;         // SL_STREAM_WIDE       0 = never use the 256 x 128 streaming block, 1 = default rule, 2 = whenever it applies
  int stream_wsplits;      // SL_STREAM_WSPLITS    K splits of the 256 x 128 form when its blocks do not cover the CUs (0 = rule)
  int stream_fixup;        // SL_STREAM_FIXUP      1 = K splits of the 256 x 128 form are closed inside the kernel (default 0: reduce launch, faster)
  int stream_nl;           // SL_STREAM_NL         loader waves of the 128-row streaming GEMM blocks (0 = default)
};
const SlEnv& sl_env();

// Kernel-family pin (api.hip).  Every choice of the decode step that depends on the row count — skinny / streaming / tiled GEMM form,
// K-split counts, the RMSNorm-scale chain, single-pass or split attention, fused lm_head top-1 — is made on sl_family_rows(M) =
// max(M, pinned rows of the calling thread).  A compacting generation pins the rows of the batch it STARTED with (sl_generate): the
// live rows shrink, the arithmetic each row goes through does not change, so a sequence's bf16 ids cannot depend on when its
// neighbours finish (tests/test_fullsize_gpu.py::test_configs1_compacted_batch_ids_equal_uncompacted_at_bench_stop_mix).
// train_ops.hip: y (cols, ld_out) = x (rows, cols)^T for a list of matrices in one launch per SL_TRANSPOSE_BATCH records (sl_transpose_pad's
// argument rules per record; tiles_r is filled by the launcher)
#define SL_TRANSPOSE_BATCH 64
struct SlTransposeRec {
  const void* x; void* y;
  int64_t ldx, ldy;
  int32_t rows, cols, ld_out, tiles_r;
};
struct SlTransposeBatch {
  SlTransposeRec rec[SL_TRANSPOSE_BATCH];
  int32_t first_block[SL_TRANSPOSE_BATCH];
  int32_t n, pad;
};
int sl_transpose_pad_batch_impl(const SlTransposeRec* recs, int n, int32_t dtype, sl_stream stream);
bool sl_gemm_post_ok(int64_t M, int N, int K, int dtype);      // gemm.hip: a product of this shape may carry sl_gemm_ex_args.post_op / colsum_out
int sl_family_rows(int rows);
int sl_family_pin(int rows);      // returns the previous pin (0 = none)
struct SlFamilyPin {
  int prev;
  explicit SlFamilyPin(int rows) : prev(sl_family_pin(rows)) {}
  ~SlFamilyPin() { sl_family_pin(prev); }
};
