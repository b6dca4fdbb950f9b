// train_ops.hip — kernels of the knowledge-distillation step (ref:trainer.py:270-374) that are not GEMMs:
// activation / norm backward, explicit-softmax attention backward pieces, the KD losses with their
// gradients (ref:model/audio_llama.py:72-101, ref:utils.py:167-178, ref:trainer.py:358-370), and the
// HuBERT front-end backward (pool, strided-conv col2im, fused conv0).  All HBM-bound row kernels:
// 16-byte accesses, fp32 math, one wave per row where a row reduction is needed.
#include <stdlib.h>

#include "common.h"

constexpr int TR_MAXF = 64;  // floats per lane for row kernels (rows up to 4096 elements)

// ----------------------------------------------------------------------------------------------
// elementwise: GELU backward, axpby, SwiGLU on the 16-row interleaved gate/up layout
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ pre, T* __restrict__ dx, int64_t n) {
  constexpr int VEC = Vec16<T>::VEC;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < n; i += (int64_t)gridDim.x * 256 * VEC) {
    float a[VEC], u[VEC], o[VEC];
    Vec16<T>::unpack(*(const uint4*)(dy + i), a);
    Vec16<T>::unpack(*(const uint4*)(pre + i), u);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = a[e] * gelu_grad(u[e]);
    *(uint4*)(dx + i) = Vec16<T>::pack(o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(const T* __restrict__ x, T* __restrict__ y, float a, float b, int64_t n) {
  constexpr int VEC = Vec16<T>::VEC;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < n; i += (int64_t)gridDim.x * 256 * VEC) {
    float xv[VEC], yv[VEC];
    Vec16<T>::unpack(*(const uint4*)(x + i), xv);
    Vec16<T>::unpack(*(const uint4*)(y + i), yv);
#pragma unroll
    for (int e = 0; e < VEC; ++e) yv[e] = a * xv[e] + b * yv[e];
    *(uint4*)(y + i) = Vec16<T>::pack(yv);
  }
}

// Dropout with a counter-based mask: element i of a site is kept iff u24(mix(i, seed)) >= p * 2^24, where mix is two rounds
// of the 32-bit "lowbias" integer hash over (low word ^ hash(high word ^ seed_lo)) ^ seed_hi.  No mask tensor is stored:
// the backward pass applies the same function to the gradient (same seed), and the test oracle rebuilds the mask in numpy.
// y = (res ? res : 0) + keep * x / (1 - p)  (the residual form is HuBERT's  h = residual + dropout(sublayer(h))).
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y, int64_t n, float scale,
                                                      uint32_t thr24, uint64_t seed) {
  constexpr int VEC = Vec16<T>::VEC;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC; i < n; i += (int64_t)gridDim.x * 256 * VEC) {
    float xv[VEC], rv[VEC];
    Vec16<T>::unpack(*(const uint4*)(x + i), xv);
    if (res) Vec16<T>::unpack(*(const uint4*)(res + i), rv);
    // the chunk's VEC indices share their upper half (i is a multiple of VEC): one inner hash round per chunk, one outer round per element
    const uint32_t inner = drop_inner((uint32_t)((uint64_t)i >> 32), seed);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float d = dropout_keep_lo((uint32_t)(i + e), inner, thr24) ? xv[e] * scale : 0.f;
      xv[e] = res ? rv[e] + d : d;
    }
    *(uint4*)(y + i) = Vec16<T>::pack(xv);
  }
}

// attention-probability dropout, backward side: same mask index as the forward kernels (attention.hip)
template <typename T>
__global__ __launch_bounds__(256) void attn_dropout_bwd_kernel(const T* __restrict__ p, T* __restrict__ pd, float* __restrict__ dp, int smax,
                                                               const int32_t* __restrict__ dims, int ld, const int32_t* __restrict__ cu_q, int nh,
                                                               int nkv, int r, float scale, uint32_t thr24, uint64_t seed) {
  const int z = blockIdx.y, si = z / nkv, head = (z % nkv) * (nh / nkv) + r;
  const int n = dims[z];
  const int64_t base = (int64_t)z * smax * ld;
  const int64_t q0 = cu_q[si];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < (int64_t)n * n; i += (int64_t)gridDim.x * 256) {
    const int row = (int)(i / n), col = (int)(i % n);
    const bool keep = dropout_keep((((q0 + row) * nh + head) << 16) | col, seed, thr24);
    const int64_t o = base + (int64_t)row * ld + col;
    pd[o] = keep ? from_f32<T>(to_f32(p[o]) * scale) : from_f32<T>(0.f);
    dp[o] = keep ? dp[o] * scale : 0.f;
  }
}

// y (cols, ld_out) = x (rows, cols)^T, columns rows..ld_out-1 of y zero-filled: the K-contiguous operand copies of the weight-
// gradient products (dY^T, X^T padded to whole K slabs) and of the data-gradient products (W^T).  64 x 64 tiles through LDS,
// 16-byte global accesses on both sides.
template <typename T>
__device__ __forceinline__ void transpose_pad_tile(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int rows, int cols, int ld_out,
                                                   int tile_r, int tile_c) {
  constexpr int VEC = Vec16<T>::VEC, TS = 64;
  const int r0 = tile_r * TS, c0 = tile_c * TS;
  const int tid = threadIdx.x;
  constexpr int CPR = TS / VEC;                       // 16-byte chunks per 64-element row
  if constexpr (sizeof(T) == 2) {
    // bf16: the tile stays row-major in LDS (128-byte rows, 16-byte stores) and is read back COLUMN-wise by
    // ds_read_b64_tr_b16: a 16-lane group gathers a 4-row x 16-column block, lane i receives column i — i.e. four
    // consecutive elements of OUTPUT row c0 + i.  Two reads give a 16-byte output chunk; the four groups of a wave take
    // the r-chunks 0-7 / 8-15 / 16-23 / 24-31 of the same 16 output rows (64 contiguous bytes per row and instruction).
    // Swizzle: chunk ^ (f(row) << 1), f = ((row >> 1) & 1) | (((row >> 3) & 1) << 1): the rows a 32-lane half touches in
    // one read ({R..R+3, R+8..R+11}) land on distinct banks.  (The 2-byte LDS form below ran at 1.5 TB/s.)
    __shared__ __attribute__((aligned(16))) unsigned char tile[TS * 128];
    auto off = [](int row, int ch) { return 128 * row + 16 * (ch ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1)); };
    for (int i = tid; i < TS * CPR; i += 256) {
      const int r = i / CPR, ch = i % CPR;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (r0 + r < rows) {
        if (c0 + ch * VEC + VEC <= cols) v = *(const uint4*)(x + (int64_t)(r0 + r) * ldx + c0 + ch * VEC);
        else {
          uint32_t w4[4] = {0, 0, 0, 0};
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            if (c0 + ch * VEC + j < cols) w4[j >> 1] |= (uint32_t)((const uint16_t*)x)[(int64_t)(r0 + r) * ldx + c0 + ch * VEC + j] << (16 * (j & 1));
          v = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
      }
      *(uint4*)(tile + off(r, ch)) = v;
    }
    __syncthreads();
    typedef __attribute__((address_space(3))) void* lds_t;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_tt;
    const uint32_t base = (uint32_t)(uintptr_t)(lds_t)tile;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4, qq = li >> 2, pp = li & 3;
    const int cb = 16 * wave;                          // this wave's 16 output rows (tile columns cb .. cb + 15)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int R = 32 * half + 8 * g;                 // first of the 8 tile rows (= output columns) of this lane's chunk
      u32x2_tt lo, hi;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(base + (uint32_t)(off(R + qq, 2 * wave + (pp >> 1)) + 8 * (pp & 1))) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(base + (uint32_t)(off(R + 4 + qq, 2 * wave + (pp >> 1)) + 8 * (pp & 1))) : "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi));
      const int oc = c0 + cb + li, o = r0 + R;         // output row, first output column of the chunk
      if (oc < cols) {
        const uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
        if (o + VEC <= ld_out) *(uint4*)(y + (int64_t)oc * ldy + o) = v;
        else {   // ragged tail of the output row: element j = half (j & 1) of word j / 2 (no address of a register: that costs scratch)
          const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            if (o + j < ld_out) ((uint16_t*)y)[(int64_t)oc * ldy + o + j] = (uint16_t)(w4[j >> 1] >> (16 * (j & 1)));
        }
      }
    }
  } else {
    static_assert(sizeof(T) == 4 && VEC == 4, "the generic branch is the fp32 one");
    __shared__ float tile[TS][TS + 1];
    for (int i = tid; i < TS * CPR; i += 256) {
      const int r = i / CPR, ch = i % CPR;
      f32x4 e = {0.f, 0.f, 0.f, 0.f};
      if (r0 + r < rows) {
        if (c0 + ch * VEC + VEC <= cols) e = *(const f32x4*)(x + (int64_t)(r0 + r) * ldx + c0 + ch * VEC);
        else {
#pragma unroll
          for (int j = 0; j < VEC; ++j)
            if (c0 + ch * VEC + j < cols) e[j] = x[(int64_t)(r0 + r) * ldx + c0 + ch * VEC + j];
        }
      }
#pragma unroll
      for (int j = 0; j < VEC; ++j) tile[r][ch * VEC + j] = e[j];
    }
    __syncthreads();
    for (int i = tid; i < TS * CPR; i += 256) {
      const int c = i / CPR, ch = i % CPR;              // output row c0 + c, elements r0 + ch*VEC ..
      if (c0 + c >= cols) continue;
      f32x4 e;
#pragma unroll
      for (int j = 0; j < VEC; ++j) e[j] = tile[ch * VEC + j][c];     // rows past `rows` were loaded as zeros
      const int o = r0 + ch * VEC;
      if (o + VEC <= ld_out) *(f32x4*)(y + (int64_t)(c0 + c) * ldy + o) = e;
      else {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (o + j < ld_out) y[(int64_t)(c0 + c) * ldy + o + j] = e[j];
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y, int64_t ldy, int rows, int cols,
                                                            int ld_out) {
  transpose_pad_tile<T>(x, ldx, y, ldy, rows, cols, ld_out, blockIdx.x, blockIdx.y);
}

// up to SL_TRANSPOSE_BATCH matrices in one launch (the records travel as a kernel argument): block b belongs to the record r with
// first_block[r] <= b < first_block[r + 1].  The encoder tape turns every layer's four weight matrices for the data-gradient products this
// way, once per window, beside the forward (train_tape.hip WtCache) — 96 launches of 5-8 us on the backward's critical path before.
template <typename T>
__global__ __launch_bounds__(256) void transpose_pad_batch_kernel(SlTransposeBatch b) {
  int r = 0;
  const int blk = blockIdx.x;
#pragma unroll 1
  for (int step = SL_TRANSPOSE_BATCH / 2; step > 0; step >>= 1)
    if (r + step < b.n && b.first_block[r + step] <= blk) r += step;
  const SlTransposeRec& q = b.rec[r];
  const int local = blk - b.first_block[r];
  transpose_pad_tile<T>((const T*)q.x, q.ldx, (T*)q.y, q.ldy, q.rows, q.cols, q.ld_out, local % q.tiles_r, local / q.tiles_r);
}

// gu: (M, 2F) with blocks [16 gate | 16 up];  out/dy: (M, F)
template <typename T, bool BWD>
__global__ __launch_bounds__(256) void silu_mul_kernel(const T* __restrict__ gu, const T* __restrict__ dy, T* __restrict__ out, int64_t M, int F_) {
  constexpr int VEC = Vec16<T>::VEC;
  const int cpr = F_ / VEC;
  const int64_t total = M * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t m = i / cpr;
    const int o = (int)(i % cpr) * VEC, p = o >> 4, c = o & 15;
    const T* gp = gu + m * 2 * F_ + 32 * p + c;
    float g[VEC], u[VEC], r[VEC];
    Vec16<T>::unpack(*(const uint4*)gp, g);
    Vec16<T>::unpack(*(const uint4*)(gp + 16), u);
    if constexpr (!BWD) {
#pragma unroll
      for (int e = 0; e < VEC; ++e) r[e] = silu(g[e]) * u[e];
      *(uint4*)(out + m * F_ + o) = Vec16<T>::pack(r);
    } else {
      float d[VEC], r2[VEC];
      Vec16<T>::unpack(*(const uint4*)(dy + m * F_ + o), d);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float sg = 1.0f / (1.0f + __expf(-g[e]));
        r[e] = d[e] * u[e] * sg * (1.0f + g[e] * (1.0f - sg));  // d gate
        r2[e] = d[e] * g[e] * sg;                                 // d up
      }
      T* op = out + m * 2 * F_ + 32 * p + c;
      *(uint4*)op = Vec16<T>::pack(r);
      *(uint4*)(op + 16) = Vec16<T>::pack(r2);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// RoPE in place on the first n_rot heads of (n_tok, heads*D) rows; sign = -1 is the backward (inverse) rotation
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_inplace_kernel(T* __restrict__ x, const int32_t* __restrict__ tok_pos, const float* __restrict__ cosT,
                                                           const float* __restrict__ sinT, int64_t n_tok, int heads, int n_rot, int D, float sign) {
  constexpr int VEC = Vec16<T>::VEC;
  const int half = D / 2, cph = half / VEC;
  const int64_t total = n_tok * n_rot * cph;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % cph);
    const int h = (int)((i / cph) % n_rot);
    const int64_t t = i / ((int64_t)cph * n_rot);
    T* p = x + t * (int64_t)heads * D + (int64_t)h * D + j * VEC;
    const int pos = tok_pos[t];
    float a[VEC], b[VEC], o1[VEC], o2[VEC];
    Vec16<T>::unpack(*(const uint4*)p, a);
    Vec16<T>::unpack(*(const uint4*)(p + half), b);
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      const float c = cosT[(int64_t)pos * half + j * VEC + e], s = sign * sinT[(int64_t)pos * half + j * VEC + e];
      o1[e] = a[e] * c - b[e] * s;
      o2[e] = b[e] * c + a[e] * s;
    }
    *(uint4*)p = Vec16<T>::pack(o1);
    *(uint4*)(p + half) = Vec16<T>::pack(o2);
  }
}

// ----------------------------------------------------------------------------------------------
// LayerNorm backward (optionally through a fused GELU): dx, and fp32 dgamma/dbeta accumulated with one
// atomic per column per block.  RMSNorm backward: dx only (the LLM is frozen).
// ----------------------------------------------------------------------------------------------
// MAXF = floats a lane holds of one row (64 * MAXF >= cols), NW = waves per block.  The general form (MAXF = TR_MAXF = 64, 4 waves)
// keeps four MAXF-float arrays per lane live — 256 VGPRs, one wave per SIMD — and, because the column atomics of all blocks
// land on the same 2 x cols addresses, runs on ~one block per CU: a 7 984 x 1 024 encoder LayerNorm took 127 us for 48 MB of
// traffic (latency-bound: ~1 wave per SIMD, 8 rows one after the other).  Rows of <= 1 024 elements (every HuBERT / Whisper
// LayerNorm: 512 conv channels, hidden 1 024) take MAXF = 16 and 16 waves per block: the same ~250 blocks, hence the same
// number of atomics, but 16 rows in flight per CU instead of 4.
// keeps the compiler from carrying the converted copy of a loaded vector from one pass to the next (it would: common subexpressions)
__device__ __forceinline__ void reconvert_here(uint4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
// LEAN (rows without the fused GELU): a lane keeps its share of x and dy as LOADED (16-byte vectors) and converts them again in every
// pass instead of holding xhat and dy * gamma as floats, and beta is not needed at all: the 16-wave form has 128 registers per lane
// and the float copies made it spill 34 of them to scratch inside the row loop.  Same operations in the same order: same bits.
// optional extras of the training tapes (train_tape.hip): a second residual-branch gradient, and a second OUTPUT — the dropout of the stored dx
// under the mask of the Linear the gradient flows into next (h = h + dropout(sublayer(h)): the sublayer's output gradient is dropout(dx)),
// which saves that layer's sl_dropout launch and its read of dx
struct NormBwdX {
  const void* add2;      // dx = round(dx + add2) after the first add (the bits of a following sl_axpby(add2, dx))
  void* dx_drop;         // dropout(dx as stored) at element index row * cols + col
  uint32_t thr24; float scale; uint64_t seed;
  // dy left as the S fp32 partial products of a K-split GEMM (sl_gemm_ex_args.deferred_splits): dy = round(sum of the runs, in run order) is formed
  // while loading — the reduce launch between the product and this kernel, and its round trip of dy through memory, are gone
  const float* dy_parts; int dy_S; int64_t dy_slab;
  // LayerNorm, scratch-record form, 16-wave blocks: the block that arrives LAST at this counter (zero between launches; it leaves it zero) sums the
  // blocks' records into dgamma / dbeta itself, in norm_colreduce_kernel's order (same bits) — no second launch.  Records and counter cross the XCDs'
  // L2s as agent-scope (sc1) stores / loads drained with s_waitcnt around the arrival (the hand-off form of gemm_stream.hip).
  int32_t* colred_cnt;
};

template <typename T>
__device__ __forceinline__ uint4 norm_bwd_load_dy(const T* __restrict__ dy, const NormBwdX& ex, int64_t row, int cols, int c0) {
  if (!ex.dy_parts) return *(const uint4*)(dy + row * cols + c0);
  constexpr int VEC = Vec16<T>::VEC;
  float acc[VEC];
  const float* p0 = ex.dy_parts + row * cols + c0;
#pragma unroll
  for (int e = 0; e < VEC; e += 4) {
    const f32x4 v = *(const f32x4*)(p0 + e);
    acc[e] = v[0]; acc[e + 1] = v[1]; acc[e + 2] = v[2]; acc[e + 3] = v[3];
  }
  for (int z = 1; z < ex.dy_S; ++z) {
    const float* pz = p0 + (int64_t)z * ex.dy_slab;
#pragma unroll
    for (int e = 0; e < VEC; e += 4) {
      const f32x4 v = *(const f32x4*)(pz + e);
      acc[e] += v[0]; acc[e + 1] += v[1]; acc[e + 2] += v[2]; acc[e + 3] += v[3];
    }
  }
  return Vec16<T>::pack(acc);      // rounded to the storage type, as the reduce pass would have stored it
}

template <typename T>
__device__ __forceinline__ void norm_bwd_store(T* __restrict__ dx, const T* __restrict__ add, const NormBwdX& ex, int64_t row, int cols, int c0, float (&o)[Vec16<T>::VEC]) {
  constexpr int VEC = Vec16<T>::VEC;
  if (add) {      // rounded to the storage type first, then added: the bits of dx = round(backward) followed by axpby(add, dx)
    float av[VEC];
    Vec16<T>::unpack(*(const uint4*)(add + row * cols + c0), av);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = to_f32(from_f32<T>(o[e])) + av[e];
  }
  if (ex.add2) {
    float av[VEC];
    Vec16<T>::unpack(*(const uint4*)((const T*)ex.add2 + row * cols + c0), av);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = to_f32(from_f32<T>(o[e])) + av[e];
  }
  const uint4 pk = Vec16<T>::pack(o);
  *(uint4*)(dx + row * cols + c0) = pk;
  if (ex.dx_drop) {
    float sv[VEC];
    Vec16<T>::unpack(pk, sv);          // the values as stored
    const int64_t i0 = row * cols + c0;         // a multiple of VEC: the chunk's indices share their upper half
    const uint32_t inner = drop_inner((uint32_t)((uint64_t)i0 >> 32), ex.seed);
#pragma unroll
    for (int e = 0; e < VEC; ++e) sv[e] = dropout_keep_lo((uint32_t)(i0 + e), inner, ex.thr24) ? sv[e] * ex.scale : 0.f;
    *(uint4*)((T*)ex.dx_drop + row * cols + c0) = Vec16<T>::pack(sv);
  }
}

template <typename T, bool RMS, int MAXF, int NW, bool LEAN = false>
__global__ __launch_bounds__(64 * NW) void norm_bwd_kernel(const T* __restrict__ x, const T* __restrict__ g, const T* __restrict__ b,
                                                           const T* __restrict__ dy, T* __restrict__ dx, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int64_t rows, int cols, float eps, int gelu, int rows_per_block,
                                                           float* __restrict__ part, const T* __restrict__ add = nullptr, NormBwdX ex = NormBwdX{}) {
  // add (optional, same shape as dx, not aliasing it): dx = backward(dy) + add — the residual branch's gradient joins here instead of in a
  // separate axpby launch (one read + one write of the whole activation less per LayerNorm / RMSNorm in the training tapes)
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int MAXCH = MAXF / VEC;
  __shared__ float red[NW][64 * VEC + 4];  // per-wave column partials of one 64-chunk group, folded in two passes (gamma, beta)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nch = cols / VEC;
  float gg[MAXCH][VEC], bb[MAXCH][VEC], ag[MAXCH][VEC], ab[MAXCH][VEC];
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int ch = lane + 64 * i;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { gg[i][e] = 0.f; bb[i][e] = 0.f; ag[i][e] = 0.f; ab[i][e] = 0.f; }
    if (ch < nch) {
      Vec16<T>::unpack(*(const uint4*)(g + ch * VEC), gg[i]);
      if (!RMS && !LEAN) Vec16<T>::unpack(*(const uint4*)(b + ch * VEC), bb[i]);
    }
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block) < rows ? (r0 + rows_per_block) : rows;
  if constexpr (LEAN) {
    // (the gain, too, stays as loaded where no column partials are kept — RMSNorm: 24 registers instead of 48 at 3 072 columns)
    uint4 gl[RMS ? MAXCH : 1];
    if constexpr (RMS) {
#pragma unroll
      for (int i = 0; i < MAXCH; ++i) gl[i] = lane + 64 * i < nch ? *(const uint4*)(g + (lane + 64 * i) * VEC) : make_uint4(0, 0, 0, 0);
    }
    auto gain = [&](int i, float (&out)[VEC]) {
      if constexpr (RMS) { reconvert_here(gl[i]); Vec16<T>::unpack(gl[i], out); }
      else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) out[e] = gg[i][e];
      }
    };
    for (int64_t row = r0 + wave; row < r1; row += NW) {
      uint4 xr[MAXCH], dr[MAXCH];
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
          xr[i] = *(const uint4*)(x + row * cols + ch * VEC);
          dr[i] = norm_bwd_load_dy<T>(dy, ex, row, cols, ch * VEC);
          float xv[VEC];
          Vec16<T>::unpack(xr[i], xv);
#pragma unroll
          for (int e = 0; e < VEC; ++e) s += RMS ? xv[e] * xv[e] : xv[e];
        }
      }
      s = wave_sum(s);
      float mean = 0.f, rstd;
      if constexpr (RMS) {
        rstd = rsqrtf(s / (float)cols + eps);
      } else {
        mean = s / (float)cols;
        float s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXCH; ++i)
          if (lane + 64 * i < nch) {
            float xv[VEC];
            reconvert_here(xr[i]);
            Vec16<T>::unpack(xr[i], xv);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = xv[e] - mean; s2 += d * d; }
          }
        rstd = rsqrtf(wave_sum(s2) / (float)cols + eps);
      }
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXCH; ++i)
        if (lane + 64 * i < nch) {
          float xv[VEC], dv[VEC], gv[VEC];
          reconvert_here(xr[i]);
          reconvert_here(dr[i]);
          Vec16<T>::unpack(xr[i], xv);
          Vec16<T>::unpack(dr[i], dv);
          gain(i, gv);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float xh = (xv[e] - mean) * rstd;
            const float d = dv[e];
            ag[i][e] += d * xh;
            ab[i][e] += d;
            const float dg = d * gv[e];
            s1 += dg;
            s2 += dg * xh;
          }
        }
      s1 = wave_sum(s1) / (float)cols;
      s2 = wave_sum(s2) / (float)cols;
#pragma unroll
      for (int i = 0; i < MAXCH; ++i) {
        const int ch = lane + 64 * i;
        if (ch < nch) {
          float xv[VEC], dv[VEC], gv[VEC], o[VEC];
          reconvert_here(xr[i]);
          reconvert_here(dr[i]);
          Vec16<T>::unpack(xr[i], xv);
          Vec16<T>::unpack(dr[i], dv);
          gain(i, gv);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const float xh = (xv[e] - mean) * rstd;
            const float dg = dv[e] * gv[e];
            o[e] = RMS ? rstd * (dg - xh * s2) : rstd * (dg - s1 - xh * s2);
          }
          norm_bwd_store<T>(dx, add, ex, row, cols, ch * VEC, o);
        }
      }
    }
  } else
  for (int64_t row = r0 + wave; row < r1; row += NW) {
    float xv[MAXCH][VEC], dv[MAXCH][VEC];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        Vec16<T>::unpack(*(const uint4*)(x + row * cols + ch * VEC), xv[i]);
        Vec16<T>::unpack(norm_bwd_load_dy<T>(dy, ex, row, cols, ch * VEC), dv[i]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += RMS ? xv[i][e] * xv[i][e] : xv[i][e];
      }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if constexpr (RMS) {
      rstd = rsqrtf(s / (float)cols + eps);
    } else {
      mean = s / (float)cols;
      float s2 = 0.f;
#pragma unroll
      for (int i = 0; i < MAXCH; ++i)
        if (lane + 64 * i < nch) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) { const float d = xv[i][e] - mean; s2 += d * d; }
        }
      rstd = rsqrtf(wave_sum(s2) / (float)cols + eps);
    }
    // xhat in xv, dy*(gelu') in dv, accumulate parameter grads, row sums
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i)
      if (lane + 64 * i < nch) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const float xh = (xv[i][e] - mean) * rstd;
          float d = dv[i][e];
          if (!RMS && gelu) d *= gelu_grad(xh * gg[i][e] + bb[i][e]);
          ag[i][e] += d * xh;
          ab[i][e] += d;
          const float dg = d * gg[i][e];
          s1 += dg;
          s2 += dg * xh;
          xv[i][e] = xh;
          dv[i][e] = dg;
        }
      }
    s1 = wave_sum(s1) / (float)cols;
    s2 = wave_sum(s2) / (float)cols;
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + 64 * i;
      if (ch < nch) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = RMS ? rstd * (dv[i][e] - xv[i][e] * s2) : rstd * (dv[i][e] - s1 - xv[i][e] * s2);
        norm_bwd_store<T>(dx, add, ex, row, cols, ch * VEC, o);
      }
    }
  }
  if (RMS || !dgamma) return;
  // fold the waves' column partials, then one atomic per column per block — or, with a scratch buffer (`part`), one plain store
  // per column per block into this block's record [2][cols], which norm_colreduce_kernel sums: the atomics of all blocks land on the
  // same 2 x cols addresses and serialise at the memory side (~0.3 us per block: 79 us at 256 blocks for a 7 984 x 1 024 LayerNorm
  // that moves 48 MB), the records do not
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    float* dst = part ? part + ((int64_t)blockIdx.x * 2 + pass) * cols : (pass == 0 ? dgamma : dbeta);
#pragma unroll
    for (int i = 0; i < MAXCH; ++i) {
      const int ch = lane + 64 * i;
      __syncthreads();
      if (ch < nch) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[wave][(lane * VEC + e)] = pass == 0 ? ag[i][e] : ab[i][e];
      }
      __syncthreads();
      if (wave == 0 && ch < nch) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int k = lane * VEC + e;
          float t = red[0][k];
#pragma unroll
          for (int w = 1; w < NW; ++w) t += red[w][k];
          if (part) {
            if (ex.colred_cnt) __hip_atomic_store(dst + ch * VEC + e, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else dst[ch * VEC + e] = t;
          } else atomicAdd(dst + ch * VEC + e, t);
        }
      }
    }
  }
  if constexpr (NW == 16) {
    if (part && ex.colred_cnt) {
      __shared__ int last_s;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's record stores have reached the memory side
      __syncthreads();
      if (threadIdx.x == 0) {
        const int old = atomicAdd(ex.colred_cnt, 1);
        const int last = old == (int)gridDim.x - 1;
        if (last) __hip_atomic_store(ex.colred_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every block has arrived: leave the counter as found
        last_s = last;
      }
      __syncthreads();
      if (!last_s) return;
      // norm_colreduce_kernel's sum, chunk of 64 record entries after chunk: 16 stripes of blocks, eight records in flight per thread, the stripes
      // folded in stripe order
      const int nblocks = (int)gridDim.x, stripe = wave;
      float (*rr)[64 * VEC + 4] = red;
      for (int c0 = 0; c0 < 2 * cols; c0 += 64) {
        const int i = c0 + lane;
        const int ic = i < 2 * cols ? i : 2 * cols - 1;
        float t = 0.f;
        for (int b0 = stripe; b0 < nblocks; b0 += 16 * 8) {
          float v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int bq = b0 + 16 * u;
            v[u] = bq < nblocks ? __hip_atomic_load(part + (int64_t)bq * 2 * cols + ic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) t += v[u];
        }
        __syncthreads();
        rr[stripe][lane] = t;
        __syncthreads();
        if (stripe == 0 && i < 2 * cols) {
          float sum = rr[0][lane];
#pragma unroll
          for (int w2 = 1; w2 < 16; ++w2) sum += rr[w2][lane];
          float* dstp = i < cols ? dgamma + i : dbeta + (i - cols);
          *dstp += sum;
        }
      }
    }
  }
}

// second pass of the scratch-buffer form: dgamma[c] += sum over blocks of part[b][0][c], dbeta likewise.  Block = 64 consecutive
// entries of the [2 * cols] record x 16 stripes of blocks (a wave reads 256 contiguous bytes of one record), eight records in
// flight per thread; the stripes meet in LDS in a fixed order, so the sum is reproducible.
__global__ __launch_bounds__(1024) void norm_colreduce_kernel(const float* __restrict__ part, int nblocks, int cols, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, stripe = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;          // [0, 2 * cols): pass = i / cols
  const int ic = i < 2 * cols ? i : 2 * cols - 1;
  float t = 0.f;
  for (int b0 = stripe; b0 < nblocks; b0 += 16 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int b = b0 + 16 * u;
      v[u] = b < nblocks ? part[(int64_t)b * 2 * cols + ic] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) t += v[u];
  }
  red[stripe][lane] = t;
  __syncthreads();
  if (stripe == 0 && i < 2 * cols) {
    float s = red[0][lane];
#pragma unroll
    for (int w = 1; w < 16; ++w) s += red[w][lane];
    float* dst = i < cols ? dgamma + i : dbeta + (i - cols);
    *dst += s;
  }
}

// column sums of a (rows, cols) matrix into fp32 (bias gradients), one atomic per column per block
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ out, int64_t rows, int cols,
                                                     int rows_per_block) {
  const int col = blockIdx.y * 256 + threadIdx.x;
  if (col >= cols) return;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block) < rows ? (r0 + rows_per_block) : rows;
  float s = 0.f;
  for (int64_t r = r0; r < r1; ++r) s += to_f32(x[r * ld + col]);
  atomicAdd(out + col, s);
}

// 16-byte form.  Float atomics on a handful of hot addresses serialise at the memory side (MI355X_MICROARCH "Global float
// atomics": every workgroup adding into one row is 14x slower), so the number of adders per column is kept at <= 64 row
// blocks while the COLUMNS are split over blocks to fill the chip: block = 32 chunk lanes (32 x VEC adjacent columns) x 8 row
// lanes, partial sums folded through LDS, one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void colsum_vec_kernel(const T* __restrict__ x, int64_t ld, float* __restrict__ out, int64_t rows, int cols,
                                                         int rows_per_block) {
  constexpr int VEC = Vec16<T>::VEC;
  __shared__ float red[8][32 * VEC + 1];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int ch = blockIdx.y * 32 + cl;
  const bool on = ch * VEC < cols;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block) < rows ? (r0 + rows_per_block) : rows;
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
  if (on)
    for (int64_t r = r0 + rl; r < r1; r += 8) {
      float f[VEC];
      Vec16<T>::unpack(*(const uint4*)(x + r * ld + ch * VEC), f);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += f[j];
    }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[rl][cl * VEC + j] = s[j];
  __syncthreads();
  for (int c = threadIdx.x; c < 32 * VEC; c += 256) {
    const int col = blockIdx.y * 32 * VEC + c;
    if (col >= cols) continue;
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) t += red[k][c];
    atomicAdd(out + col, t);
  }
}

// ----------------------------------------------------------------------------------------------
// explicit softmax (attention backward recomputes P): rows of fp32 scores -> P (T); and its backward
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ S, T* __restrict__ P, int64_t nrows, int rows_per_mat, int cols,
                                                           int64_t ld, float scale, int causal_shift, int causal) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int qi = (int)(row % rows_per_mat);
  const int lim = causal ? ((qi + causal_shift + 1) < cols ? (qi + causal_shift + 1) : cols) : cols;  // visible columns
  const float* s = S + row * ld;
  T* p = P + row * ld;
  float m = -INFINITY;
  for (int c = lane; c < lim; c += 64) m = fmaxf(m, s[c] * scale);
  m = wave_max(m);
  float l = 0.f;
  for (int c = lane; c < lim; c += 64) l += __expf(s[c] * scale - m);
  l = wave_sum(l);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  for (int c = lane; c < ld; c += 64) p[c] = from_f32<T>(c < lim ? __expf(s[c] * scale - m) * inv : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const T* __restrict__ P, const float* __restrict__ dP, T* __restrict__ dS, int64_t nrows,
                                                          int cols, int64_t ld, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const T* p = P + row * ld;
  const float* dp = dP + row * ld;
  float dot = 0.f;
  for (int c = lane; c < cols; c += 64) dot += to_f32(p[c]) * dp[c];
  dot = wave_sum(dot);
  for (int c = lane; c < ld; c += 64) dS[row * ld + c] = from_f32<T>(c < cols ? scale * to_f32(p[c]) * (dp[c] - dot) : 0.f);
}

// ragged forms: n_mats square matrices of sizes mat_dim[i] at a common pitch (rows_per_mat x ld)
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_var_kernel(const float* __restrict__ S, T* __restrict__ P, int64_t nrows, int rows_per_mat,
                                                               const int32_t* __restrict__ mat_dim, int64_t ld, float scale, int causal) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int qi = (int)(row % rows_per_mat), n = mat_dim[row / rows_per_mat];
  if (qi >= n) return;
  const int lim = causal ? ((qi + 1) < n ? (qi + 1) : n) : n;
  const float* s = S + row * ld;
  T* p = P + row * ld;
  float m = -INFINITY;
  for (int c = lane; c < lim; c += 64) m = fmaxf(m, s[c] * scale);
  m = wave_max(m);
  float l = 0.f;
  for (int c = lane; c < lim; c += 64) l += __expf(s[c] * scale - m);
  l = wave_sum(l);
  const float inv = l > 0.f ? 1.0f / l : 0.f;
  for (int c = lane; c < ld; c += 64) p[c] = from_f32<T>(c < lim ? __expf(s[c] * scale - m) * inv : 0.f);
}

template <typename T>
__global__ __launch_bounds__(256) void softmax_bwd_var_kernel(const T* __restrict__ P, const float* __restrict__ dP, T* __restrict__ dS, int64_t nrows,
                                                              int rows_per_mat, const int32_t* __restrict__ mat_dim, int64_t ld, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nrows) return;
  const int qi = (int)(row % rows_per_mat), n = mat_dim[row / rows_per_mat];
  if (qi >= n) return;
  const T* p = P + row * ld;
  const float* dp = dP + row * ld;
  float dot = 0.f;
  for (int c = lane; c < n; c += 64) dot += to_f32(p[c]) * dp[c];
  dot = wave_sum(dot);
  for (int c = lane; c < ld; c += 64) dS[row * ld + c] = from_f32<T>(c < n ? scale * to_f32(p[c]) * (dp[c] - dot) : 0.f);
}

// ----------------------------------------------------------------------------------------------
// KD losses with gradients.  One block per row of fp32 logits.
//   ce:      loss += coef * (lse(s) - s[label]);           d s (+)= coef * (softmax(s) - onehot)
//   soft-ce: loss += coef * (lse(s) - sum softmax(t) s);   d s (+)= coef * (softmax(s) - softmax(t))
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
  v = is_max ? wave_max(v) : wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float r = sh[0];
  for (int w = 1; w < nw; ++w) r = is_max ? fmaxf(r, sh[w]) : r + sh[w];
  return r;
}

template <typename T>
__global__ __launch_bounds__(1024) void logit_loss_kernel(const float* __restrict__ s, const float* __restrict__ t, const int32_t* __restrict__ labels,
                                                          int V, float coef, float* __restrict__ loss, T* __restrict__ ds, int accumulate) {
  __shared__ float sh[16];
  const int64_t row = blockIdx.x;
  const float* sr = s + row * V;
  const float* tr = t ? t + row * V : nullptr;
  float ms = -INFINITY, mt = -INFINITY;
  for (int i = threadIdx.x; i < V; i += blockDim.x) { ms = fmaxf(ms, sr[i]); if (tr) mt = fmaxf(mt, tr[i]); }
  ms = block_reduce(ms, true, sh);
  if (tr) mt = block_reduce(mt, true, sh);
  float ls = 0.f, lt = 0.f, cross = 0.f;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    ls += __expf(sr[i] - ms);
    if (tr) { const float e = __expf(tr[i] - mt); lt += e; cross += e * sr[i]; }
  }
  ls = block_reduce(ls, false, sh);
  if (tr) { lt = block_reduce(lt, false, sh); cross = block_reduce(cross, false, sh); }
  const float lse = ms + logf(ls);
  const int lab = labels ? labels[row] : -1;
  if (threadIdx.x == 0) atomicAdd(loss, coef * (tr ? (lse - cross / lt) : (lse - sr[lab])));
  if (!ds) return;
  T* dr = ds + row * V;
  const float inv_s = 1.0f / ls, inv_t = tr ? 1.0f / lt : 0.f;
  for (int i = threadIdx.x; i < V; i += blockDim.x) {
    float gval = __expf(sr[i] - ms) * inv_s - (tr ? __expf(tr[i] - mt) * inv_t : (i == lab ? 1.0f : 0.f));
    gval *= coef;
    if (accumulate) gval += to_f32(dr[i]);
    dr[i] = from_f32<T>(gval);
  }
}

// All logit losses of one accumulation window in ONE launch (KD step, ref:trainer.py:325-354): a block per row of the packed
// student / teacher tail logits.  labels[row] >= 0: next-token CE term; teacher != NULL and coef[row][2..3] != 0: soft-CE term.
//   losses[slot[row]][0] += c0 * (lse(s) - s[label]);   losses[slot[row]][1] += c2 * (lse(s) - sum softmax(t) s)
//   ds[row] = c1 * (softmax(s) - onehot(label)) + c3 * (softmax(s) - softmax(t))       (written once, no read-modify-write)
// Same arithmetic as logit_loss_kernel; rows are read as 16-byte vectors, the maxima of s and t in one pass.
template <typename T>
__global__ __launch_bounds__(1024) void kd_logit_losses_kernel(const float* __restrict__ s, const float* __restrict__ t, const int32_t* __restrict__ labels,
                                                               const float* __restrict__ coef, const int32_t* __restrict__ slot, int V,
                                                               float* __restrict__ losses, int loss_ld, T* __restrict__ ds) {
  __shared__ float sh[16];
  const int64_t row = blockIdx.x;
  const float* sr = s + row * V;
  const float c0 = coef[row * 4], c1 = coef[row * 4 + 1], c2 = coef[row * 4 + 2], c3 = coef[row * 4 + 3];
  const bool soft = t != nullptr && (c2 != 0.f || c3 != 0.f);
  const float* tr = soft ? t + row * V : nullptr;
  const int lab = labels[row];
  if (V & 3) {   // rows are not 16-byte aligned: scalar form (same arithmetic)
    const float* tr1 = tr;
    float ms = -INFINITY, mt = -INFINITY;
    for (int i = threadIdx.x; i < V; i += 1024) { ms = fmaxf(ms, sr[i]); if (soft) mt = fmaxf(mt, tr1[i]); }
    ms = block_reduce(ms, true, sh);
    if (soft) mt = block_reduce(mt, true, sh);
    float ls = 0.f, lt = 0.f, cross = 0.f;
    for (int i = threadIdx.x; i < V; i += 1024) {
      ls += __expf(sr[i] - ms);
      if (soft) { const float e = __expf(tr1[i] - mt); lt += e; cross += e * sr[i]; }
    }
    ls = block_reduce(ls, false, sh);
    if (soft) { lt = block_reduce(lt, false, sh); cross = block_reduce(cross, false, sh); }
    const float lse = ms + logf(ls);
    if (threadIdx.x == 0) {
      float* lrow = losses + (int64_t)slot[row] * loss_ld;
      if (lab >= 0 && c0 != 0.f) atomicAdd(lrow, c0 * (lse - sr[lab]));
      if (soft && c2 != 0.f) atomicAdd(lrow + 1, c2 * (lse - cross / lt));
    }
    if (!ds) return;
    T* dr = ds + row * V;
    const float inv_s = 1.0f / ls, inv_t = soft ? 1.0f / lt : 0.f;
    const float cce = lab >= 0 ? c1 : 0.f;
    const float cs = cce + (soft ? c3 : 0.f);
    for (int i = threadIdx.x; i < V; i += 1024) {
      float g = cs * (__expf(sr[i] - ms) * inv_s);
      if (soft) g -= c3 * (__expf(tr1[i] - mt) * inv_t);
      if (i == lab) g -= cce;
      dr[i] = from_f32<T>(g);
    }
    return;
  }
  const int n4 = V >> 2;
  const f32x4* s4 = (const f32x4*)sr;
  const f32x4* t4 = (const f32x4*)tr;
  float ms = -INFINITY, mt = -INFINITY;
  for (int i = threadIdx.x; i < n4; i += 1024) {
    const f32x4 a = s4[i];
    ms = fmaxf(fmaxf(ms, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
    if (soft) { const f32x4 b = t4[i]; mt = fmaxf(fmaxf(mt, fmaxf(b[0], b[1])), fmaxf(b[2], b[3])); }
  }
  ms = block_reduce(ms, true, sh);
  if (soft) mt = block_reduce(mt, true, sh);
  float ls = 0.f, lt = 0.f, cross = 0.f;
  for (int i = threadIdx.x; i < n4; i += 1024) {
    const f32x4 a = s4[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) ls += __expf(a[j] - ms);
    if (soft) {
      const f32x4 b = t4[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float e = __expf(b[j] - mt); lt += e; cross += e * a[j]; }
    }
  }
  ls = block_reduce(ls, false, sh);
  if (soft) { lt = block_reduce(lt, false, sh); cross = block_reduce(cross, false, sh); }
  const float lse = ms + logf(ls);
  if (threadIdx.x == 0) {
    float* lrow = losses + (int64_t)slot[row] * loss_ld;
    if (lab >= 0 && c0 != 0.f) atomicAdd(lrow, c0 * (lse - sr[lab]));
    if (soft && c2 != 0.f) atomicAdd(lrow + 1, c2 * (lse - cross / lt));
  }
  if (!ds) return;
  T* dr = ds + row * V;
  const float inv_s = 1.0f / ls, inv_t = soft ? 1.0f / lt : 0.f;
  const float cce = lab >= 0 ? c1 : 0.f;
  const float cs = cce + (soft ? c3 : 0.f);          // weight of softmax(s)
  for (int i = threadIdx.x; i < n4; i += 1024) {
    const f32x4 a = s4[i];
    float g[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) g[j] = cs * (__expf(a[j] - ms) * inv_s);
    if (soft) {
      const f32x4 b = t4[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] -= c3 * (__expf(b[j] - mt) * inv_t);
    }
    if (lab >= 0 && (lab >> 2) == i) g[lab & 3] -= cce;
    if constexpr (sizeof(T) == 2) *(uint2*)(dr + 4 * i) = make_uint2(pack2_bf16(g[0], g[1]), pack2_bf16(g[2], g[3]));
    else *(f32x4*)(dr + 4 * i) = f32x4{g[0], g[1], g[2], g[3]};
  }
}

// Feature-distillation MSE of one tap for every utterance of the window in ONE launch (ref:trainer.py:358-370): a wave per row
// of the packed tail hidden states; losses[slot[row]][2] += c0 * sum_h d^2;  da[row] = c1 * d   (c0 = 1 / (n_u H),
// c1 = 2 w / (n_u H acc) supplied per row by the host).
template <typename T>
__global__ __launch_bounds__(256) void kd_mse_rows_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ coef,
                                                          const int32_t* __restrict__ slot, int64_t rows, int H, float* __restrict__ losses, int loss_ld,
                                                          int loss_col, T* __restrict__ da) {
  constexpr int VEC = Vec16<T>::VEC;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float c0 = coef[row * 2], c1 = coef[row * 2 + 1];
  float acc = 0.f;
  for (int ch = lane; ch < H / VEC; ch += 64) {
    float fa[VEC], fb[VEC];
    Vec16<T>::unpack(*(const uint4*)(a + row * H + ch * VEC), fa);
    Vec16<T>::unpack(*(const uint4*)(b + row * H + ch * VEC), fb);
#pragma unroll
    for (int j = 0; j < VEC; ++j) { fa[j] -= fb[j]; acc += fa[j] * fa[j]; fa[j] *= c1; }
    if (da) *(uint4*)(da + row * H + ch * VEC) = Vec16<T>::pack(fa);
  }
  acc = wave_sum(acc);
  if (lane == 0) atomicAdd(losses + (int64_t)slot[row] * loss_ld + loss_col, c0 * acc);
}

template <typename T>
__global__ __launch_bounds__(256) void mse_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t n, float coef, float* __restrict__ loss,
                                                  T* __restrict__ da, int accumulate) {
  __shared__ float sh[4];
  float acc = 0.f;
  const float gscale = 2.0f * coef / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float d = to_f32(a[i]) - to_f32(b[i]);
    acc += d * d;
    if (da) da[i] = from_f32<T>(gscale * d + (accumulate ? to_f32(da[i]) : 0.f));
  }
  acc = block_reduce(acc, false, sh);
  if (threadIdx.x == 0) atomicAdd(loss, coef * acc / (float)n);
}

// ----------------------------------------------------------------------------------------------
// HuBERT front-end backward pieces
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int64_t T_, int H, int kernel, int stride, int64_t P) {
  constexpr int VEC = Vec16<T>::VEC;
  const int cpr = H / VEC;
  const int64_t total = T_ * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    const int64_t t = i / cpr;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    const int64_t first = t - kernel + 1;  // windows p with p*stride <= t < p*stride + kernel
    const int64_t p_lo = first <= 0 ? 0 : (first + stride - 1) / stride;
    for (int64_t p = p_lo; p < P && p * stride <= t; ++p) {
      float f[VEC];
      Vec16<T>::unpack(*(const uint4*)(dy + p * H + ch * VEC), f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += f[e];
    }
    const float inv = 1.0f / (float)kernel;
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] *= inv;
    *(uint4*)(dx + t * H + ch * VEC) = Vec16<T>::pack(acc);
  }
}

// strided-conv data gradient: dcol (Lout, k*C) -> dx (Lin, C), dx[t][c] = sum_{j, to*s + j == t} dcol[to][j*C + c]
template <typename T>
__global__ __launch_bounds__(256) void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int64_t Lin, int64_t Lout, int Cc, int k, int s) {
  constexpr int VEC = Vec16<T>::VEC;
  const int cpr = Cc / VEC;
  const int64_t total = Lin * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    const int64_t t = i / cpr;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int j = 0; j < k; ++j) {
      const int64_t d = t - j;
      if (d < 0 || d % s) continue;
      const int64_t to = d / s;
      if (to >= Lout) continue;
      float f[VEC];
      Vec16<T>::unpack(*(const uint4*)(dcol + to * (int64_t)k * Cc + (int64_t)j * Cc + ch * VEC), f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += f[e];
    }
    *(uint4*)(dx + t * Cc + ch * VEC) = Vec16<T>::pack(acc);
  }
}

// conv0 backward: recompute conv + LN per time step, push dy through GELU and LN, accumulate fp32 grads
// of (w, bias, gamma, beta).  Same wave-strip geometry as the forward kernel.
// per-wave state of the conv0 backward: taps / LN parameters of the lane's CPL channels and their fp32 gradient sums
template <int CPL, int K>
struct Conv0Acc {
  float wr[CPL][K], br[CPL], gr[CPL], ber[CPL];
  float aw[CPL][K], abias[CPL], ag[CPL], ab[CPL];
  __device__ __forceinline__ void load(const float* w, const float* bias, const float* gamma, const float* beta, int lane) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int ch = lane * CPL + c;
#pragma unroll
      for (int j = 0; j < K; ++j) { wr[c][j] = w[ch * K + j]; aw[c][j] = 0.f; }
      br[c] = bias[ch]; gr[c] = gamma[ch]; ber[c] = beta[ch];
      abias[c] = 0.f; ag[c] = 0.f; ab[c] = 0.f;
    }
  }
  static constexpr int NV = CPL * (K + 3);          // gradient sums per lane
  template <typename F>
  __device__ __forceinline__ void each(F&& f) {     // every sum with its (compile-time) index
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
#pragma unroll
      for (int j = 0; j < K; ++j) f(aw[c][j], c * (K + 3) + j);
      f(abias[c], c * (K + 3) + K);
      f(ag[c], c * (K + 3) + K + 1);
      f(ab[c], c * (K + 3) + K + 2);
    }
  }
  // the four waves of a block folded into wave 0 through LDS (red: 2 x NV x 64 floats) in two rounds — (1 -> 0, 3 -> 2), then 2 -> 0 — so that a
  // block ends in ONE set of atomics instead of four; every wave of the block must call it
  __device__ __forceinline__ void fold_block(float* red, int wave, int lane) {
    if (wave & 1) each([&](float& v, int i) { red[((wave >> 1) * NV + i) * 64 + lane] = v; });
    __syncthreads();
    if (!(wave & 1)) each([&](float& v, int i) { v += red[((wave >> 1) * NV + i) * 64 + lane]; });
    __syncthreads();
    if (wave == 2) each([&](float& v, int i) { red[i * 64 + lane] = v; });
    __syncthreads();
    if (wave == 0) each([&](float& v, int i) { v += red[i * 64 + lane]; });
  }
  __device__ __forceinline__ void flush(float* dw, float* dbias, float* dgamma, float* dbeta, int lane) {
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const int ch = lane * CPL + c;
#pragma unroll
      for (int j = 0; j < K; ++j) atomicAdd(dw + ch * K + j, aw[c][j]);
      atomicAdd(dbias + ch, abias[c]);
      atomicAdd(dgamma + ch, ag[c]);
      atomicAdd(dbeta + ch, ab[c]);
    }
  }
};

// one strip of TS time steps of one utterance: recompute conv + LN, push dy through GELU and LN, add into the wave's sums
template <typename T, int CPL, int K, int STRIDE>
__device__ __forceinline__ void conv0_bwd_strip(Conv0Acc<CPL, K>& A, const float* __restrict__ wave, int64_t n_samples, const T* __restrict__ dy,
                                                int64_t L, float eps, int64_t strip, int lane) {
  constexpr int C = 64 * CPL;
  constexpr int TS = (128 - (K - STRIDE)) / STRIDE;
  const int64_t t0 = strip * TS;
  if (t0 >= L) return;
  const int64_t s0 = t0 * STRIDE;
  const int64_t i0 = s0 + lane, i1 = s0 + 64 + lane;
  const int b0 = __builtin_bit_cast(int, i0 < n_samples ? wave[i0] : 0.f), b1 = __builtin_bit_cast(int, i1 < n_samples ? wave[i1] : 0.f);
  const int nt = (int)((L - t0) < TS ? (L - t0) : TS);
  for (int tt = 0; tt < nt; ++tt) {
    float xs[K];
#pragma unroll
    for (int j = 0; j < K; ++j) {
      const int idx = tt * STRIDE + j;
      const int lo = __builtin_amdgcn_readlane(b0, idx & 63), hi = __builtin_amdgcn_readlane(b1, idx & 63);
      xs[j] = __builtin_bit_cast(float, idx < 64 ? lo : hi);
    }
    float y[CPL], s = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      float a = A.br[c];
#pragma unroll
      for (int j = 0; j < K; ++j) a = fmaf(A.wr[c][j], xs[j], a);
      y[c] = a; s += a;
    }
    const float mean = wave_sum(s) / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) { const float d = y[c] - mean; s2 += d * d; }
    const float rstd = rsqrtf(wave_sum(s2) / (float)C + eps);
    const T* drow = dy + (t0 + tt) * C + lane * CPL;
    float dg[CPL], xh[CPL], r1 = 0.f, r2 = 0.f;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      xh[c] = (y[c] - mean) * rstd;
      const float d = to_f32(drow[c]) * gelu_grad(xh[c] * A.gr[c] + A.ber[c]);
      A.ag[c] += d * xh[c];
      A.ab[c] += d;
      dg[c] = d * A.gr[c];
      r1 += dg[c];
      r2 += dg[c] * xh[c];
    }
    r1 = wave_sum(r1) / (float)C;
    r2 = wave_sum(r2) / (float)C;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
      const float dc = rstd * (dg[c] - r1 - xh[c] * r2);
      A.abias[c] += dc;
#pragma unroll
      for (int j = 0; j < K; ++j) A.aw[c][j] = fmaf(dc, xs[j], A.aw[c][j]);
    }
  }
}

template <typename T, int CPL, int K, int STRIDE>
__global__ __launch_bounds__(256) void conv0_bwd_kernel(const float* __restrict__ wave, int64_t n_samples, const float* __restrict__ w,
                                                        const float* __restrict__ bias, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const T* __restrict__ dy, int64_t L, float eps, float* __restrict__ dw, float* __restrict__ dbias,
                                                        float* __restrict__ dgamma, float* __restrict__ dbeta) {
  constexpr int TS = (128 - (K - STRIDE)) / STRIDE;
  const int lane = threadIdx.x & 63;
  const int64_t strip = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (strip * TS >= L) return;
  Conv0Acc<CPL, K> A;
  A.load(w, bias, gamma, beta, lane);
  conv0_bwd_strip<T, CPL, K, STRIDE>(A, wave, n_samples, dy, L, eps, strip, lane);
  A.flush(dw, dbias, dgamma, dbeta, lane);
}

// The whole ragged batch with a fixed number of waves: a wave walks (utterance, strip) pairs spref[u] + strip with a
// stride of the grid; the four waves of a block fold their sums through LDS and the block flushes ONCE.  One launch per
// utterance had every wave end in 104 atomic instructions on the same 6.6 k addresses after only 24 time steps (0.9 ms per
// 10 s clip, almost all of it atomic traffic); a flush per WAVE of a 512-block grid was still 13.6 M atomics = ~1 ms whatever
// the batch (1.08 ms for 2 utterances, 1.85 ms for 16: profiles/r06_kd_window{2,16}_ops_after.txt) — the launcher now also
// gives every wave >= 4 strips before it widens the grid.
template <typename T, int CPL, int K, int STRIDE>
__global__ __launch_bounds__(256) void conv0_bwd_batch_kernel(const float* __restrict__ waves, const int64_t* __restrict__ soff,
                                                              const int64_t* __restrict__ row0, const int64_t* __restrict__ spref, int n_utt,
                                                              const float* __restrict__ w, const float* __restrict__ bias,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, const T* __restrict__ dy,
                                                              float eps, float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int fold) {
  __shared__ float red[2 * Conv0Acc<CPL, K>::NV * 64];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t wave_id = (int64_t)blockIdx.x * 4 + wave, n_waves = (int64_t)gridDim.x * 4;
  const int64_t total = spref[n_utt];
  Conv0Acc<CPL, K> A;
  A.load(w, bias, gamma, beta, lane);
  for (int64_t gs = wave_id; gs < total; gs += n_waves) {
    int lo = 0, hi = n_utt - 1;                      // utterance u with spref[u] <= gs < spref[u + 1]
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (spref[mid] <= gs) lo = mid; else hi = mid - 1;
    }
    const int64_t s0 = soff[lo], r0 = row0[lo];
    conv0_bwd_strip<T, CPL, K, STRIDE>(A, waves + s0, soff[lo + 1] - s0, dy + r0 * (64 * CPL), row0[lo + 1] - r0, eps, gs - spref[lo], lane);
  }
  if (fold) A.fold_block(red, wave, lane);                   // (uniform over the grid)
  if (wave == 0 || (!fold && wave_id < total)) A.flush(dw, dbias, dgamma, dbeta, lane);
}

// ----------------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------------
static inline unsigned grid_for(int64_t items, int cap = 8192) {
  const int64_t g = ceil_div64(items, 256);
  return (unsigned)(g < 1 ? 1 : (g < cap ? g : cap));
}

extern "C" int sl_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(dy && pre && dx && n >= 0, "sl_gelu_bwd: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(n % vec == 0, "sl_gelu_bwd: n must be a multiple of %d", vec);
  if (n == 0) return 0;
  SL_DISPATCH_DTYPE(dtype, T, { hipLaunchKernelGGL((gelu_bwd_kernel<T>), dim3(grid_for(n / vec)), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (const T*)pre, (T*)dx, n); });
  SL_CHECK_LAUNCH("gelu_bwd");
  return 0;
}

extern "C" int sl_axpby(const void* x, void* y, float a, float b, int64_t n, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(x && y && n >= 0, "sl_axpby: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(n % vec == 0, "sl_axpby: n must be a multiple of %d", vec);
  if (n == 0) return 0;
  SL_DISPATCH_DTYPE(dtype, T, { hipLaunchKernelGGL((axpby_kernel<T>), dim3(grid_for(n / vec)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, a, b, n); });
  SL_CHECK_LAUNCH("axpby");
  return 0;
}

extern "C" int sl_dropout(const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(x && y && n >= 0 && p >= 0.f && p < 1.f, "sl_dropout: bad arguments (p=%f)", (double)p);
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(n % vec == 0, "sl_dropout: n must be a multiple of %d", vec);
  if (n == 0) return 0;
  const uint32_t thr24 = (uint32_t)((double)p * 16777216.0);
  const float scale = 1.0f / (1.0f - p);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((dropout_kernel<T>), dim3(grid_for(n / vec)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)residual, (T*)y, n, scale,
                       thr24, (uint64_t)seed);
  });
  SL_CHECK_LAUNCH("dropout");
  return 0;
}

extern "C" int sl_attn_dropout_bwd(const void* p, void* p_dropped, float* d_p, int64_t n_mat, int32_t smax, const int32_t* dims, int32_t ld,
                                   const int32_t* cu_q, int32_t n_heads, int32_t n_kv_heads, int32_t r, float dropout_p, uint64_t seed, int32_t dtype,
                                   sl_stream stream) {
  SL_CHECK_ARG(p && p_dropped && d_p && dims && cu_q && n_mat > 0 && smax > 0 && smax <= 65536 && ld >= smax && n_heads > 0 && n_kv_heads > 0 &&
                   n_heads % n_kv_heads == 0 && r >= 0 && r < n_heads / n_kv_heads && dropout_p >= 0.f && dropout_p < 1.f,
               "sl_attn_dropout_bwd: bad arguments");
  const uint32_t thr24 = (uint32_t)((double)dropout_p * 16777216.0);
  const float scale = 1.0f / (1.0f - dropout_p);
  const int64_t cells = (int64_t)smax * smax;
  const unsigned gx = (unsigned)(ceil_div64(cells, 256) < 256 ? ceil_div64(cells, 256) : 256);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((attn_dropout_bwd_kernel<T>), dim3(gx, (unsigned)n_mat), dim3(256), 0, (hipStream_t)stream, (const T*)p, (T*)p_dropped, d_p, smax,
                       dims, ld, cu_q, n_heads, n_kv_heads, r, scale, thr24, (uint64_t)seed);
  });
  SL_CHECK_LAUNCH("attn_dropout_bwd");
  return 0;
}

// internal (train_tape.hip): every record checked like sl_transpose_pad's arguments; n may exceed SL_TRANSPOSE_BATCH (several launches)
int sl_transpose_pad_batch_impl(const SlTransposeRec* recs, int n, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(recs && n > 0, "sl_transpose_pad_batch: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  for (int i0 = 0; i0 < n; i0 += SL_TRANSPOSE_BATCH) {
    SlTransposeBatch b;
    memset(&b, 0, sizeof(b));
    b.n = n - i0 < SL_TRANSPOSE_BATCH ? n - i0 : SL_TRANSPOSE_BATCH;
    int blocks = 0;
    for (int i = 0; i < b.n; ++i) {
      SlTransposeRec q = recs[i0 + i];
      SL_CHECK_ARG(q.x && q.y && q.rows > 0 && q.cols > 0 && q.ld_out >= q.rows && q.ldy >= q.ld_out && q.ldx >= q.cols, "sl_transpose_pad_batch: bad shape (record %d)", i0 + i);
      SL_CHECK_ARG(q.ldx % vec == 0 && q.ldy % vec == 0 && ((uintptr_t)q.x & 15) == 0 && ((uintptr_t)q.y & 15) == 0, "sl_transpose_pad_batch: rows must stay 16-byte aligned");
      q.tiles_r = (int)ceil_div64(q.ld_out, 64);
      b.rec[i] = q;
      b.first_block[i] = blocks;
      blocks += q.tiles_r * (int)ceil_div64(q.cols, 64);
    }
    for (int i = b.n; i < SL_TRANSPOSE_BATCH; ++i) b.first_block[i] = blocks;
    SL_DISPATCH_DTYPE(dtype, T, { hipLaunchKernelGGL((transpose_pad_batch_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, b); });
    SL_CHECK_LAUNCH("transpose_pad_batch");
  }
  return 0;
}

extern "C" int sl_transpose_pad(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, int32_t ld_out, int32_t dtype,
                                sl_stream stream) {
  SL_CHECK_ARG(x && y && rows > 0 && cols > 0 && ld_out >= rows && ldy >= ld_out && ldx >= cols, "sl_transpose_pad: bad shape");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(ldx % vec == 0 && ldy % vec == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0, "sl_transpose_pad: rows must stay 16-byte aligned");
  dim3 grid((ld_out + 63) / 64, (cols + 63) / 64);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((transpose_pad_kernel<T>), grid, dim3(256), 0, (hipStream_t)stream, (const T*)x, ldx, (T*)y, ldy, rows, cols, ld_out);
  });
  SL_CHECK_LAUNCH("transpose_pad");
  return 0;
}

extern "C" int sl_silu_mul(const void* gu, void* out, int64_t M, int32_t F_, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(gu && out && M >= 0 && F_ > 0 && F_ % 16 == 0, "sl_silu_mul: bad arguments");
  if (M == 0) return 0;
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_DISPATCH_DTYPE(dtype, T, { hipLaunchKernelGGL((silu_mul_kernel<T, false>), dim3(grid_for(M * (F_ / vec))), dim3(256), 0, (hipStream_t)stream, (const T*)gu, (const T*)nullptr, (T*)out, M, F_); });
  SL_CHECK_LAUNCH("silu_mul");
  return 0;
}

extern "C" int sl_silu_mul_bwd(const void* gu, const void* dy, void* dgu, int64_t M, int32_t F_, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(gu && dy && dgu && M >= 0 && F_ > 0 && F_ % 16 == 0, "sl_silu_mul_bwd: bad arguments");
  if (M == 0) return 0;
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_DISPATCH_DTYPE(dtype, T, { hipLaunchKernelGGL((silu_mul_kernel<T, true>), dim3(grid_for(M * (F_ / vec))), dim3(256), 0, (hipStream_t)stream, (const T*)gu, (const T*)dy, (T*)dgu, M, F_); });
  SL_CHECK_LAUNCH("silu_mul_bwd");
  return 0;
}

extern "C" int sl_rope_inplace(void* x, const int32_t* tok_pos, const float* cos, const float* sin, int64_t n_tok, int32_t heads, int32_t n_rot,
                               int32_t D, int32_t inverse, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(x && tok_pos && cos && sin && n_rot <= heads, "sl_rope_inplace: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(D % (2 * vec) == 0, "sl_rope_inplace: head_dim=%d must be a multiple of %d", D, 2 * vec);
  if (n_tok == 0 || n_rot == 0) return 0;
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((rope_inplace_kernel<T>), dim3(grid_for(n_tok * n_rot * (D / 2 / vec))), dim3(256), 0, (hipStream_t)stream, (T*)x, tok_pos, cos, sin,
                       n_tok, heads, n_rot, D, inverse ? -1.0f : 1.0f);
  });
  SL_CHECK_LAUNCH("rope_inplace");
  return 0;
}

// blocks of the scratch-buffer form: 16 rows in flight per block (16 waves), ~one block per CU (a block pays for its gain / bias
// loads and the LDS fold of its partials whatever its share of the rows: 500 blocks of 16 rows took 47 us on 7 984 x 1 024)
static int ln_bwd_nw() { const int n = sl_env().lnbwd_nw; return n == 4 || n == 8 ? n : 16; }
static int ln_bwd_ws_blocks(int64_t rows, int* rpb_out) {
  const int nw = ln_bwd_nw();
  int rpb = (int)ceil_div64(ceil_div64(rows, 256 * (16 / nw)), nw) * nw;      // the same rows in flight per CU whatever the block size
  rpb = rpb < nw ? nw : rpb;
  if (rpb_out) *rpb_out = rpb;
  return (int)ceil_div64(rows, rpb);
}

extern "C" size_t sl_layernorm_bwd_ws_bytes(int64_t rows, int32_t cols) {
  if (rows <= 0 || cols <= 0 || cols > 64 * 16) return 0;     // wider rows take the general (atomics) form
  // sized for the smallest block form (SL_LNBWD_NW=4: four times the blocks), so that the tuning switch can change under a live workspace
  const int64_t rpb4 = ceil_div64(ceil_div64(rows, 1024), 4) * 4;
  const size_t nb_max = (size_t)ceil_div64(rows, rpb4 < 4 ? 4 : rpb4);
  const size_t nb = (size_t)ln_bwd_ws_blocks(rows, nullptr);
  return (nb_max > nb ? nb_max : nb) * 2 * (size_t)cols * sizeof(float) + 256;      // + the arrival counter of the in-kernel column reduce (its last 256 bytes)
}

static int layernorm_bwd_impl(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, float* dgamma, float* dbeta, int64_t rows,
                              int32_t cols, float eps, int32_t gelu, int32_t dtype, void* ws, size_t ws_bytes, sl_stream stream, const void* add = nullptr,
                              NormBwdX ex = NormBwdX{}, int32_t* colred_cnt = nullptr) {
  SL_CHECK_ARG(!add || add != dx, "sl_layernorm_bwd: the residual gradient must not alias dx");
  SL_CHECK_ARG(!ex.dx_drop || (ex.dx_drop != dx && ex.dx_drop != dy && ex.dx_drop != add), "sl_layernorm_bwd: the dropped copy must not alias dx / dy / add");
  SL_CHECK_ARG(x && gamma && beta && dy && dx && rows >= 0 && cols > 0, "sl_layernorm_bwd: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(cols % vec == 0 && cols <= 64 * TR_MAXF, "sl_layernorm_bwd: cols=%d must be a multiple of %d and <= %d", cols, vec, 64 * TR_MAXF);
  if (rows == 0) return 0;
  if (cols <= 64 * 16) {      // <= 1 024 elements per row: 16 floats per lane, 16 waves per block
    int rpb = 16;
    const int nb = ln_bwd_ws_blocks(rows, &rpb);
    if (ws && dgamma && dbeta && ws_bytes >= (size_t)nb * 2 * (size_t)cols * sizeof(float)) {
      const bool in_kernel = colred_cnt && !gelu && ln_bwd_nw() == 16 && sl_env().ln_colred_inkernel;      // the 16-wave LEAN form carries the last-block reduce
      ex.colred_cnt = in_kernel ? colred_cnt : nullptr;
      SL_DISPATCH_DTYPE(dtype, T, {
        if (gelu)
          hipLaunchKernelGGL((norm_bwd_kernel<T, false, 16, 16>), dim3((unsigned)nb), dim3(1024), 0, (hipStream_t)stream, (const T*)x, (const T*)gamma,
                             (const T*)beta, (const T*)dy, (T*)dx, dgamma, dbeta, rows, cols, eps, gelu, rpb, (float*)ws, (const T*)add, ex);
        else if (ln_bwd_nw() == 8)
          hipLaunchKernelGGL((norm_bwd_kernel<T, false, 16, 8, true>), dim3((unsigned)nb), dim3(512), 0, (hipStream_t)stream, (const T*)x, (const T*)gamma,
                             (const T*)beta, (const T*)dy, (T*)dx, dgamma, dbeta, rows, cols, eps, 0, rpb, (float*)ws, (const T*)add, ex);
        else if (ln_bwd_nw() == 4)
          hipLaunchKernelGGL((norm_bwd_kernel<T, false, 16, 4, true>), dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)gamma,
                             (const T*)beta, (const T*)dy, (T*)dx, dgamma, dbeta, rows, cols, eps, 0, rpb, (float*)ws, (const T*)add, ex);
        else
          hipLaunchKernelGGL((norm_bwd_kernel<T, false, 16, 16, true>), dim3((unsigned)nb), dim3(1024), 0, (hipStream_t)stream, (const T*)x, (const T*)gamma,
                             (const T*)beta, (const T*)dy, (T*)dx, dgamma, dbeta, rows, cols, eps, 0, rpb, (float*)ws, (const T*)add, ex);
      });
      SL_CHECK_LAUNCH("layernorm_bwd");
      if (in_kernel) return 0;
      hipLaunchKernelGGL(norm_colreduce_kernel, dim3((unsigned)((2 * cols + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, (const float*)ws, nb, cols, dgamma,
                         dbeta);
      SL_CHECK_LAUNCH("layernorm_bwd(colreduce)");
      return 0;
    }
    // no scratch: atomics, so few blocks (measured, tools/time_lnbwd.py: 7 984 x 1 024 takes 79 / 53 / 46 us at 256 / 128 / 64 blocks,
    // 255 984 x 512 403 / 666 / 1 287 us)
    int64_t lnb = rows / 64;
    lnb = lnb < 64 ? 64 : (lnb > 256 ? 256 : lnb);
    int rpb16 = (int)ceil_div64(ceil_div64(rows, lnb), 16) * 16;
    rpb16 = rpb16 < 16 ? 16 : rpb16;
    SL_DISPATCH_DTYPE(dtype, T, {
      if (gelu)
        hipLaunchKernelGGL((norm_bwd_kernel<T, false, 16, 16>), dim3((unsigned)ceil_div64(rows, rpb16)), dim3(1024), 0, (hipStream_t)stream, (const T*)x,
                           (const T*)gamma, (const T*)beta, (const T*)dy, (T*)dx, dgamma, dbeta, rows, cols, eps, gelu, rpb16, (float*)nullptr, (const T*)add, ex);
      else
        hipLaunchKernelGGL((norm_bwd_kernel<T, false, 16, 16, true>), dim3((unsigned)ceil_div64(rows, rpb16)), dim3(1024), 0, (hipStream_t)stream, (const T*)x,
                           (const T*)gamma, (const T*)beta, (const T*)dy, (T*)dx, dgamma, dbeta, rows, cols, eps, 0, rpb16, (float*)nullptr, (const T*)add, ex);
    });
    SL_CHECK_LAUNCH("layernorm_bwd");
    return 0;
  }
  // rows per block: 64 rows per block left a 7 984-row encoder LayerNorm on 125 blocks (115 us for 48 MB); ~1 000 blocks were
  // WORSE (254 us): the per-block column atomics all land on the same 2 x cols addresses and serialise at the memory side
  int rpb = (int)ceil_div64(ceil_div64(rows, 256), 4) * 4;   // ~one block per CU; more blocks = more atomics on the same 2 x cols addresses
  rpb = rpb < 4 ? 4 : (rpb > 64 ? 64 : rpb);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((norm_bwd_kernel<T, false, TR_MAXF, 4>), dim3((unsigned)ceil_div64(rows, rpb)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)gamma,
                       (const T*)beta, (const T*)dy, (T*)dx, dgamma, dbeta, rows, cols, eps, gelu, rpb, (float*)nullptr, (const T*)add, ex);
  });
  SL_CHECK_LAUNCH("layernorm_bwd");
  return 0;
}

// the training tapes' forms (train_tape.hip): dx = backward(dy) + add
// dx_drop (optional): also dropout(dx) with (p, seed) — the incoming gradient of the Linear below (h = h + dropout(sublayer(h)))
int sl_layernorm_bwd_ws_add_impl(const void* x, const void* gamma, const void* beta, const void* dy, const void* add, void* dx, float* dgamma, float* dbeta,
                                 int64_t rows, int32_t cols, float eps, int32_t dtype, void* workspace, size_t workspace_bytes, sl_stream stream,
                                 void* dx_drop, float drop_p, uint64_t drop_seed, const float* dy_parts, int dy_splits, int32_t* colred_cnt) {
  NormBwdX ex{};
  if (dy_parts && dy_splits >= 2) { ex.dy_parts = dy_parts; ex.dy_S = dy_splits; ex.dy_slab = rows * (int64_t)cols; }
  if (dx_drop && drop_p > 0.f) {
    ex.dx_drop = dx_drop; ex.thr24 = (uint32_t)((double)drop_p * 16777216.0); ex.scale = 1.0f / (1.0f - drop_p); ex.seed = drop_seed;
  }
  return layernorm_bwd_impl(x, gamma, beta, dy, dx, dgamma, dbeta, rows, cols, eps, 0, dtype, workspace, workspace_bytes, stream, add, ex, colred_cnt);
}

extern "C" int sl_layernorm_bwd(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, float* dgamma, float* dbeta,
                                int64_t rows, int32_t cols, float eps, int32_t gelu, int32_t dtype, sl_stream stream) {
  return layernorm_bwd_impl(x, gamma, beta, dy, dx, dgamma, dbeta, rows, cols, eps, gelu, dtype, nullptr, 0, stream);
}

extern "C" int sl_layernorm_bwd_ws(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, float* dgamma, float* dbeta,
                                   int64_t rows, int32_t cols, float eps, int32_t gelu, int32_t dtype, void* workspace, size_t workspace_bytes,
                                   sl_stream stream) {
  return layernorm_bwd_impl(x, gamma, beta, dy, dx, dgamma, dbeta, rows, cols, eps, gelu, dtype, workspace, workspace_bytes, stream);
}

// add2 (optional): a second gradient joining at this hidden state (the feature-distillation term), added after `add` with the roundings of a
// following sl_axpby(add2, dx)
int sl_rmsnorm_bwd_add_impl(const void* x, const void* w, const void* dy, const void* add, void* dx, int64_t rows, int32_t cols, float eps, int32_t dtype,
                            sl_stream stream, const void* add2, const float* dy_parts, int dy_splits) {
  SL_CHECK_ARG(x && w && dy && dx && rows >= 0 && cols > 0 && (!add || add != dx) && (!add2 || add2 != dx), "sl_rmsnorm_bwd: bad arguments");
  NormBwdX ex{};
  ex.add2 = add2;
  if (dy_parts && dy_splits >= 2) { ex.dy_parts = dy_parts; ex.dy_S = dy_splits; ex.dy_slab = rows * (int64_t)cols; }
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(cols % vec == 0 && cols <= 64 * TR_MAXF, "sl_rmsnorm_bwd: cols=%d must be a multiple of %d and <= %d", cols, vec, 64 * TR_MAXF);
  if (rows == 0) return 0;
  // rows per block (4 waves, a wave per row at a time): 16 keeps ~300 blocks at the 5 072 rows of a 16-sample window; the per-rank window's 634 rows
  // would make 40 blocks of four sequential rows per wave (31 us for 4 MB) — one row per wave there: 159 blocks
  // The lean form (x / dy kept as loaded, converted again in every pass — the same operations in the same order, the same bits) needs ~130 registers
  // where the float copies need 256: three waves per SIMD instead of one, so the passes of different rows overlap.  With that many resident
  // waves the rows are spread thin (rpb 4) at every row count.
  const int lean = sl_env().rms_bwd_lean;
  const int rpb = rows <= 2048 || lean == 2 ? 4 : (lean == 3 ? 8 : 16);
  SL_DISPATCH_DTYPE(dtype, T, {
    if (lean && cols <= 64 * 48)           // (Llama-3.2-3B: 3 072 = 48 floats per lane — arrays sized for it, not for the 4 096 the general form admits)
      hipLaunchKernelGGL((norm_bwd_kernel<T, true, 48, 4, true>), dim3((unsigned)ceil_div64(rows, rpb)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)w,
                         (const T*)nullptr, (const T*)dy, (T*)dx, (float*)nullptr, (float*)nullptr, rows, cols, eps, 0, rpb, (float*)nullptr, (const T*)add, ex);
    else if (lean)
      hipLaunchKernelGGL((norm_bwd_kernel<T, true, TR_MAXF, 4, true>), dim3((unsigned)ceil_div64(rows, rpb)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)w,
                         (const T*)nullptr, (const T*)dy, (T*)dx, (float*)nullptr, (float*)nullptr, rows, cols, eps, 0, rpb, (float*)nullptr, (const T*)add, ex);
    else
      hipLaunchKernelGGL((norm_bwd_kernel<T, true, TR_MAXF, 4>), dim3((unsigned)ceil_div64(rows, rpb)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (const T*)w,
                         (const T*)nullptr, (const T*)dy, (T*)dx, (float*)nullptr, (float*)nullptr, rows, cols, eps, 0, rpb, (float*)nullptr, (const T*)add, ex);
  });
  SL_CHECK_LAUNCH("rmsnorm_bwd");
  return 0;
}

extern "C" int sl_rmsnorm_bwd(const void* x, const void* w, const void* dy, void* dx, int64_t rows, int32_t cols, float eps, int32_t dtype,
                              sl_stream stream) {
  return sl_rmsnorm_bwd_add_impl(x, w, dy, nullptr, dx, rows, cols, eps, dtype, stream, nullptr, nullptr, 0);
}

extern "C" int sl_colsum(const void* x, int64_t ld, float* out, int64_t rows, int32_t cols, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(x && out && rows >= 0 && cols > 0, "sl_colsum: bad arguments");
  if (rows == 0) return 0;
  const int vec = dtype == SL_F32 ? 4 : 8;
  if (cols % vec == 0 && ld % vec == 0 && ((uintptr_t)x & 15) == 0) {
    int rpb = (int)ceil_div64(rows, 64);          // <= 64 adders per column
    rpb = rpb < 8 ? 8 : rpb;
    SL_DISPATCH_DTYPE(dtype, T, {
      hipLaunchKernelGGL((colsum_vec_kernel<T>), dim3((unsigned)ceil_div64(rows, rpb), (cols / vec + 31) / 32), dim3(256), 0, (hipStream_t)stream,
                         (const T*)x, ld, out, rows, cols, rpb);
    });
    SL_CHECK_LAUNCH("colsum");
    return 0;
  }
  const int rpb = 128;
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((colsum_kernel<T>), dim3((unsigned)ceil_div64(rows, rpb), (cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const T*)x, ld, out,
                       rows, cols, rpb);
  });
  SL_CHECK_LAUNCH("colsum");
  return 0;
}

extern "C" int sl_softmax_rows(const float* S, void* P, int64_t n_mats, int32_t rows, int32_t cols, int64_t ld, float scale, int32_t causal,
                               int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(S && P && n_mats > 0 && rows > 0 && cols > 0 && ld >= cols, "sl_softmax_rows: bad arguments");
  const int64_t nrows = n_mats * rows;
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((softmax_rows_kernel<T>), dim3((unsigned)ceil_div64(nrows, 4)), dim3(256), 0, (hipStream_t)stream, S, (T*)P, nrows, rows, cols, ld,
                       scale, cols - rows, causal);
  });
  SL_CHECK_LAUNCH("softmax_rows");
  return 0;
}

extern "C" int sl_softmax_bwd(const void* P, const float* dP, void* dS, int64_t nrows, int32_t cols, int64_t ld, float scale, int32_t dtype,
                              sl_stream stream) {
  SL_CHECK_ARG(P && dP && dS && nrows > 0 && cols > 0 && ld >= cols, "sl_softmax_bwd: bad arguments");
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((softmax_bwd_kernel<T>), dim3((unsigned)ceil_div64(nrows, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)P, dP, (T*)dS, nrows, cols,
                       ld, scale);
  });
  SL_CHECK_LAUNCH("softmax_bwd");
  return 0;
}

extern "C" int sl_softmax_rows_var(const float* S, void* P, int64_t n_mats, int32_t rows_per_mat, const int32_t* mat_dim, int64_t ld, float scale,
                                   int32_t causal, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(S && P && mat_dim && n_mats > 0 && rows_per_mat > 0 && ld >= rows_per_mat, "sl_softmax_rows_var: bad arguments");
  const int64_t nrows = n_mats * rows_per_mat;
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((softmax_rows_var_kernel<T>), dim3((unsigned)ceil_div64(nrows, 4)), dim3(256), 0, (hipStream_t)stream, S, (T*)P, nrows,
                       rows_per_mat, mat_dim, ld, scale, causal);
  });
  SL_CHECK_LAUNCH("softmax_rows_var");
  return 0;
}

extern "C" int sl_softmax_bwd_var(const void* P, const float* dP, void* dS, int64_t n_mats, int32_t rows_per_mat, const int32_t* mat_dim,
                                  int64_t ld, float scale, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(P && dP && dS && mat_dim && n_mats > 0 && rows_per_mat > 0 && ld >= rows_per_mat, "sl_softmax_bwd_var: bad arguments");
  const int64_t nrows = n_mats * rows_per_mat;
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((softmax_bwd_var_kernel<T>), dim3((unsigned)ceil_div64(nrows, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)P, dP, (T*)dS,
                       nrows, rows_per_mat, mat_dim, ld, scale);
  });
  SL_CHECK_LAUNCH("softmax_bwd_var");
  return 0;
}

extern "C" int sl_ce_loss(const float* logits, const int32_t* labels, int64_t rows, int32_t V, float coef, float* loss, void* dlogits,
                          int32_t accumulate, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(logits && labels && loss && rows >= 0 && V > 0, "sl_ce_loss: bad arguments");
  if (rows == 0) return 0;
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((logit_loss_kernel<T>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, logits, (const float*)nullptr, labels, V, coef, loss,
                       (T*)dlogits, accumulate);
  });
  SL_CHECK_LAUNCH("ce_loss");
  return 0;
}

extern "C" int sl_soft_ce_loss(const float* student, const float* teacher, int64_t rows, int32_t V, float coef, float* loss, void* dstudent,
                               int32_t accumulate, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(student && teacher && loss && rows >= 0 && V > 0, "sl_soft_ce_loss: bad arguments");
  if (rows == 0) return 0;
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((logit_loss_kernel<T>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, student, teacher, (const int32_t*)nullptr, V, coef,
                       loss, (T*)dstudent, accumulate);
  });
  SL_CHECK_LAUNCH("soft_ce_loss");
  return 0;
}

extern "C" int sl_kd_logit_losses(const float* student, const float* teacher, const int32_t* labels, const float* row_coef, const int32_t* row_slot,
                                  int64_t rows, int32_t V, float* losses, int32_t loss_ld, void* dstudent, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(student && labels && row_coef && row_slot && losses && rows >= 0 && V > 0 && loss_ld >= 2, "sl_kd_logit_losses: bad arguments");
  SL_CHECK_ARG(((uintptr_t)student & 15) == 0 && ((uintptr_t)teacher & 15) == 0 && ((uintptr_t)dstudent & 15) == 0, "sl_kd_logit_losses: 16-byte aligned buffers");
  if (rows == 0) return 0;
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((kd_logit_losses_kernel<T>), dim3((unsigned)rows), dim3(1024), 0, (hipStream_t)stream, student, teacher, labels, row_coef, row_slot, V,
                       losses, loss_ld, (T*)dstudent);
  });
  SL_CHECK_LAUNCH("kd_logit_losses");
  return 0;
}

extern "C" int sl_kd_mse_rows(const void* a, const void* b, const float* row_coef, const int32_t* row_slot, int64_t rows, int32_t H, float* losses,
                              int32_t loss_ld, int32_t loss_col, void* da, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(a && b && row_coef && row_slot && losses && rows >= 0 && H > 0 && loss_col >= 0 && loss_col < loss_ld, "sl_kd_mse_rows: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(H % vec == 0, "sl_kd_mse_rows: H must be a multiple of %d", vec);
  if (rows == 0) return 0;
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((kd_mse_rows_kernel<T>), dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, row_coef, row_slot,
                       rows, H, losses, loss_ld, loss_col, (T*)da);
  });
  SL_CHECK_LAUNCH("kd_mse_rows");
  return 0;
}

extern "C" int sl_mse_loss(const void* a, const void* b, int64_t n, float coef, float* loss, void* da, int32_t accumulate, int32_t dtype,
                           sl_stream stream) {
  SL_CHECK_ARG(a && b && loss && n > 0, "sl_mse_loss: bad arguments");
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((mse_kernel<T>), dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, (const T*)a, (const T*)b, n, coef, loss, (T*)da, accumulate);
  });
  SL_CHECK_LAUNCH("mse_loss");
  return 0;
}

extern "C" int sl_avgpool_bwd(const void* dy, void* dx, int64_t T_, int32_t H, int32_t kernel, int32_t stride, int64_t P, int32_t dtype,
                              sl_stream stream) {
  SL_CHECK_ARG(dy && dx && T_ > 0 && H > 0 && kernel > 0 && stride > 0, "sl_avgpool_bwd: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(H % vec == 0, "sl_avgpool_bwd: H must be a multiple of %d", vec);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((avgpool_bwd_kernel<T>), dim3(grid_for(T_ * (H / vec))), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, T_, H, kernel, stride, P);
  });
  SL_CHECK_LAUNCH("avgpool_bwd");
  return 0;
}

// whole ragged batch in one launch: utterance u = blockIdx.y; desc[u] = {rows in, rows out, first input row, first output row} (int64)
template <typename T>
__global__ __launch_bounds__(256) void col2im_batch_kernel(const T* __restrict__ dcol, T* __restrict__ dx, const int64_t* __restrict__ desc, int Cc, int k, int s) {
  const int64_t* d = desc + 4 * (int64_t)blockIdx.y;
  const int64_t Lout = d[0], Lin = d[1];
  dcol += d[2] * (int64_t)k * Cc;
  dx += d[3] * Cc;
  constexpr int VEC = Vec16<T>::VEC;
  const int cpr = Cc / VEC;
  const int64_t total = Lin * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    const int64_t t = i / cpr;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int j = 0; j < k; ++j) {
      const int64_t dd = t - j;
      if (dd < 0 || dd % s) continue;
      const int64_t to = dd / s;
      if (to >= Lout) continue;
      float f[VEC];
      Vec16<T>::unpack(*(const uint4*)(dcol + to * (int64_t)k * Cc + (int64_t)j * Cc + ch * VEC), f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += f[e];
    }
    *(uint4*)(dx + t * Cc + ch * VEC) = Vec16<T>::pack(acc);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_batch_kernel(const T* __restrict__ dy, T* __restrict__ dx, const int64_t* __restrict__ desc, int H, int kernel,
                                                                int stride) {
  const int64_t* d = desc + 4 * (int64_t)blockIdx.y;
  const int64_t P = d[0], T_ = d[1];
  dy += d[2] * H;
  dx += d[3] * H;
  constexpr int VEC = Vec16<T>::VEC;
  const int cpr = H / VEC;
  const int64_t total = T_ * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ch = (int)(i % cpr);
    const int64_t t = i / cpr;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    const int64_t first = t - kernel + 1;
    const int64_t p_lo = first <= 0 ? 0 : (first + stride - 1) / stride;
    for (int64_t pp = p_lo; pp < P && pp * stride <= t; ++pp) {
      float f[VEC];
      Vec16<T>::unpack(*(const uint4*)(dy + pp * H + ch * VEC), f);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += f[e];
    }
    const float inv = 1.0f / (float)kernel;
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] *= inv;
    *(uint4*)(dx + t * H + ch * VEC) = Vec16<T>::pack(acc);
  }
}

// desc_dev: n_utt records {rows of the source (Lout / P), rows of the destination (Lin / T), first source row, first destination row}
extern "C" int sl_col2im_batch(const void* dcol, void* dx, const int64_t* desc_dev, int32_t n_utt, int64_t max_Lin, int32_t Cc, int32_t k, int32_t s, int32_t dtype,
                               sl_stream stream) {
  SL_CHECK_ARG(dcol && dx && desc_dev && n_utt > 0 && max_Lin > 0 && Cc > 0 && k > 0 && s > 0, "sl_col2im_batch: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(Cc % vec == 0, "sl_col2im_batch: C must be a multiple of %d", vec);
  const unsigned gx = grid_for(max_Lin * (Cc / vec), 1024);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((col2im_batch_kernel<T>), dim3(gx, (unsigned)n_utt), dim3(256), 0, (hipStream_t)stream, (const T*)dcol, (T*)dx, desc_dev, Cc, k, s);
  });
  SL_CHECK_LAUNCH("col2im_batch");
  return 0;
}

extern "C" int sl_avgpool_bwd_batch(const void* dy, void* dx, const int64_t* desc_dev, int32_t n_utt, int64_t max_T, int32_t H, int32_t kernel, int32_t stride,
                                    int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(dy && dx && desc_dev && n_utt > 0 && max_T > 0 && H > 0 && kernel > 0 && stride > 0, "sl_avgpool_bwd_batch: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(H % vec == 0, "sl_avgpool_bwd_batch: H must be a multiple of %d", vec);
  const unsigned gx = grid_for(max_T * (H / vec), 1024);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((avgpool_bwd_batch_kernel<T>), dim3(gx, (unsigned)n_utt), dim3(256), 0, (hipStream_t)stream, (const T*)dy, (T*)dx, desc_dev, H, kernel, stride);
  });
  SL_CHECK_LAUNCH("avgpool_bwd_batch");
  return 0;
}

extern "C" int sl_col2im(const void* dcol, void* dx, int64_t Lin, int64_t Lout, int32_t Cc, int32_t k, int32_t s, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(dcol && dx && Lin > 0 && Lout > 0 && Cc > 0 && k > 0 && s > 0, "sl_col2im: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(Cc % vec == 0, "sl_col2im: C must be a multiple of %d", vec);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((col2im_kernel<T>), dim3(grid_for(Lin * (Cc / vec))), dim3(256), 0, (hipStream_t)stream, (const T*)dcol, (T*)dx, Lin, Lout, Cc, k, s);
  });
  SL_CHECK_LAUNCH("col2im");
  return 0;
}

template <typename T, int CPL>
static int launch_conv0_bwd(const float* wave, int64_t n, const float* w, const float* b, const float* g, const float* be, const void* dy, int64_t L,
                            float eps, float* dw, float* db, float* dg, float* dbe, hipStream_t st) {
  constexpr int TS = (128 - 5) / 5;
  const int64_t strips = ceil_div64(L, TS);
  hipLaunchKernelGGL((conv0_bwd_kernel<T, CPL, 10, 5>), dim3((unsigned)ceil_div64(strips, 4)), dim3(256), 0, st, wave, n, w, b, g, be, (const T*)dy, L, eps, dw,
                     db, dg, dbe);
  SL_CHECK_LAUNCH("conv0_bwd");
  return 0;
}

template <typename T, int CPL>
static int launch_conv0_bwd_batch(const float* waves, const int64_t* soff, const int64_t* row0, const int64_t* spref, int n_utt, int64_t total_strips,
                                  const float* w, const float* b, const float* g, const float* be, const void* dy, float eps, float* dw, float* db,
                                  float* dg, float* dbe, hipStream_t st) {
  const int fold = sl_env().conv0_fold != 0;
  // folded: >= 4 strips per wave (4 waves per block) before the grid widens, and at 512 channels (256 registers per lane: one block per CU is
  // resident) no second round of blocks — each would only add a flush
  const int64_t blocks = ceil_div64(total_strips, fold ? 16 : 4);
  const int64_t cap = fold && CPL >= 8 ? 256 : 512;
  hipLaunchKernelGGL((conv0_bwd_batch_kernel<T, CPL, 10, 5>), dim3((unsigned)(blocks < cap ? blocks : cap)), dim3(256), 0, st, waves, soff, row0, spref, n_utt,
                     w, b, g, be, (const T*)dy, eps, dw, db, dg, dbe, fold);
  SL_CHECK_LAUNCH("conv0_bwd_batch");
  return 0;
}

extern "C" int sl_hubert_conv0_bwd_batch(const float* waves, const int64_t* sample_offsets_dev, const int64_t* row_offsets_dev,
                                         const int64_t* strip_prefix_dev, int32_t n_utt, int64_t total_strips, const float* w, const float* bias,
                                         const float* gamma, const float* beta, const void* dy, int32_t C, int32_t k, int32_t stride, float eps,
                                         float* dw, float* dbias, float* dgamma, float* dbeta, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(waves && sample_offsets_dev && row_offsets_dev && strip_prefix_dev && w && bias && gamma && beta && dy && dw && dbias && dgamma && dbeta &&
                   n_utt > 0 && total_strips > 0, "sl_hubert_conv0_bwd_batch: bad arguments");
  SL_CHECK_ARG(k == 10 && stride == 5, "sl_hubert_conv0_bwd_batch: only k=10, stride=5 is built");
  hipStream_t st = (hipStream_t)stream;
  SL_DISPATCH_DTYPE(dtype, T, {
    switch (C) {
      case 64: return launch_conv0_bwd_batch<T, 1>(waves, sample_offsets_dev, row_offsets_dev, strip_prefix_dev, n_utt, total_strips, w, bias, gamma, beta, dy, eps, dw, dbias, dgamma, dbeta, st);
      case 128: return launch_conv0_bwd_batch<T, 2>(waves, sample_offsets_dev, row_offsets_dev, strip_prefix_dev, n_utt, total_strips, w, bias, gamma, beta, dy, eps, dw, dbias, dgamma, dbeta, st);
      case 256: return launch_conv0_bwd_batch<T, 4>(waves, sample_offsets_dev, row_offsets_dev, strip_prefix_dev, n_utt, total_strips, w, bias, gamma, beta, dy, eps, dw, dbias, dgamma, dbeta, st);
      case 512: return launch_conv0_bwd_batch<T, 8>(waves, sample_offsets_dev, row_offsets_dev, strip_prefix_dev, n_utt, total_strips, w, bias, gamma, beta, dy, eps, dw, dbias, dgamma, dbeta, st);
      default: sl_set_error("sl_hubert_conv0_bwd_batch: C=%d must be 64, 128, 256 or 512", C); return SL_ERR_ARG;
    }
  });
}

extern "C" int sl_hubert_conv0_bwd(const float* wave, int64_t n_samples, const float* w, const float* bias, const float* gamma, const float* beta,
                                   const void* dy, int32_t C, int32_t k, int32_t stride, float eps, float* dw, float* dbias, float* dgamma,
                                   float* dbeta, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(wave && w && bias && gamma && beta && dy && dw && dbias && dgamma && dbeta, "sl_hubert_conv0_bwd: null pointer");
  SL_CHECK_ARG(k == 10 && stride == 5 && n_samples >= k, "sl_hubert_conv0_bwd: only k=10, stride=5 is built");
  const int64_t L = (n_samples - k) / stride + 1;
  hipStream_t st = (hipStream_t)stream;
  SL_DISPATCH_DTYPE(dtype, T, {
    switch (C) {
      case 64: return launch_conv0_bwd<T, 1>(wave, n_samples, w, bias, gamma, beta, dy, L, eps, dw, dbias, dgamma, dbeta, st);
      case 128: return launch_conv0_bwd<T, 2>(wave, n_samples, w, bias, gamma, beta, dy, L, eps, dw, dbias, dgamma, dbeta, st);
      case 256: return launch_conv0_bwd<T, 4>(wave, n_samples, w, bias, gamma, beta, dy, L, eps, dw, dbias, dgamma, dbeta, st);
      case 512: return launch_conv0_bwd<T, 8>(wave, n_samples, w, bias, gamma, beta, dy, L, eps, dw, dbias, dgamma, dbeta, st);
      default: sl_set_error("sl_hubert_conv0_bwd: C=%d must be 64, 128, 256 or 512", C); return SL_ERR_ARG;
    }
  });
}


// ----------------------------------------------------------------------------------------------
// AdamW over every trainable tensor in ONE launch (torch.optim.AdamW's arithmetic, ref:trainer.py:97-105,380-383), with the
// compute-dtype copy of the updated weight written in the same pass.  torch's foreach path is ~10 passes over the 1.27 GB of
// fp32 masters per optimizer step plus a cast / copy per tensor to refresh the kernels' weights; here a parameter element is read
// (p, g, m, v) and written (p, m, v, bf16 copy) exactly once: 30 B per parameter.
// Block b works on tensor t = the last one with first_block[t] <= b (binary search over the prefix table), elements
// [ (b - first_block[t]) * ADAMW_CHUNK, ... ).
// ----------------------------------------------------------------------------------------------
constexpr int ADAMW_CHUNK = 4096;   // elements per block: 256 threads x 4 floats x 4 rounds

template <bool DST_BF16>
__device__ __forceinline__ void adamw_store_dst(void* dst, int64_t i, const float (&o)[4], int n_valid) {
  if constexpr (DST_BF16) {
    bf16_t* d = (bf16_t*)dst + i;
    if (n_valid == 4 && ((uintptr_t)d & 7) == 0) {
      *(uint2*)d = uint2{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3])};
    } else {
      for (int e = 0; e < n_valid; ++e) d[e] = from_f32<bf16_t>(o[e]);
    }
  } else {
    float* d = (float*)dst + i;
    for (int e = 0; e < n_valid; ++e) d[e] = o[e];
  }
}

__global__ __launch_bounds__(256) void adamw_multi_kernel(const sl_adamw_tensor* __restrict__ tensors, const int64_t* __restrict__ first_block,
                                                          int n_tensors, float decay, float w1, float beta2, float w2, float eps, float step_size,
                                                          float bias_c2_sqrt) {
  int lo = 0, hi = n_tensors - 1;
  const int64_t b = blockIdx.x;
  while (lo < hi) {                      // last t with first_block[t] <= b
    const int mid = (lo + hi + 1) >> 1;
    if (first_block[mid] <= b) lo = mid; else hi = mid - 1;
  }
  const sl_adamw_tensor t = tensors[lo];
  const int64_t base = (b - first_block[lo]) * ADAMW_CHUNK;
  const bool al = (((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0;
#pragma unroll
  for (int r = 0; r < ADAMW_CHUNK / 1024; ++r) {
    const int64_t i = base + r * 1024 + threadIdx.x * 4;
    if (i >= t.n) break;
    const int nv = (t.n - i) >= 4 ? 4 : (int)(t.n - i);
    float p[4], g[4], m[4], v[4];
    if (nv == 4 && al) {
      const float4 P = *(const float4*)(t.p + i), G = *(const float4*)(t.g + i), M = *(const float4*)(t.m + i), V = *(const float4*)(t.v + i);
      p[0] = P.x; p[1] = P.y; p[2] = P.z; p[3] = P.w; g[0] = G.x; g[1] = G.y; g[2] = G.z; g[3] = G.w;
      m[0] = M.x; m[1] = M.y; m[2] = M.z; m[3] = M.w; v[0] = V.x; v[1] = V.y; v[2] = V.z; v[3] = V.w;
    } else {
      for (int e = 0; e < nv; ++e) { p[e] = t.p[i + e]; g[e] = t.g[i + e]; m[e] = t.m[i + e]; v[e] = t.v[i + e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (e >= nv) break;
      p[e] = p[e] * decay;                                   // param.mul_(1 - lr * weight_decay)
      m[e] = m[e] + w1 * (g[e] - m[e]);                      // exp_avg.lerp_(grad, 1 - beta1)
      v[e] = v[e] * beta2 + (w2 * g[e]) * g[e];              // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
      const float denom = sqrtf(v[e]) / bias_c2_sqrt + eps;  // (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
      p[e] = p[e] - step_size * (m[e] / denom);              // param.addcdiv_(exp_avg, denom, value = -step_size)
    }
    if (nv == 4 && al) {
      *(float4*)(t.p + i) = float4{p[0], p[1], p[2], p[3]};
      *(float4*)(t.m + i) = float4{m[0], m[1], m[2], m[3]};
      *(float4*)(t.v + i) = float4{v[0], v[1], v[2], v[3]};
    } else {
      for (int e = 0; e < nv; ++e) { t.p[i + e] = p[e]; t.m[i + e] = m[e]; t.v[i + e] = v[e]; }
    }
    if (t.dst) {
      if (t.dst_dtype == SL_BF16) adamw_store_dst<true>(t.dst, i, p, nv); else adamw_store_dst<false>(t.dst, i, p, nv);
    }
  }
}

extern "C" size_t sl_adamw_blocks(int64_t n) { return n <= 0 ? 0 : (size_t)((n + ADAMW_CHUNK - 1) / ADAMW_CHUNK); }

extern "C" int sl_adamw_step(const sl_adamw_tensor* tensors_dev, const int64_t* first_block_dev, int32_t n_tensors, int64_t total_blocks, double lr,
                             double beta1, double beta2, double eps, double weight_decay, int64_t step, sl_stream stream) {
  SL_CHECK_ARG(tensors_dev && first_block_dev && n_tensors > 0 && total_blocks > 0 && total_blocks < (int64_t)1 << 31 && step >= 1,
               "sl_adamw_step: bad arguments (n_tensors=%d total_blocks=%lld step=%lld)", n_tensors, (long long)total_blocks, (long long)step);
  SL_CHECK_ARG(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0. && weight_decay >= 0., "sl_adamw_step: bad hyper-parameters");
  // the scalars torch.optim.adamw._multi_tensor_adamw forms on the host, in double (python floats), before they meet fp32 tensors:
  // 1 - beta, the bias corrections, lr / bias_correction1, 1 - lr * weight_decay — each rounded to fp32 once, as torch's scalar ops do
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  const float step_size = (float)(lr / bc1), bc2_sqrt = (float)sqrt(bc2), decay = (float)(1.0 - lr * weight_decay);
  hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, tensors_dev, first_block_dev, n_tensors, decay,
                     (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, step_size, bc2_sqrt);
  SL_CHECK_LAUNCH("adamw_multi");
  return 0;
}

// ----------------------------------------------------------------------------------------------
// Weight-norm backward of the positional conv (hf:models/hubert/modeling_hubert.py:50-68 wraps the conv in weight_norm(dim = 2):
// W[h][j][t] = g[t] v[h][j][t] / ||v[:, :, t]||).  The tape leaves dW in the kernel layout (H, k, Hg) fp32; the optimizer wants
//   d g[t] = <dW[:, :, t], v[:, :, t]> / ||v_t||,     d v = (g / ||v_t||) (dW - v <dW_t, v_t> / ||v_t||^2)     in v's (H, Hg, k) layout.
// Two launches, fixed summation order: per-block partial {||v_t||^2, <dW_t, v_t>} over a run of h (the (k, Hg) slab of dW turned
// through LDS so both tensors are read in whole lines), then every block sums the partial records in block order and finishes its
// run.  Replaces six torch element-wise / reduction launches and a permuted copy per optimizer step.
// ----------------------------------------------------------------------------------------------
constexpr int WN_BLOCKS = 64;

template <bool FINISH>
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(const float* __restrict__ dWk, const float* __restrict__ v, const float* __restrict__ g,
                                                              float* __restrict__ dg, float* __restrict__ dv, float* __restrict__ ws, int H, int Hg, int k) {
  extern __shared__ float wn_lds[];          // [k][Hg + 1] slab of dW, then 2 k totals
  const int tid = threadIdx.x, b = blockIdx.x, nb = gridDim.x;
  const int h0 = (int)((int64_t)H * b / nb), h1 = (int)((int64_t)H * (b + 1) / nb);
  const int pitch = Hg + 1;
  float* tot = wn_lds + k * pitch;
  if constexpr (FINISH) {
    for (int i = tid; i < 2 * k; i += 256) {
      float s_ = 0.f;
      for (int bb = 0; bb < nb; ++bb) s_ += ws[(int64_t)bb * 2 * k + i];
      tot[i] = s_;
    }
    __syncthreads();
    if (b == 0)
      for (int t = tid; t < k; t += 256) dg[t] = tot[k + t] * rsqrtf(tot[t]);
  }
  float n2 = 0.f, dt_ = 0.f;                  // thread t < k: tap t's sums over this block's rows
  for (int h = h0; h < h1; ++h) {
    __syncthreads();
    for (int i = tid; i < k * Hg; i += 256) wn_lds[(i / Hg) * pitch + (i % Hg)] = dWk[(int64_t)h * k * Hg + i];
    __syncthreads();
    if constexpr (!FINISH) {
      if (tid < k) {
        for (int j = 0; j < Hg; ++j) {
          const float vv = v[((int64_t)h * Hg + j) * k + tid];
          n2 = fmaf(vv, vv, n2);
          dt_ = fmaf(wn_lds[tid * pitch + j], vv, dt_);
        }
      }
    } else {
      for (int i = tid; i < Hg * k; i += 256) {
        const int j = i / k, t = i % k;
        const float inv = rsqrtf(tot[t]);
        const float vv = v[((int64_t)h * Hg + j) * k + t];
        dv[((int64_t)h * Hg + j) * k + t] = g[t] * inv * (wn_lds[t * pitch + j] - vv * tot[k + t] * inv * inv);
      }
    }
  }
  if constexpr (!FINISH) {
    if (tid < k) { ws[(int64_t)b * 2 * k + tid] = n2; ws[(int64_t)b * 2 * k + k + tid] = dt_; }
  }
}

extern "C" size_t sl_weight_norm_bwd_workspace_bytes(int32_t k) { return (size_t)WN_BLOCKS * 2 * (size_t)(k > 0 ? k : 0) * sizeof(float); }

extern "C" int sl_weight_norm_bwd(const float* dW_khg, const float* v, const float* g, float* dg, float* dv, float* workspace, int32_t H, int32_t Hg,
                                  int32_t k, sl_stream stream) {
  SL_CHECK_ARG(dW_khg && v && g && dg && dv && workspace && H > 0 && Hg > 0 && k > 0 && k <= 256, "sl_weight_norm_bwd: bad arguments (H=%d Hg=%d k=%d)", H, Hg, k);
  const size_t lds = ((size_t)k * (Hg + 1) + 2 * (size_t)k) * sizeof(float);
  SL_CHECK_ARG(lds <= 64 * 1024, "sl_weight_norm_bwd: a (k, Hg) slab of %zu bytes does not fit the staging LDS", lds);
  const int nb = H < WN_BLOCKS ? H : WN_BLOCKS;
  hipLaunchKernelGGL((weight_norm_bwd_kernel<false>), dim3(nb), dim3(256), lds, (hipStream_t)stream, dW_khg, v, g, dg, dv, workspace, H, Hg, k);
  SL_CHECK_LAUNCH("weight_norm_bwd (sums)");
  hipLaunchKernelGGL((weight_norm_bwd_kernel<true>), dim3(nb), dim3(256), lds, (hipStream_t)stream, dW_khg, v, g, dg, dv, workspace, H, Hg, k);
  SL_CHECK_LAUNCH("weight_norm_bwd (finish)");
  return 0;
}
