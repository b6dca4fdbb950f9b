// gemm.hip — C = act(A . W^T + bias) + residual for gfx950, bf16 (v_mfma_f32_16x16x32_bf16) and exact
// fp32 (v_mfma_f32_16x16x4_f32).  Two kernels:
//   * gemm_tiled_kernel : 128x128 block tile, 4 waves (2x2, 64x64 each), 128-byte K slabs staged
//     through XOR-swizzled LDS (conflict-free ds_read_b128), register double-buffering of the global
//     loads.  MFMA-bound shapes: HuBERT conv-as-GEMM, encoder layers, Llama prefill.
//   * gemm_skinny_kernel: M <= 64 rows (KV-cached decode, M = batch).  HBM-bound weight streaming:
//     one block owns 16 (or 32 gate/up) weight rows, its waves interleave 64-byte K steps, weight
//     fragments go global -> VGPR -> MFMA with no LDS round trip, partial sums meet in LDS once.
// A rows may overlap (lda < K): that is how the strided convolutions run without im2col.
#include <stdlib.h>

#include "common.h"
#include "gemm_internal.h"

static int g_disable_glds = 0;  // tuning switch (SL_DISABLE_GLDS=1): A/B the two staging paths in one process

// packed-weight GEMMs with more rows than this run the streaming kernel; SL_STREAM_MIN_M overrides (tuning)
static int stream_min_m() { return sl_env().stream_min_m; }

// ----------------------------------------------------------------------------------------------
// tiled kernel
// ----------------------------------------------------------------------------------------------
// shared epilogue of the tiled kernels: +bias, [aux store], act, +residual, store.  The wave owns MT x NT 16x16 fragments
// whose first row / column in the output are row_base / col_base (lane (r, q) holds rows 4q..4q+3 of column r of each).
// Fused row-wise top-1 in place of the store (greedy decode: lm_head + argmax, hf:generation/utils.py:2911-2925 `torch.argmax(
// next_token_scores)` over ref:model/audio_llama.py:67's logits).  The wave holds MT*16 rows x 64 columns; lane (r, q) has rows
// 4q..4q+3 of column r of each 16-column fragment.  Per row: the best of the lane's four fragments, then across the 16 lanes of
// the row group, always with (value, column) compared the way greedy_select_kernel does — the first maximum wins, NaN never wins.  One (value, column) pair per row and 64-column group
// goes out at [group][row]: 64 contiguous bytes per 16 rows, 1/64 of the logits the select pass would otherwise re-read.
template <typename T, int MT, int NT>
__device__ __forceinline__ void tile_argmax(const GemmP& p, f32x4 (&acc)[MT][NT], int row_base, int col_base, int q, int r, int wz) {
  static_assert(NT == 4, "one 64-column group per wave");
  if (col_base >= p.N) return;
  const T* bias = p.bias ? (const T*)p.bias + (int64_t)wz * p.sBias : nullptr;
  const int64_t g = col_base >> 6;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float bv[4];
    int bi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { bv[i] = -INFINITY; bi[i] = 0x7fffffff; }
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = col_base + n * 16 + r;
      if (col >= p.N) continue;
      const float b = bias ? to_f32(bias[col]) : 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float v = acc[m][n][i] + b;
        if (v > bv[i] || (v == bv[i] && col < bi[i])) { bv[i] = v; bi[i] = col; }   // greedy_select_kernel's rule, -inf columns included
      }
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float ov = __shfl_xor(bv[i], o, 64);
        const int oi = __shfl_xor(bi[i], o, 64);
        if (ov > bv[i] || (ov == bv[i] && oi < bi[i])) { bv[i] = ov; bi[i] = oi; }
      }
    }
    if (r == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row_base + m * 16 + 4 * q + i;
        if (row < p.M) { p.amax_val[g * p.M + row] = bv[i]; p.amax_idx[g * p.M + row] = bi[i]; }
      }
    }
  }
}

template <typename T, int ACT, int MT, int NT>
__device__ __forceinline__ void tile_epilogue_g(const GemmP& p, f32x4 (&acc)[MT][NT], int row_base, int col_base, int q, int r, int z, int wz) {
  if constexpr (ACT == SL_ACT_NONE && NT == 4) {
    if (p.amax_val) { tile_argmax<T, MT, NT>(p, acc, row_base, col_base, q, r, wz); return; }
  }
  const int64_t co = (int64_t)z * p.sC + p.cx, ro = (int64_t)z * p.sR + p.rx;
  void* Cb = p.out_f32 ? (void*)((float*)p.C + co) : (void*)((T*)p.C + co);
  const T* bias = p.bias ? (const T*)p.bias + (int64_t)wz * p.sBias : nullptr;
  const void* Rb = p.res ? (p.res_f32 ? (const void*)((const float*)p.res + ro) : (const void*)((const T*)p.res + ro)) : nullptr;
  const int row0 = row_base + q * 4;
  const int col0 = col_base + r;
  if constexpr (ACT == SL_ACT_SILU_MUL) {
    static_assert(NT % 2 == 0, "gate/up fragments come in pairs");
    const int nout = p.N >> 1;
#pragma unroll
    for (int pr = 0; pr < NT / 2; ++pr) {
      const int gcol = col0 + (2 * pr) * 16, ucol = gcol + 16;
      const int ocol = (col_base >> 1) + pr * 16 + r;
      if (ocol >= nout) continue;
      const float bg = bias ? to_f32(bias[gcol]) : 0.f, bu = bias ? to_f32(bias[ucol]) : 0.f;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int row = row0 + m * 16 + i;
          if (row < p.M) store_out<T>(p, Cb, Rb, row, ocol, silu(acc[m][2 * pr][i] + bg) * (acc[m][2 * pr + 1][i] + bu));
        }
    }
  } else {
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      const int col = col0 + n * 16;
      if (col >= p.N) continue;
      const float b = bias ? to_f32(bias[col]) : 0.f;
      float csum = 0.f;
      // residual column first, all rows at once on clamped addresses: loads under the per-row bounds test are issued
      // and waited for one by one (MT*4 memory latencies in a chain per column, measured 2x on K = 1024 products)
      constexpr int MG = ACT == SL_ACT_GELU ? 1 : (MT < 4 ? MT : 4);   // 16 residual loads in flight per column (4 beside erf: more spills the 256-row tile)
#pragma unroll
      for (int mg = 0; mg < MT; mg += MG) {
        float rv[MG][4];
        if (Rb) {
#pragma unroll
          for (int m = 0; m < MG; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              int row = row0 + (mg + m) * 16 + i;
              row = row < p.M ? row : p.M - 1;
              rv[m][i] = p.res_f32 ? ((const float*)Rb)[(int64_t)row * p.ldr + col] : to_f32(((const T*)Rb)[(int64_t)row * p.ldr + col]);
            }
        }
#pragma unroll
        for (int m = 0; m < MG; ++m)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int row = row0 + (mg + m) * 16 + i;
            if (row < p.M) {
              float v = acc[mg + m][n][i] + b;
              if (p.aux) ((T*)p.aux + co)[(int64_t)row * p.ldc + col] = from_f32<T>(v);
              if constexpr (ACT == SL_ACT_GELU) v = gelu_act<T>(v);
              if (p.post == SL_POST_SILU_MUL_BWD) {       // (M, 2 N) output in the interleaved [16 gate | 16 up] layout
                const int64_t o = (int64_t)row * p.post_ld + 32 * (col >> 4) + (col & 15);
                float dg, du;
                post_silu_bwd<T>(v, to_f32(((const T*)p.post_in)[o]), to_f32(((const T*)p.post_in)[o + 16]), dg, du);
                T* op = (T*)Cb + (int64_t)row * p.ldc + 32 * (col >> 4) + (col & 15);
                op[0] = from_f32<T>(dg); op[16] = from_f32<T>(du);
                continue;
              }
              if (p.post) { float v1[1] = {v}; post_apply<T, 1>(p, row, col, v1); v = v1[0]; }
              if (Rb) v += rv[m][i];
              store_out<T>(p, Cb, nullptr, row, col, v);
              if (p.colsum) csum += p.out_f32 ? v : round_as<T>(v);
            }
          }
      }
      if (p.colsum) {          // the lane's rows of this column, then the four row groups of the wave: one atomic per column and wave
        csum += __shfl_xor(csum, 16, 64);
        csum += __shfl_xor(csum, 32, 64);
        if (q == 0) atomicAdd(p.colsum + col, csum);
      }
    }
  }
}

// four consecutive output elements <-> registers: 8-byte (bf16) / 16-byte (f32) accesses
__device__ __forceinline__ void ld4(const float* ptr, float (&f)[4]) {
  const f32x4 v = *(const f32x4*)ptr;
  f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
}
__device__ __forceinline__ void ld4(const bf16_t* ptr, float (&f)[4]) {
  const uint2 u = *(const uint2*)ptr;
  f[0] = bf16_bits_to_f32(u.x & 0xffffu); f[1] = bf16_bits_to_f32(u.x >> 16);
  f[2] = bf16_bits_to_f32(u.y & 0xffffu); f[3] = bf16_bits_to_f32(u.y >> 16);
}
__device__ __forceinline__ void unpack4(const f32x4& v, float (&f)[4]) { f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3]; }
__device__ __forceinline__ void unpack4(const uint2& u, float (&f)[4]) {
  f[0] = bf16_bits_to_f32(u.x & 0xffffu); f[1] = bf16_bits_to_f32(u.x >> 16);
  f[2] = bf16_bits_to_f32(u.y & 0xffffu); f[3] = bf16_bits_to_f32(u.y >> 16);
}
__device__ __forceinline__ void st4(float* ptr, const float (&f)[4]) { *(f32x4*)ptr = f32x4{f[0], f[1], f[2], f[3]}; }
__device__ __forceinline__ void st4(bf16_t* ptr, const float (&f)[4]) { *(uint2*)ptr = make_uint2(pack2_bf16(f[0], f[1]), pack2_bf16(f[2], f[3])); }

// Sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), left in every lane: four rotate-and-add steps on the VALU's DPP path
// (row_ror 8, 4, 2, 1), no LDS crossbar traffic — the order of the additions is fixed, so the result is reproducible.
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// Row-contiguous epilogue: the MFMA accumulator layout gives a lane ONE column of four rows, so the direct epilogue
// above moves 2-byte elements (a wave-level access = 4 rows x 32 B; the residual read alone doubled the time of the
// K = 1024 encoder products).  Here each wave turns its 64 x 64 sub-tile through its own 16 KiB of the (now idle)
// staging LDS — written in accumulator layout, column index XOR 16*(row/4 % 4) so the four row groups of a store hit
// different banks, read back as rows — and 16 lanes then cover 128 contiguous bytes of one output row: bias, residual,
// pre-activation copy and result all move as 8/16-byte vectors.  Returns false (nothing done) when the operands do
// not allow 4-element vectors; the caller falls back to the direct epilogue.
// The features of a launch are uniform, but tested per row pass they leave ~10 scalar branches in each pass and the compiler
// cannot move the LDS read of pass t+1 over them (one block per CU: the epilogue is an exposed tail of every tile).  F fixes
// them at compile time for the forms the encoder / prefill / KD launches use; EPI_GENERIC keeps every test at run time.
// EPI_POST (with EPI_GENERIC): the training tapes' post-ops (sl_gemm_ex_args.post_op / colsum_out) — their own instantiation, so that the plain
// generic form keeps its registers; EPI_SBWD on top of it: SL_POST_SILU_MUL_BWD (two prefetched operand rows per pass).  Swapped-operand
// epilogue: EPI_DROP / EPI_GBWD / EPI_SBWD select the post-op at compile time.
enum : int { EPI_GENERIC = 1, EPI_RES = 2, EPI_LN = 4, EPI_STATS = 8, EPI_AUX = 16, EPI_POST = 32, EPI_SBWD = 64, EPI_DROP = 128, EPI_GBWD = 256 };

template <typename T, int ACT, int MT, int F>
__device__ __forceinline__ void tile_epilogue_rows_impl(const GemmP& p, f32x4 (&acc)[MT][4], int row_base, int col_base, int lane, int wz, float* wsm,
                                                        const float2* mr_lds, int64_t co, int64_t ro) {
  constexpr bool G = (F & EPI_GENERIC) != 0, BF = sizeof(T) == 2;      // the LayerNorm fold is a bf16 form (sl_gemm_impl checks)
  const bool f_aux = G && p.aux != nullptr, f_out32 = G && p.out_f32, f_res32 = G && p.res && p.res_f32;
  const bool f_rest = G ? (p.res && !p.res_f32) : (F & EPI_RES) != 0;
  const bool f_ln = BF && (G ? p.ln_mr != nullptr : (F & EPI_LN) != 0);
  const bool f_stats = BF && (G ? p.stats_out != nullptr : (F & EPI_STATS) != 0);
  constexpr bool PO = (F & EPI_POST) != 0, SB = (F & EPI_SBWD) != 0;
  const int f_post = PO ? p.post : 0;                 // training-tape post-ops (their own instantiations: tile_epilogue_rows)
  const bool f_cs = PO && p.colsum != nullptr;
  const bool f_pin = PO && !SB && p.post == SL_POST_GELU_BWD;      // the saved pre-activation rows are requested up front, like a residual
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  const int q = lane >> 4, r = lane & 15;
  const int c4 = r * 4;                       // read phase: lane = (row within a 4-row pass, 4-column group)
  const int col = col_base + c4;
  const bool col_ok = col < p.N;
  const int colc = col_ok ? col : 0;
  float b4[4] = {0.f, 0.f, 0.f, 0.f};
  if (p.bias && col_ok) {
    const T* bias = (const T*)p.bias + (int64_t)wz * p.sBias + col;
#pragma unroll
    for (int j = 0; j < 4; ++j) b4[j] = to_f32(bias[j]);
  }
  // LayerNorm fold, consumer side: the four columns' ln_u / ln_c stay in registers, {mean, rstd} come per row pass
  float u4[4] = {0.f, 0.f, 0.f, 0.f};
  if (f_ln && col_ok) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { u4[j] = p.ln_u[col + j]; b4[j] = p.ln_c[col + j]; }
  }
  const int segs = (p.N + 63) >> 6, seg = col_base >> 6;
  using RawT = typename std::conditional<sizeof(T) == 2, uint2, f32x4>::type;   // four residual elements of type T as loaded
#pragma unroll
  for (int mg = 0; mg < MT; mg += 4) {
    // residual in the output's type (the encoder / prefill form): all 16 row passes of this 64-row group are requested
    // before the tile is turned through LDS, on clamped addresses, so one memory latency is exposed per group (issued
    // pass by pass under the bounds test they cost ~45 % on the K = 1024 products)
    RawT raw[16];
    RawT raw2[SB ? 16 : 1];
    if (f_rest) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        int64_t row = row_base + mg * 16 + t * 4 + q;
        row = row < p.M ? row : p.M - 1;
        raw[t] = *(const RawT*)((const T*)p.res + ro + row * p.ldr + colc);
      }
    } else if (f_pin) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        int64_t row = row_base + mg * 16 + t * 4 + q;
        row = row < p.M ? row : p.M - 1;
        raw[t] = *(const RawT*)((const T*)p.post_in + row * p.post_ld + colc);
      }
    }
    if constexpr (SB) {       // gate and up pre-activations of the lane's four columns: [16 gate | 16 up] blocks
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        int64_t row = row_base + mg * 16 + t * 4 + q;
        row = row < p.M ? row : p.M - 1;
        const T* gp = (const T*)p.post_in + row * p.post_ld + 32 * (colc >> 4) + (colc & 15);
        raw[t] = *(const RawT*)gp;
        raw2[t] = *(const RawT*)(gp + 16);
      }
    }
    // LayerNorm fold, consumer side: {mean, rstd} of the group's 16 row passes, requested up front for the same reason (a load
    // issued between the stores of two passes waits for those stores: vmcnt retires in order) — unless the 256-row tile kernel
    // staged its rows' pairs in LDS under the main loop (mr_lds, indexed by the row within the wave's tile)
    float2 keep = make_float2(0.f, 0.f);
    float2 mr[16];
    if (f_ln && !mr_lds) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        int64_t row = row_base + mg * 16 + t * 4 + q;
        row = row < p.M ? row : p.M - 1;
        mr[t] = ((const float2*)p.ln_mr)[row];
      }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) wsm[(m * 16 + 4 * q + i) * 64 + ((n * 16 + r) ^ (q << 4))] = acc[mg + m][n][i];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t0 = 0; t0 < 16; t0 += 4) {
      float rv[4][4];
      if (f_res32) {   // fp32 accumulation targets (weight gradients): four passes at a time
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          int64_t row = row_base + mg * 16 + (t0 + u) * 4 + q;
          row = row < p.M ? row : p.M - 1;
          ld4((const float*)p.res + ro + row * p.ldr + colc, rv[u]);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u, lr = t * 4 + q;
        const int64_t row = row_base + mg * 16 + lr;
        const f32x4 a = *(const f32x4*)&wsm[lr * 64 + (c4 ^ ((t & 3) << 4))];
        float v[4];
        if (f_ln) {   // rstd a + (c - rstd mean u): two packed fp32 FMAs per pair of columns
          const float2 mrt = mr_lds ? mr_lds[mg * 16 + lr] : mr[t];
          const float nk = -mrt.y * mrt.x;
          const f32x2_t k2 = {nk, nk}, r2 = {mrt.y, mrt.y};
#pragma unroll
          for (int j = 0; j < 4; j += 2) {
            const f32x2_t t2 = __builtin_elementwise_fma(k2, f32x2_t{u4[j], u4[j + 1]}, f32x2_t{b4[j], b4[j + 1]});
            const f32x2_t v2 = __builtin_elementwise_fma(r2, f32x2_t{a[j], a[j + 1]}, t2);
            v[j] = v2[0]; v[j + 1] = v2[1];
          }
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = a[j] + b4[j];
        }
        float s1 = 0.f, s2 = 0.f;   // LayerNorm fold, producer side: statistics of the values as stored
        if (row < p.M && col_ok) {
          if (f_aux) st4((T*)p.aux + co + row * p.ldc + col, v);
          if constexpr (ACT == SL_ACT_GELU) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = gelu_act<T>(v[j]);
          }
          if constexpr (SB) {     // (M, 2 N) output, interleaved [16 gate | 16 up]: the lane's four columns sit in one 16-group
            float g4[4], up4[4], dg[4], du[4];
            unpack4(raw[t], g4);
            unpack4(raw2[t], up4);
#pragma unroll
            for (int j = 0; j < 4; ++j) post_silu_bwd<T>(v[j], g4[j], up4[j], dg[j], du[j]);
            T* op = (T*)p.C + co + row * p.ldc + 32 * (col >> 4) + (col & 15);
            st4(op, dg);
            st4(op + 16, du);
            continue;
          }
          if (f_post == SL_POST_DROPOUT) {
            post_drop<T, 4>(p, row, col, v);
          } else if (f_pin) {
            post_drop<T, 4>(p, row, col, v);
            float pre4[4];
            unpack4(raw[t], pre4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = round_as<T>(v[j]) * gelu_grad(pre4[j]);
          }
          if (f_rest) {
            float rr[4];
            unpack4(raw[t], rr);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += rr[j];
          } else if (f_res32) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += rv[u][j];
          }
          if (f_out32) st4((float*)p.C + co + row * p.ldc + col, v);
          else st4((T*)p.C + co + row * p.ldc + col, v);
          if (f_cs) {
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] += f_out32 ? v[j] : round_as<T>(v[j]);
          }
          if (f_stats) {
            f32x2_t f01 = {v[0], v[1]}, f23 = {v[2], v[3]};
            if (!f_out32) {               // the values as stored: the same v_cvt_pk_bf16_f32 st4 issued, its halves shifted back up
              const uint32_t lo = pack2_bf16(v[0], v[1]), hi = pack2_bf16(v[2], v[3]);
              f01 = f32x2_t{__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xffff0000u)};
              f23 = f32x2_t{__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
            }
            const f32x2_t a2 = f01 + f23, q2 = __builtin_elementwise_fma(f23, f23, f01 * f01);
            s1 = a2[0] + a2[1];
            s2 = q2[0] + q2[1];
          }
        }
        if (f_stats) {             // the 16 lanes of a row pass cover the wave's 64 columns: fixed-order sum
          s1 = row16_sum(s1); s2 = row16_sum(s2);
          if (r == t) keep = make_float2(s1, s2);     // every lane of the row has the sums; lane r holds on to pass r's
        }
      }
    }
    if (f_stats) {                   // one store per 64-row group: lane (q, r) has row 4 r + q of it
      const int64_t row = row_base + mg * 16 + r * 4 + q;
      if (row < p.M && col_base < p.N) ((float2*)p.stats_out)[row * segs + seg] = keep;
    }
    __builtin_amdgcn_wave_barrier();
  }
  if (f_cs) {      // the wave's rows of its 64 columns: the four row lanes of a column group meet, one atomic per column and wave
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      cs[j] += __shfl_xor(cs[j], 16, 64);
      cs[j] += __shfl_xor(cs[j], 32, 64);
    }
    if (q == 0 && col_ok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(p.colsum + col + j, cs[j]);
    }
  }
}

// POSTS: this kernel may be handed products with training-tape post-ops (launch_tiled routes them to the LDS-DMA 128-tile kernel and the
// phased 256-tile kernel only — the other kernels do not carry those instantiations: compile time)
template <typename T, int ACT, int MT, bool POSTS = false>
__device__ __forceinline__ bool tile_epilogue_rows(const GemmP& p, f32x4 (&acc)[MT][4], int row_base, int col_base, int lane, int z, int wz, float* wsm,
                                                   const float2* mr_lds = nullptr) {
  static_assert(MT % 4 == 0 && ACT != SL_ACT_SILU_MUL, "64-row passes; the gate/up pairing keeps the direct epilogue");
  if (p.amax_val) return false;     // fused top-1: nothing is stored, the accumulator layout is what the reduction wants
  const int64_t co = (int64_t)z * p.sC + p.cx, ro = (int64_t)z * p.sR + p.rx;
  const uintptr_t ca = p.out_f32 ? 15 : (4 * sizeof(T) - 1), ra = p.res_f32 ? 15 : (4 * sizeof(T) - 1);
  if ((p.N & 3) || (p.ldc & 3) || (co & 3) || ((uintptr_t)p.C & ca) || (p.aux && ((uintptr_t)p.aux & (4 * sizeof(T) - 1))) ||
      (p.res && ((p.ldr & 3) || (ro & 3) || ((uintptr_t)p.res & ra))) || (p.post_in && ((p.post_ld & 3) || ((uintptr_t)p.post_in & (4 * sizeof(T) - 1)))))
    return false;
  if constexpr (sizeof(T) == 2) {
    if (!p.aux && !p.out_f32 && !(p.res && p.res_f32) && !p.post && !p.colsum) {
      const bool res = p.res != nullptr, ln = p.ln_mr != nullptr, st = p.stats_out != nullptr;
      if (!ln && !st) {
        if (res) tile_epilogue_rows_impl<T, ACT, MT, EPI_RES>(p, acc, row_base, col_base, lane, wz, wsm, mr_lds, co, ro);
        else tile_epilogue_rows_impl<T, ACT, MT, 0>(p, acc, row_base, col_base, lane, wz, wsm, mr_lds, co, ro);
        return true;
      }
      if (ln && !res && !st) { tile_epilogue_rows_impl<T, ACT, MT, EPI_LN>(p, acc, row_base, col_base, lane, wz, wsm, mr_lds, co, ro); return true; }
      if (st && res && !ln) { tile_epilogue_rows_impl<T, ACT, MT, EPI_RES | EPI_STATS>(p, acc, row_base, col_base, lane, wz, wsm, mr_lds, co, ro); return true; }
    }
  }
  if constexpr (POSTS) {
    if (p.post == SL_POST_SILU_MUL_BWD) {
      if constexpr (ACT == SL_ACT_NONE) { tile_epilogue_rows_impl<T, ACT, MT, EPI_GENERIC | EPI_POST | EPI_SBWD>(p, acc, row_base, col_base, lane, wz, wsm, mr_lds, co, ro); return true; }
      else return false;
    }
    if (p.post || p.colsum) { tile_epilogue_rows_impl<T, ACT, MT, EPI_GENERIC | EPI_POST>(p, acc, row_base, col_base, lane, wz, wsm, mr_lds, co, ro); return true; }
  } else {
    if (p.post || p.colsum) return false;       // (never routed here: the direct epilogue still applies them)
  }
  tile_epilogue_rows_impl<T, ACT, MT, EPI_GENERIC>(p, acc, row_base, col_base, lane, wz, wsm, mr_lds, co, ro);
  return true;
}

// ----------------------------------------------------------------------------------------------
// Register epilogue of the swapped-operand 256-tile kernel (bf16).  With the MFMA operands exchanged (D = W_frag . A_frag^T)
// a lane holds four consecutive COLUMNS of one output row; the kernel reads W fragment n of lane r from tile row
// 32 (n >> 1) + 8 (r >> 2) + 4 (n & 1) + (r & 3), which makes lane (q, r)'s sixteen values of row m*16 + r the columns
// [8q, 8q + 8) and [32 + 8q, 32 + 8q + 8) of the wave's 64: two 16-byte stores per row, the four q's of a row filling 64
// contiguous bytes per instruction.  Nothing is turned through LDS (the LDS turn was ~30 % of the rows epilogue: 64 ds_write_b32 +
// 16 ds_read_b128 per 64-row group and wave, with the read latency in every pass's dependency chain), bias / LayerNorm-fold
// vectors stay in registers per column, the residual arrives as 16-byte loads, four rows requested at a time.
// The launch code guarantees: N, ldc, ldr, the batch strides multiples of 8, 16-byte aligned C / residual, no aux / fp32 forms.
// ----------------------------------------------------------------------------------------------
template <int ACT, int F>
__device__ __forceinline__ void tile_epilogue_sw(const GemmP& p, f32x4 (&acc)[8][4], int row_base, int col_base, int lane, int z, int wz, const float2* mr_lds) {
  using T = bf16_t;
  constexpr bool RES = (F & EPI_RES) != 0, LN = (F & EPI_LN) != 0, ST = (F & EPI_STATS) != 0, AUX = (F & EPI_AUX) != 0;
  // training-tape post-ops (sl_gemm_ex_args.post_op): dropout of the value before the residual add; GELU' x dropout behind a data-gradient product
  // (+ the bias gradient's column sums); SwiGLU' writing the (M, 2 N) interleaved gate / up gradient
  constexpr bool DROP = (F & EPI_DROP) != 0, GBWD = (F & EPI_GBWD) != 0, SBWD = (F & EPI_SBWD) != 0;
  static_assert(!(GBWD || SBWD) || !(RES || LN || ST || AUX || DROP), "the backward post-ops take the plain product");
  float csum[2][8];            // GBWD + colsum_out: this lane's rows of its sixteen columns
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[h][j] = 0.f;
  const int64_t co = (int64_t)z * p.sC, ro = (int64_t)z * p.sR;
  const int q = lane >> 4, r = lane & 15;
  int colh[2];
  bool okh[2];
  float bc[2][8], uu[2][8];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    colh[h] = col_base + 32 * h + 8 * q;
    okh[h] = colh[h] < p.N;
#pragma unroll
    for (int j = 0; j < 8; ++j) { bc[h][j] = 0.f; uu[h][j] = 0.f; }
    if (!okh[h]) { colh[h] = 0; continue; }
    if constexpr (LN) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { uu[h][j] = p.ln_u[colh[h] + j]; bc[h][j] = p.ln_c[colh[h] + j]; }
    } else if (p.bias) {
      const T* bias = (const T*)p.bias + (int64_t)wz * p.sBias + colh[h];
#pragma unroll
      for (int j = 0; j < 8; ++j) bc[h][j] = to_f32(bias[j]);
    }
  }
  float s1[8][2], s2[8][2];
#pragma unroll
  for (int mb = 0; mb < 8; mb += 4) {
    uint4 raw[4][2];
    uint4 raw2[SBWD ? 4 : 1][2];
    if constexpr (RES) {
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) {
        int64_t row = row_base + (mb + m4) * 16 + r;
        row = row < p.M ? row : p.M - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) raw[m4][h] = *(const uint4*)((const T*)p.res + ro + row * p.ldr + colh[h]);
      }
    }
    if constexpr (GBWD) {        // the saved pre-activation, four rows at a time like a residual
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) {
        int64_t row = row_base + (mb + m4) * 16 + r;
        row = row < p.M ? row : p.M - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) raw[m4][h] = *(const uint4*)((const T*)p.post_in + row * p.post_ld + colh[h]);
      }
    }
    if constexpr (SBWD) {        // gate / up pre-activations of the lane's eight columns (one half of a [16 gate | 16 up] block)
#pragma unroll
      for (int m4 = 0; m4 < 4; ++m4) {
        int64_t row = row_base + (mb + m4) * 16 + r;
        row = row < p.M ? row : p.M - 1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const T* gp = (const T*)p.post_in + row * p.post_ld + 32 * (colh[h] >> 4) + (colh[h] & 15);
          raw[m4][h] = *(const uint4*)gp;
          raw2[m4][h] = *(const uint4*)(gp + 16);
        }
      }
    }
#pragma unroll
    for (int m4 = 0; m4 < 4; ++m4) {
      const int m = mb + m4;
      const int64_t row = row_base + m * 16 + r;
      f32x2_t k2 = {0.f, 0.f}, r2 = {0.f, 0.f};
      if constexpr (LN) {
        const float2 mrt = mr_lds[m * 16 + r];
        const float nk = -mrt.y * mrt.x;
        k2 = f32x2_t{nk, nk}; r2 = f32x2_t{mrt.y, mrt.y};
      }
      float la[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, lq[2][2] = {{0.f, 0.f}, {0.f, 0.f}};     // [h][4-column leaf]: sums, sums of squares
      uint4 pkh[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; j += 2) {
          const f32x2_t a2 = {acc[m][2 * h + (j >> 2)][j & 3], acc[m][2 * h + (j >> 2)][(j & 3) + 1]};
          f32x2_t v2;
          if constexpr (LN) v2 = __builtin_elementwise_fma(r2, a2, __builtin_elementwise_fma(k2, f32x2_t{uu[h][j], uu[h][j + 1]}, f32x2_t{bc[h][j], bc[h][j + 1]}));
          else v2 = a2 + f32x2_t{bc[h][j], bc[h][j + 1]};
          v[j] = v2[0]; v[j + 1] = v2[1];
        }
        if constexpr (AUX) {            // the training forward keeps the pre-activation (after bias), same layout as C
          if (row < p.M && okh[h]) *(uint4*)((T*)p.aux + co + row * p.ldc + colh[h]) = Vec16<T>::pack(v);
        }
        if constexpr (ACT == SL_ACT_GELU) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = gelu_act<T>(v[j]);
        }
        if constexpr (SBWD) {
          float g8[8], u8[8], dg[8], du[8];
          Vec16<T>::unpack(raw[m4][h], g8);
          Vec16<T>::unpack(raw2[m4][h], u8);
#pragma unroll
          for (int j = 0; j < 8; ++j) post_silu_bwd<T>(v[j], g8[j], u8[j], dg[j], du[j]);
          if (row < p.M && okh[h]) {
            T* op = (T*)p.C + co + row * p.ldc + 32 * (colh[h] >> 4) + (colh[h] & 15);
            *(uint4*)op = Vec16<T>::pack(dg);
            *(uint4*)(op + 16) = Vec16<T>::pack(du);
          }
          continue;
        }
        if constexpr (DROP) post_drop<T, 8>(p, row, colh[h], v);
        if constexpr (GBWD) {
          post_drop<T, 8>(p, row, colh[h], v);
          float pre8[8];
          Vec16<T>::unpack(raw[m4][h], pre8);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = round_as<T>(v[j]) * gelu_grad(pre8[j]);
        }
        if constexpr (RES) {
          float rr[8];
          Vec16<T>::unpack(raw[m4][h], rr);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] += rr[j];
        }
        const uint4 pk = Vec16<T>::pack(v);
        pkh[h] = pk;
        if constexpr (GBWD) {
          if (p.colsum && row < p.M && okh[h]) {
            float sv[8];
            Vec16<T>::unpack(pk, sv);       // the values as stored
#pragma unroll
            for (int j = 0; j < 8; ++j) csum[h][j] += sv[j];
          }
        }
        if (row < p.M && okh[h]) {
          if constexpr (ST) {             // statistics of the values as stored (the packed halves shifted back up), per 4-column leaf
            const uint32_t w[4] = {pk.x, pk.y, pk.z, pk.w};   // exactly as the rows epilogue forms them: the tree below is its tree
#pragma unroll
            for (int g = 0; g < 2; ++g) {
              const f32x2_t f01 = {__builtin_bit_cast(float, w[2 * g] << 16), __builtin_bit_cast(float, w[2 * g] & 0xffff0000u)};
              const f32x2_t f23 = {__builtin_bit_cast(float, w[2 * g + 1] << 16), __builtin_bit_cast(float, w[2 * g + 1] & 0xffff0000u)};
              const f32x2_t a2 = f01 + f23, q2 = __builtin_elementwise_fma(f23, f23, f01 * f01);
              la[h][g] = a2[0] + a2[1];
              lq[h][g] = q2[0] + q2[1];
            }
          }
        }
      }
      if constexpr (!SBWD) {
        // Stores.  As computed, an instruction would put 64 bytes into each of 16 rows — 16 half-written 128-byte lines; a CU's store
        // path takes ~4 clocks per line touched whatever it carries (tools/probe_store_rate.hip, one CU storing alone: a 128 KiB tile in
        // 3.47 us that way, 1.23 us as 8 full lines per instruction, 1.84 us as the LDS-turned epilogue's 4 lines of 8-byte pieces).
        // So lanes r and r + 8 of a DPP row trade halves first (row_ror:8): the lower eight lanes then hold the left 64 bytes of rows
        // r and r + 8, the upper eight the right 64 bytes of rows r - 8 and r, and each of the two instructions writes eight whole lines.
        const bool lo = r < 8;
        const uint32_t a0[4] = {pkh[0].x, pkh[0].y, pkh[0].z, pkh[0].w}, a1[4] = {pkh[1].x, pkh[1].y, pkh[1].z, pkh[1].w};
        uint32_t e1[4], e2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {       // component-wise selects: an indexed pick between the two vectors goes through scratch memory
          const uint32_t send = lo ? a1[j] : a0[j];
          const uint32_t recv = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)send, 0x128, 0xf, 0xf, false);
          e1[j] = lo ? a0[j] : recv;
          e2[j] = lo ? recv : a1[j];
        }
        const uint4 d1 = make_uint4(e1[0], e1[1], e1[2], e1[3]), d2 = make_uint4(e2[0], e2[1], e2[2], e2[3]);
        const int64_t row1 = row_base + m * 16 + (r & 7), row2 = row1 + 8;
        const int colx = lo ? colh[0] : colh[1];
        const bool okx = lo ? okh[0] : okh[1];
        if (row1 < p.M && okx) *(uint4*)((T*)p.C + co + row1 * p.ldc + colx) = d1;       // (non-temporal stores: 0.89 x at N = K = 1024 without residual, 1.00-1.04 x on every encoder shape)
        if (row2 < p.M && okx) *(uint4*)((T*)p.C + co + row2 * p.ldc + colx) = d2;
      }
      if constexpr (ST) {   // leaves 2q, 2q + 1 (h = 0) and 2q + 8, 2q + 9 (h = 1) of the row's 16: the first level of the 16-lane tree is in-lane
        s1[m][0] = la[0][0] + la[1][0]; s1[m][1] = la[0][1] + la[1][1];
        s2[m][0] = lq[0][0] + lq[1][0]; s2[m][1] = lq[0][1] + lq[1][1];
      }
    }
  }
  if constexpr (GBWD) {
    if (p.colsum) {        // the sixteen row lanes of a column group (one DPP row) meet; lane r = 0 adds the wave's sums: one atomic per column and wave
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = row16_sum(csum[h][j]);
          if (r == 0 && okh[h]) atomicAdd(p.colsum + colh[h] + j, t);
        }
    }
  }
  if constexpr (ST) {
    // The rows epilogue sums a row's sixteen 4-column leaves l_0..l_15 as S_i = l_i + l_(i+8), E_i = S_i + S_(i+4), T_i = E_i + E_(i+2),
    // T_0 + T_1 (row16_sum); a batch and its single utterances may take different tile kernels and must get the same bits, so this
    // is that tree: lane q holds S_2q and S_2q+1, partners are q ^ 2 (lane ^ 32) and then q ^ 1 (lane ^ 16).  Lane (q, r) stores
    // the rows of m = q and m = q + 4 (two store instructions for the wave's 128 rows).
    const int segs = (p.N + 63) >> 6, seg = col_base >> 6;
    float2 k0 = make_float2(0.f, 0.f), k1 = make_float2(0.f, 0.f);
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      float a0 = s1[m][0], a1 = s1[m][1], b0 = s2[m][0], b1 = s2[m][1];
      a0 += __shfl_xor(a0, 32, 64); a1 += __shfl_xor(a1, 32, 64); b0 += __shfl_xor(b0, 32, 64); b1 += __shfl_xor(b1, 32, 64);
      a0 += __shfl_xor(a0, 16, 64); a1 += __shfl_xor(a1, 16, 64); b0 += __shfl_xor(b0, 16, 64); b1 += __shfl_xor(b1, 16, 64);
      const float a = a0 + a1, b = b0 + b1;
      if ((m & 3) == q) { if (m < 4) k0 = make_float2(a, b); else k1 = make_float2(a, b); }
    }
    if (col_base < p.N) {
      const int64_t row0 = row_base + q * 16 + r, row1 = row0 + 64;
      if (row0 < p.M) ((float2*)p.stats_out)[row0 * segs + seg] = k0;
      if (row1 < p.M) ((float2*)p.stats_out)[row1 * segs + seg] = k1;
    }
  }
}

template <typename T, int ACT>
__device__ __forceinline__ void tile_epilogue(const GemmP& p, f32x4 (&acc)[4][4], int bm, int bn, int wm, int wn, int q, int r, int z, int wz) {
  tile_epilogue_g<T, ACT, 4, 4>(p, acc, bm * TBM + wm * 64, bn * TBN + wn * 64, q, r, z, wz);
}

template <typename T, int ACT>
__global__ __launch_bounds__(256, 2) void gemm_tiled_kernel(GemmP p) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int BK = TROWB / (int)sizeof(T);  // 64 bf16 / 32 f32
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][TBM * TROWB];  // [buf][A|W]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, q = lane >> 4;

  // XCD-aware tile order: consecutive blocks of one XCD (blockIdx % 8 equal) walk tiles that share
  // the same W panel, so the panel stays in that XCD's L2 (bijective remap, guide §5 T1).
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int qn = nt >> 3, rn = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  int bm, bn;   // 8 x 8 patches of tiles per XCD at a time (see gemm_tiled256_kernel)
  {
    constexpr int GM = 8;
    const int per = GM * p.tiles_n, grp = bid / per, first = grp * GM;
    const int gsz = (p.tiles_m - first) < GM ? (p.tiles_m - first) : GM;
    const int in = bid - grp * per;
    bm = first + in % gsz;
    bn = in / gsz;
  }
  const int z = blockIdx.y;

  int64_t a_off; int wz;
  if (!resolve_group(p, z, bm, a_off, wz)) return;
  const T* A = (const T*)p.A + a_off;
  const T* W = (const T*)p.W + (int64_t)wz * p.sW + p.wx;
  if (p.grp_ext && bn * TBN >= p.N) return;

  // staging assignment: 4 chunks of A and 4 of W per thread
  const T* ga[4];
  const T* gw[4];
  int so[4];
  int kc;  // this thread's k offset (elements) inside a slab
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 3, ch = c & 7;
    int ar = bm * TBM + row; ar = ar < p.M ? ar : p.M - 1;
    int wr = bn * TBN + row; wr = wr < p.N ? wr : p.N - 1;
    ga[i] = A + (int64_t)ar * p.lda + ch * VEC;
    gw[i] = W + (int64_t)wr * p.ldw + ch * VEC;
    so[i] = lds_off(row, ch);
  }
  kc = (tid & 7) * VEC;
  // transposed operands (backward GEMMs): 16-byte chunks run along the OUTPUT index, one reduction row each
  constexpr int CPRT = 128 / VEC;  // chunks per reduction row of a 128-wide tile
  const T* gat[4];
  const T* gwt[4];
  int tk[4], tcol[4];
  bool aok[4], wok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i;
    tk[i] = c / CPRT;
    tcol[i] = (c % CPRT) * VEC;
    const int am = bm * TBM + tcol[i], wn = bn * TBN + tcol[i];
    aok[i] = am < p.M; wok[i] = wn < p.N;
    gat[i] = A + (int64_t)tk[i] * p.lda + (aok[i] ? am : 0);
    gwt[i] = W + (int64_t)tk[i] * p.ldw + (wok[i] ? wn : 0);
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = (p.K + BK - 1) / BK;
  uint4 ra[4], rw[4];

  auto gload = [&](int kt) {
    const int k0 = kt * BK;
    const bool ok = (k0 + kc) < p.K;  // K % VEC == 0, so a chunk is entirely in or out
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (p.ta) ra[i] = (aok[i] && k0 + tk[i] < p.K) ? *(const uint4*)(gat[i] + (int64_t)k0 * p.lda) : make_uint4(0, 0, 0, 0);
      else ra[i] = ok ? *(const uint4*)(ga[i] + k0) : make_uint4(0, 0, 0, 0);
      if (p.tw) rw[i] = (wok[i] && k0 + tk[i] < p.K) ? *(const uint4*)(gwt[i] + (int64_t)k0 * p.ldw) : make_uint4(0, 0, 0, 0);
      else rw[i] = ok ? *(const uint4*)(gw[i] + k0) : make_uint4(0, 0, 0, 0);
    }
  };
  // transposed chunk -> LDS: element e belongs to tile row tcol+e, reduction index tk
  auto scatter = [&](unsigned char* tile, const uint4& u, int col0, int k) {
    T e[VEC];
    *(uint4*)e = u;
#pragma unroll
    for (int j = 0; j < VEC; ++j) *(T*)(tile + lds_off(col0 + j, k / VEC) + (k % VEC) * (int)sizeof(T)) = e[j];
  };
  // bf16: a transposed operand keeps its [reduction row][128 outputs] shape in LDS (256-byte rows, the guide's image (b):
  // chunk ^ (((row & 3) << 2) | ((row >> 2) & 3))) — one 16-byte store per chunk instead of eight 2-byte scatters — and
  // the MFMA fragments are gathered column-wise by ds_read_b64_tr_b16 (conflict-free: a 32-lane half reads two 4-row
  // blocks 8 rows apart).  The 2-byte scatter remains for fp32 (no 32-bit transposed read).
  constexpr bool TRT = sizeof(T) == 2;
  auto t_off = [](int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (p.ta) {
        if constexpr (TRT) *(uint4*)(&smem[buf][0][t_off(tk[i], tcol[i] / VEC)]) = ra[i];
        else scatter(&smem[buf][0][0], ra[i], tcol[i], tk[i]);
      } else *(uint4*)(&smem[buf][0][so[i]]) = ra[i];
      if (p.tw) {
        if constexpr (TRT) *(uint4*)(&smem[buf][1][t_off(tk[i], tcol[i] / VEC)]) = rw[i];
        else scatter(&smem[buf][1][0], rw[i], tcol[i], tk[i]);
      } else *(uint4*)(&smem[buf][1][so[i]]) = rw[i];
    }
  };
  // fragment of 16 outputs x 32 reduction steps out of a transposed-image tile: rows s*32 + 8q .. +7, columns col0 + r
  const int qq = r >> 2, pp = r & 3;
  typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_g_t;
  // the four 16-output fragments of a wave's 64 outputs: all eight reads in flight, one wait
  auto tr_frags = [&](const unsigned char* tile, int s_, int col0, uint4 (&f)[4]) {
    const uint32_t base = (uint32_t)(uintptr_t)(lds_ptr_t)tile;
    const int r0 = s_ * 32 + 8 * q + qq;
    u32x2_g_t lo[4], hi[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int ch0 = (col0 + 16 * j) / 8;
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[j]) : "v"(base + (uint32_t)(t_off(r0, ch0 + (pp >> 1)) + 8 * (pp & 1))) : "memory");
      asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi[j]) : "v"(base + (uint32_t)(t_off(r0 + 4, ch0 + (pp >> 1)) + 8 * (pp & 1))) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]));
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = make_uint4(lo[j].x, lo[j].y, hi[j].x, hi[j].y);
  };

  gload(0);
  sstore(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const unsigned char* sa = &smem[buf][0][0];
    const unsigned char* sw = &smem[buf][1][0];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 fa[4], fb[4];
      if (TRT && p.ta) {
        if constexpr (TRT) {
          tr_frags(sa, s, wm * 64, fa);
        }
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[m] = *(const uint4*)(sa + lds_off(wm * 64 + m * 16 + r, s * 4 + q));
      }
      if (TRT && p.tw) {
        if constexpr (TRT) {
          tr_frags(sw, s, wn * 64, fb);
        }
      } else {
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[n] = *(const uint4*)(sw + lds_off(wn * 64 + n * 16 + r, s * 4 + q));
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) MMA<T>::step(acc[m][n], fa[m], fb[n]);
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }

  if constexpr (ACT != SL_ACT_SILU_MUL) {
    if (!p.direct_epi && tile_epilogue_rows<T, ACT, 4>(p, acc, bm * TBM + wm * 64, bn * TBN + wn * 64, lane, z, wz, (float*)&smem[0][0][0] + wave * 4096)) return;
  }
  tile_epilogue<T, ACT>(p, acc, bm, bn, wm, wn, q, r, z, wz);
}

// ----------------------------------------------------------------------------------------------
// tiled kernel, direct-to-LDS staging (global_load_lds_dwordx4): same tile, same swizzled LDS image — the
// swizzle moves to the per-lane SOURCE address because the LDS side of an LDS-DMA is lane-linear — no
// staging VGPRs, no ds_write pass.  Used when K is a whole number of 128-byte slabs and no operand is
// transposed (every forward GEMM of the encoder / prefill at model shapes).
// ----------------------------------------------------------------------------------------------

// ds_read_b128 the compiler cannot see: after a global_load_lds it guards every LDS read it knows about with
// s_waitcnt vmcnt(0) (it cannot prove the DMA and the read do not alias), which made the "prefetch" of the next K slab
// synchronous.  The consumer waits with lds_wait<N>(regs...) — the registers are tied to the wait so no use moves above it.
#define SL_LDS_RD(dst, addr, off) asm volatile("ds_read_b128 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
template <int N>
__device__ __forceinline__ void lds_wait8(u32x4_t& a, u32x4_t& b, u32x4_t& c, u32x4_t& d, u32x4_t& e, u32x4_t& f, u32x4_t& g, u32x4_t& h) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "n"(N));
}
__device__ __forceinline__ uint4 as_uint4(const u32x4_t& v) { return make_uint4(v.x, v.y, v.z, v.w); }

template <typename T, int ACT, bool ASMLDS = false>
__global__ __launch_bounds__(256, 2) void gemm_tiled_glds_kernel(GemmP p) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int BK = TROWB / (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][TBM * TROWB];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, q = lane >> 4;
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int qn = nt >> 3, rn = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  int bm, bn;   // 8 x 8 patches of tiles per XCD at a time (see gemm_tiled256_kernel)
  {
    constexpr int GM = 8;
    const int per = GM * p.tiles_n, grp = bid / per, first = grp * GM;
    const int gsz = (p.tiles_m - first) < GM ? (p.tiles_m - first) : GM;
    const int in = bid - grp * per;
    bm = first + in % gsz;
    bn = in / gsz;
  }
  const int z = blockIdx.y;
  int64_t a_off; int wz;
  if (!resolve_group(p, z, bm, a_off, wz)) return;
  const T* A = (const T*)p.A + a_off;
  const T* W = (const T*)p.W + (int64_t)wz * p.sW + p.wx;
  if (p.grp_ext && bn * TBN >= p.N) return;

  // LDS chunk c = tid + 256 i sits at (row c>>3, physical chunk c&7) and must hold logical chunk (c&7)^(row&7)
  const T* ga[4];
  const T* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 3, ch = (c & 7) ^ (row & 7);
    int ar = bm * TBM + row; ar = ar < p.M ? ar : p.M - 1;
    int wr = bn * TBN + row; wr = wr < p.N ? wr : p.N - 1;
    ga[i] = A + (int64_t)ar * p.lda + ch * VEC;
    gw[i] = W + (int64_t)wr * p.ldw + ch * VEC;
  }
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;  // this wave's 1 KiB piece inside a 4 KiB group

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = p.K / BK;
  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[i] + k0), (lds_ptr_t)(&smem[buf][0][i * 4096 + wave_lds]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(gw[i] + k0), (lds_ptr_t)(&smem[buf][1][i * 4096 + wave_lds]), 16, 0, 0);
    }
  };

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) issue(kt + 1, buf ^ 1);
    if constexpr (ASMLDS) {
      // both 64-byte k-steps of the slab are requested up front; the MFMAs of step 0 run under the reads of step 1,
      // and the DMA of slab kt+1 (issued above) runs under all of it
      const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(&smem[buf][0][0]);
      const uint32_t x0 = (uint32_t)((q ^ (r & 7)) << 4), x1 = (uint32_t)(((4 + q) ^ (r & 7)) << 4);
      const uint32_t ra = sb + (uint32_t)((wm * 64 + r) * TROWB), rb = sb + (uint32_t)(TBM * TROWB + (wn * 64 + r) * TROWB);
      u32x4_t a0[4], b0[4], a1[4], b1[4];
      SL_LDS_RD(a0[0], ra + x0, 0); SL_LDS_RD(a0[1], ra + x0, 2048); SL_LDS_RD(a0[2], ra + x0, 4096); SL_LDS_RD(a0[3], ra + x0, 6144);
      SL_LDS_RD(b0[0], rb + x0, 0); SL_LDS_RD(b0[1], rb + x0, 2048); SL_LDS_RD(b0[2], rb + x0, 4096); SL_LDS_RD(b0[3], rb + x0, 6144);
      SL_LDS_RD(a1[0], ra + x1, 0); SL_LDS_RD(a1[1], ra + x1, 2048); SL_LDS_RD(a1[2], ra + x1, 4096); SL_LDS_RD(a1[3], ra + x1, 6144);
      SL_LDS_RD(b1[0], rb + x1, 0); SL_LDS_RD(b1[1], rb + x1, 2048); SL_LDS_RD(b1[2], rb + x1, 4096); SL_LDS_RD(b1[3], rb + x1, 6144);
      lds_wait8<8>(a0[0], a0[1], a0[2], a0[3], b0[0], b0[1], b0[2], b0[3]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) MMA<T>::step(acc[m][n], as_uint4(a0[m]), as_uint4(b0[n]));
      __builtin_amdgcn_sched_barrier(0);
      lds_wait8<0>(a1[0], a1[1], a1[2], a1[3], b1[0], b1[1], b1[2], b1[3]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) MMA<T>::step(acc[m][n], as_uint4(a1[m]), as_uint4(b1[n]));
      __builtin_amdgcn_sched_barrier(0);   // keep the DMA wait and the barrier BELOW the MFMAs (they carry no data dependence)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      const unsigned char* sa = &smem[buf][0][0];
      const unsigned char* sw = &smem[buf][1][0];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint4 fa[4], fb[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[m] = *(const uint4*)(sa + lds_off(wm * 64 + m * 16 + r, s * 4 + q));
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[n] = *(const uint4*)(sw + lds_off(wn * 64 + n * 16 + r, s * 4 + q));
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) MMA<T>::step(acc[m][n], fa[m], fb[n]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }
  if constexpr (ACT != SL_ACT_SILU_MUL) {
    if (!p.direct_epi && tile_epilogue_rows<T, ACT, 4, ASMLDS>(p, acc, bm * TBM + wm * 64, bn * TBN + wn * 64, lane, z, wz, (float*)&smem[0][0][0] + wave * 4096)) return;
  }
  tile_epilogue<T, ACT>(p, acc, bm, bn, wm, wn, q, r, z, wz);
}

// ----------------------------------------------------------------------------------------------
// Both operands K-MAJOR (the weight-gradient product dW (No, Ni) += dY^T X with dY (tokens, No) and X (tokens, Ni) row-major, reduction
// over the token rows): 128 x 128 tile, LDS-DMA staging of [64 token rows][128 columns] slabs (256-byte rows), fragments by
// ds_read_b64_tr_b16 — a 16-lane group gathers 4 token rows x 16 columns and lane i receives column i's four values, i.e. four
// consecutive k of output row i; two reads make the 16-byte MFMA operand.  No transposed copies of dY / X are made (sl_transpose_pad
// read + wrote each of them once per product: 4 % of a KD window).  bf16 only.
//   LDS image: row = token row of the slab (256 B = eight 32-byte slots of 16 columns); slot s of row r sits at physical slot
//   s ^ f(r), f = (r & 3) | (((r >> 3) & 1) << 2): the 16 row segments one wave-wide read touches (rows 8g + {0..3} (+4), g = 0..3) fall on
//   every 32-byte bank group exactly twice — the rate of a 512-byte read.  As in the kernels above the swizzle is applied to the
//   per-lane SOURCE address of the DMA.
//   K (token) tail: rows past K are fetched from a 16-byte zero constant.  blockIdx.y = K run (split-K: fp32 partial tiles to
//   C + run * sC, summed by splitk_reduce_kernel), every run a whole number of slabs.
// ----------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

#define SL_LDS_RD_TR(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:" #off : "=v"(dst) : "v"(addr) : "memory")
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_tt_t;
template <int N>
__device__ __forceinline__ void lds_wait_tr16(u32x2_tt_t (&a)[8], u32x2_tt_t (&b)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%16)"
               : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]),
                 "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7])
               : "n"(N));
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void gemm_tiled_tt_kernel(GemmP p, int slabs_per_run) {
  using T = bf16_t;
  constexpr int BK = 64;                       // token rows per slab
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][BK * 256];   // [buf][A|W]: 64 rows x 256 B = 16 KiB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 15, q = lane >> 4;
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int qn = nt >> 3, rn = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  const int bm = bid % p.tiles_m, bn = bid / p.tiles_m;
  const int z = blockIdx.y;
  const int nkt_all = (p.K + BK - 1) / BK;
  const int kt0 = z * slabs_per_run;
  int kt1 = kt0 + slabs_per_run;
  kt1 = kt1 < nkt_all ? kt1 : nkt_all;
  const T* A = (const T*)p.A;
  const T* W = (const T*)p.W;

  // LDS chunk c = tid + 256 i sits at (row c >> 4, physical chunk c & 15) and must hold logical chunk ((pc >> 1) ^ f(row)) << 1 | (pc & 1)
  const T* ga[4];
  const T* gw[4];
  int grow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 256 * i, row = c >> 4, pc = c & 15;
    const int f = (row & 3) | (((row >> 3) & 1) << 2);
    const int lc = (((pc >> 1) ^ f) << 1) | (pc & 1);
    grow[i] = row;
    ga[i] = A + (int64_t)row * p.lda + bm * TBM + lc * 8;
    gw[i] = W + (int64_t)row * p.ldw + bn * TBN + lc * 8;
  }
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;
  const T* zero = (const T*)g_zero16;

  f32x4 acc[4][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool in = k0 + grow[i] < p.K;
      const T* sa = in ? ga[i] + (int64_t)k0 * p.lda : zero;
      const T* sw = in ? gw[i] + (int64_t)k0 * p.ldw : zero;
      __builtin_amdgcn_global_load_lds((glb_ptr_t)sa, (lds_ptr_t)(&smem[buf][0][i * 4096 + wave_lds]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)sw, (lds_ptr_t)(&smem[buf][1][i * 4096 + wave_lds]), 16, 0, 0);
    }
  };

  // fragment addresses: lane (li = r, g = q) asks for token row 8 g + (li >> 2) (+ 4 for the upper half, + 32 for the second k-step: immediates),
  // piece li & 3 of the 32-byte slot of its 16 columns; slot (4 wm + m) ^ f, f = (li >> 2) | ((g & 1) << 2)
  const int qq = r >> 2, pp = r & 3;
  const int f = qq | ((q & 1) << 2);
  uint32_t aa[4], ab[4];
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    aa[m] = (uint32_t)((8 * q + qq) * 256 + (((4 * wm + m) ^ f) << 5) + pp * 8);
    ab[m] = (uint32_t)((8 * q + qq) * 256 + (((4 * wn + m) ^ f) << 5) + pp * 8);
  }
  const uint32_t sb0 = (uint32_t)(uintptr_t)(lds_ptr_t)(&smem[0][0][0]);

  if (kt0 < kt1) {
    issue(kt0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  for (int kt = kt0; kt < kt1; ++kt) {
    const int buf = (kt - kt0) & 1;
    if (kt + 1 < kt1) issue(kt + 1, buf ^ 1);
    const uint32_t sa = sb0 + (uint32_t)buf * (2 * BK * 256), sw = sa + BK * 256;
    u32x2_tt_t a0[8], b0[8], a1[8], b1[8];
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(a0[2 * m], sa + aa[m], 0); SL_LDS_RD_TR(a0[2 * m + 1], sa + aa[m], 1024); }
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(b0[2 * m], sw + ab[m], 0); SL_LDS_RD_TR(b0[2 * m + 1], sw + ab[m], 1024); }
    lds_wait_tr16<0>(a0, b0);
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(a1[2 * m], sa + aa[m], 8192); SL_LDS_RD_TR(a1[2 * m + 1], sa + aa[m], 9216); }
#pragma unroll
    for (int m = 0; m < 4; ++m) { SL_LDS_RD_TR(b1[2 * m], sw + ab[m], 8192); SL_LDS_RD_TR(b1[2 * m + 1], sw + ab[m], 9216); }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        MMA<T>::step(acc[m][n], make_uint4(a0[2 * m].x, a0[2 * m].y, a0[2 * m + 1].x, a0[2 * m + 1].y), make_uint4(b0[2 * n].x, b0[2 * n].y, b0[2 * n + 1].x, b0[2 * n + 1].y));
    __builtin_amdgcn_sched_barrier(0);
    lds_wait_tr16<0>(a1, b1);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
        MMA<T>::step(acc[m][n], make_uint4(a1[2 * m].x, a1[2 * m].y, a1[2 * m + 1].x, a1[2 * m + 1].y), make_uint4(b1[2 * n].x, b1[2 * n].y, b1[2 * n + 1].x, b1[2 * n + 1].y));
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (!p.direct_epi && tile_epilogue_rows<T, ACT, 4>(p, acc, bm * TBM + wm * 64, bn * TBN + wn * 64, lane, z, 0, (float*)&smem[0][0][0] + wave * 4096)) return;
  tile_epilogue<T, ACT>(p, acc, bm, bn, wm, wn, q, r, z, 0);
}

// ----------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves (2 x 4, 128 x 64 each), same K slabs / swizzled LDS image / LDS-DMA staging as above.
// Why: measured, a CU sustains only ~40 GB/s of operand fetches (L2 hits + HBM through one miss queue) — the 128^2 tile
// needs 32 KiB per 2*128*128*64 FLOP and tops out at 600-980 TF/s on that, not on the MFMA pipe.  The 256^2 tile halves
// the bytes per FLOP; one block per CU (128 KiB of LDS), two waves per SIMD.  (A four-stage ring of 64-byte slabs with
// three slabs of DMA in flight — swizzle c ^ ((row >> 2) & 2) for conflict-free reads of 64-byte rows — measured 5-8 %
// SLOWER: the limit is the fetch rate per CU, not its latency; what helped is sharing slabs in L2, below.)
// Also measured slower (-4..-7 %): issuing the DMA of slab k+2 in the middle of slab k behind an extra bare barrier (1.25-1.5
// product phases of cover instead of one).  PMC on 17408x16384x3072: MFMA busy 46 %, waves 30 % parked (vmcnt/barrier), 50 %
// issue-stalled behind the MFMA pipe, 20 % issuing; no LDS bank conflicts.
// A 256 x 128 tile with dedicated loader waves (8 compute + 4 loader waves, three 48 KiB slots, fragment reads interleaved with
// the MFMAs — the structure of gemm_stream_wide_kernel on row-major operands) measured 0.93-1.09 x this kernel on the
// encoder / prefill shapes (tools/bench_gemm_lw.py, round 2): it is bound by what a CU pulls from L2 (~31 B/clk) at 48 KiB per
// 1024 MFMA-cycles, this tile needs 64 KiB per 2048; removed again.
// A persistent form (one block per CU walking its tiles, the next tile's first slab requested before the current tile's
// epilogue, which then turns 32-row groups through the other staging buffer) measured within +-2 % of this kernel at
// K = 1024..8192: the vmcnt(0) that admits the prefetched slab also drains the epilogue's stores (one counter on gfx9).
// Round 3, again with the register epilogue of the swapped-operand form (no LDS in the epilogue, the K slabs of consecutive tiles as
// one double-buffered stream, scalar tile bases): 0.97-1.02 x on the encoder shapes (tools/time_fold_epilogue.py); removed again.
// A second build with one epilogue form per instantiation (0-44 bytes of spills instead of 20-96): QKV 748-761 vs 770 us, out_proj
// 300 vs 306-309, FFN1 + GELU 1 134 vs 1 131, FFN2 equal, the LayerNorm-fold forms 3 % slower — the 3-6 us per tile that in-kernel
// stamps show between a block's last store and its successor's first product do not turn into throughput; removed again.
// Where the time goes (127744 x 4096 x 1024, bias + GELU, 1180 us): product loop alone 820-845 us (1.27-1.3 PF/s), epilogue
// arithmetic without its stores +45 us, the stores +170..290 us — 128 KiB per tile leave a CU at ~24 GB/s, and neither spreading
// the first-round blocks of an XCD over a tile time nor a block that outlives its tile changes that.
// ----------------------------------------------------------------------------------------------
constexpr int XBM = 256, XBN = 256;

template <typename T, int ACT, bool SW = false>
__global__ __launch_bounds__(512, 1) void gemm_tiled256_kernel(GemmP p) {
  static_assert(!SW || (sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL), "the swapped-operand form is the bf16 store epilogue");
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int BK = TROWB / (int)sizeof(T);
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][XBM * TROWB];   // [buf][A|W], 32 KiB each
  __shared__ float2 mr_s[XBM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  const int nt = p.tiles_m * p.tiles_n;
  int bid = blockIdx.x;
  {
    const int qn = nt >> 3, rn = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
  }
  // blocks that run together on an XCD (consecutive ids) cover an 8 x 4 patch of tiles, so they share A and W slabs in
  // that XCD's L2 (walking M only shares W: 33 slab streams per 32 blocks from beyond L2 instead of 12)
  int bm, bn;
  {
    const int GM = p.gm;
    const int per = GM * p.tiles_n, grp = bid / per, first = grp * GM;
    const int gsz = (p.tiles_m - first) < GM ? (p.tiles_m - first) : GM;
    const int in = bid - grp * per;
    bm = first + in % gsz;
    bn = in / gsz;
  }
  const int z = blockIdx.y;
  int64_t a_off; int wz;
  if (!resolve_group(p, z, bm, a_off, wz, XBM)) return;
  const T* A = (const T*)p.A + a_off;
  const T* W = (const T*)p.W + (int64_t)wz * p.sW + p.wx;
  if (p.grp_ext && bn * XBN >= p.N) return;

  // LDS chunk c = tid + 512 i sits at (row c>>3, physical chunk c&7) and must hold logical chunk (c&7)^(row&7)
  const T* ga[4];
  const T* gw[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = tid + 512 * i, row = c >> 3, ch = (c & 7) ^ (row & 7);
    // SW: the W fragments are read at rows 8 (r >> 2) + (r & 3) + {0, 4, 32, 36}; the image is swizzled by those rows' (r & 3) and
    // bit 0 of (r >> 2), which keeps each 16-lane group of a ds_read_b128 on 16 different 16-byte slots of the 256-byte bank row
    const int chw = SW ? (c & 7) ^ ((row & 3) | (((row >> 3) & 1) << 2)) : ch;
    int ar = bm * XBM + row; ar = ar < p.M ? ar : p.M - 1;
    int wr = bn * XBN + row; wr = wr < p.N ? wr : p.N - 1;
    ga[i] = A + (int64_t)ar * p.lda + ch * VEC;
    gw[i] = W + (int64_t)wr * p.ldw + chw * VEC;
  }
  const int wave_lds = __builtin_amdgcn_readfirstlane(wave) * 1024;  // this wave's 1 KiB piece inside an 8 KiB group

  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nkt = p.K / BK;
  auto issue = [&](int kt, int buf) {
    const int k0 = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(ga[i] + k0), (lds_ptr_t)(&smem[buf][0][i * 8192 + wave_lds]), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((glb_ptr_t)(gw[i] + k0), (lds_ptr_t)(&smem[buf][1][i * 8192 + wave_lds]), 16, 0, 0);
    }
  };

  issue(0, 0);
  if (sizeof(T) == 2 && p.ln_mr && tid < XBM) {   // LayerNorm fold: this tile's {mean, rstd} pairs wait in LDS for the epilogue
    int row = bm * XBM + tid;
    row = row < p.M ? row : p.M - 1;
    mr_s[tid] = ((const float2*)p.ln_mr)[row];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) issue(kt + 1, buf ^ 1);
    const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)(&smem[buf][0][0]);
    const uint32_t ra = sb + (uint32_t)((wm * 128 + r) * TROWB);
    const uint32_t rb = sb + (uint32_t)(XBM * TROWB + (wn * 64 + (SW ? 8 * (r >> 2) + (r & 3) : r)) * TROWB);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const uint32_t xs = (uint32_t)(((s * 4 + q) ^ (r & 7)) << 4);
      const uint32_t xw = SW ? (uint32_t)(((s * 4 + q) ^ ((r & 3) | (((r >> 2) & 1) << 2))) << 4) : xs;
      u32x4_t a[8], b[4];
      SL_LDS_RD(a[0], ra + xs, 0); SL_LDS_RD(a[1], ra + xs, 2048); SL_LDS_RD(a[2], ra + xs, 4096); SL_LDS_RD(a[3], ra + xs, 6144);
      if constexpr (SW) { SL_LDS_RD(b[0], rb + xw, 0); SL_LDS_RD(b[1], rb + xw, 512); SL_LDS_RD(b[2], rb + xw, 4096); SL_LDS_RD(b[3], rb + xw, 4608); }
      else { SL_LDS_RD(b[0], rb + xw, 0); SL_LDS_RD(b[1], rb + xw, 2048); SL_LDS_RD(b[2], rb + xw, 4096); SL_LDS_RD(b[3], rb + xw, 6144); }
      SL_LDS_RD(a[4], ra + xs, 8192); SL_LDS_RD(a[5], ra + xs, 10240); SL_LDS_RD(a[6], ra + xs, 12288); SL_LDS_RD(a[7], ra + xs, 14336);
      lds_wait8<4>(a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if constexpr (SW) MMA<T>::step(acc[m][n], as_uint4(b[n]), as_uint4(a[m]));
          else MMA<T>::step(acc[m][n], as_uint4(a[m]), as_uint4(b[n]));
        }
      __builtin_amdgcn_sched_barrier(0);
      lds_wait8<0>(a[4], a[5], a[6], a[7], b[0], b[1], b[2], b[3]);
#pragma unroll
      for (int m = 4; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          if constexpr (SW) MMA<T>::step(acc[m][n], as_uint4(b[n]), as_uint4(a[m]));
          else MMA<T>::step(acc[m][n], as_uint4(a[m]), as_uint4(b[n]));
        }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if constexpr (SW) {
    const int rb0 = bm * XBM + wm * 128, cb0 = bn * XBN + wn * 64;
    const float2* mrl = mr_s + wm * 128;
    const bool res = p.res != nullptr, ln = p.ln_mr != nullptr, st = p.stats_out != nullptr;    // launch_tiled admits these five forms only
    if (p.aux) tile_epilogue_sw<ACT, EPI_AUX>(p, acc, rb0, cb0, lane, z, wz, mrl);
    else if (ln) tile_epilogue_sw<ACT, EPI_LN>(p, acc, rb0, cb0, lane, z, wz, mrl);
    else if (st) tile_epilogue_sw<ACT, EPI_RES | EPI_STATS>(p, acc, rb0, cb0, lane, z, wz, mrl);
    else if (res) tile_epilogue_sw<ACT, EPI_RES>(p, acc, rb0, cb0, lane, z, wz, mrl);
    else tile_epilogue_sw<ACT, 0>(p, acc, rb0, cb0, lane, z, wz, mrl);
    return;
  } else {
    if constexpr (ACT != SL_ACT_SILU_MUL) {
      if (!p.direct_epi && tile_epilogue_rows<T, ACT, 8>(p, acc, bm * XBM + wm * 128, bn * XBN + wn * 64, lane, z, wz, (float*)&smem[0][0][0] + wave * 4096,
                                                        sizeof(T) == 2 && p.ln_mr ? mr_s + wm * 128 : nullptr)) return;
    }
    tile_epilogue_g<T, ACT, 8, 4>(p, acc, bm * XBM + wm * 128, bn * XBN + wn * 64, q, r, z, wz);
  }
}

// ----------------------------------------------------------------------------------------------
// 256 x 256 tile, staggered two-phase main loop (round 4).  Same tile, wave grid (2 x 4 waves of 128 x 64), fragment layouts and
// epilogues as gemm_tiled256_kernel; what changes is how a K slab of 64 bytes per row moves through the block:
//   * the slab is cut into four 16 KiB PIECES — PA0 (A rows of the waves' upper 64 x 64 halves), PB0 (W rows of the waves' left 32
//     columns), PB1 (right 32 columns), PA1 (lower halves) — and a wave's 128 x 64 output into an upper and a lower 64 x 64 half,
//     one per PHASE: the upper phase reads PA0 + PB0 + PB1 into registers (16 ds_read_b128), the lower one PA1 (8; W's fragments
//     are kept).  A phase = {fragment reads, DMA of the two pieces 6 and 7 pieces ahead, counted vmcnt, lgkmcnt(0), barrier,
//     32 MFMAs, barrier}: the DMA stays in flight across barriers (never vmcnt(0) in the steady loop), eight LDS slots.
//   * waves 4-7 (the lower 128 rows) run ONE BARRIER behind waves 0-3: while one wave of a SIMD issues its 32 MFMAs its partner
//     issues its reads and its DMA, so the matrix pipe of a SIMD always has a wave to draw from (in the one-barrier-per-slab loop
//     all eight waves read together and multiply together: MFMA busy 0.40-0.49).
// Ordering (guide §5 "Read a staged buffer one phase AFTER the wait that retires it"): phase P issues pieces 2P + 6 and 2P + 7 and
// waits until piece 2P + 4 has landed (vmcnt(6): three pieces stay in flight); both halves of the block have made that wait once
// the lagging half's first barrier of phase P is passed, and the pieces of phase P + 1 (4t .. 4t + 2 for an upper phase 2t) are
// read behind it.  A slot is re-filled (piece n + 8, phase (n >> 1) + 1) one phase after its last read; that is enough because every
// wave retires its fragment reads (lgkmcnt(0)) in FRONT of the phase's first barrier.  The lagging half's MFMAs of its last phase
// are still running when the leading half leaves the loop: that half passes one more barrier before the epilogue touches LDS.
// What was measured on the way (profiles/r04_b_*.txt, tools/gemm_knockout.py, tools/gemm_qvar.py — MI355X, random bf16):
//   four phases of 16 MFMAs (the first form of this loop)   1 311-1 365 TF/s on 140 288 x 5 120 x 3 072 (round-3 loop 1 212, vendor 1 396-1 440)
//   knock-outs of that form: MFMAs + barriers alone 2 154 us, reads + DMA + barriers alone 2 010 us (= 67 GB/s per CU, the rate a CU
//     pulls from L2 when every CU streams: a 256^2 tile needs 64 KiB per 2 048 MFMA cycles = 32 B/clk), both 3 461 us: the loop is
//     bound by how well two equal costs overlap, and the eight barriers of a slab cost the matrix pipe ~56 cycles each
//   reads rebalanced 8/4/8/4 (next slab's W fragments early), DMA lead 5 / 7 / 8 pieces, no s_setprio: -13 .. +3 %
//   two phases of 32 MFMAs (half the barriers)              +3-6 %;  and without s_setprio around the clusters  +1-3 % more  <- this loop
// ----------------------------------------------------------------------------------------------
template <int OFF>
__device__ __forceinline__ void lds_rd(u32x4_t& d, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void vm_wait_pieces(int n_out) {   // n_out pieces (two DMAs each per wave) may stay in flight; uniform
  switch (n_out) {
    case 0: vm_wait<0>(); break;
    case 1: vm_wait<2>(); break;
    case 2: vm_wait<4>(); break;
    default: vm_wait<6>(); break;
  }
}

// DBG (debug builds only, -DSL_GEMM_DEBUG): 8 = cycle stamps; knock-outs 1 = no fragment reads, 2 = no DMA, 4 = no MFMAs (results are then
// meaningless: timing experiments, tools/gemm_knockout.py)
//
// The main loop over `nkt` K slabs starting at element k_first, shared by the one-tile-per-block kernel and the stream-K kernel.
// gp[kind][i]: this thread's two source rows of piece kind {PA0, PB0, PB1, PA1} at k = 0; all eight waves call it together and
// leave it together (the leading half waits for the lagging one), with every DMA landed and every fragment read retired.
template <typename T, bool SW, int DBG>
__device__ __forceinline__ void t256_mainloop(unsigned char* smem, const T* const (&gp)[4][2], int64_t k_first, int nkt, int wave, int lane,
                                              f32x4 (&acc)[8][4], uint32_t* stamps) {
  constexpr bool STAMP = (DBG & 8) != 0, KO_RD = (DBG & 1) != 0, KO_DMA = (DBG & 2) != 0, KO_MMA = (DBG & 4) != 0;
  constexpr int BK = TROWB / (int)sizeof(T);
  constexpr int PIECE = 128 * TROWB;            // 16 KiB
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 15, q = lane >> 4;
  auto stamp = [&](int i) {
    if constexpr (STAMP) {
      if ((wave & 3) == 0 && lane == 0) stamps[(wave >> 2) * 32 + i] = (uint32_t)__builtin_amdgcn_s_memtime();
    }
  };
  const int NP = 4 * nkt;
  // piece n = 4 * slab + kind goes to slot n & 7
  auto issue = [&](int n, int kind) {
    if constexpr (KO_DMA) return;
    const int64_t k0 = k_first + (int64_t)(n >> 2) * BK;
    unsigned char* dst = smem + (n & 7) * PIECE + wave * 1024;
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(gp[kind][0] + k0), (lds_ptr_t)dst, 16, 0, 0);
    __builtin_amdgcn_global_load_lds((glb_ptr_t)(gp[kind][1] + k0), (lds_ptr_t)(dst + 8192), 16, 0, 0);
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int n = 0; n < 6; ++n)
    if (n < NP) issue(n, n & 3);
  {
    const int last = 5 < (NP - 1) ? 5 : (NP - 1);
    vm_wait_pieces(last - 2 > 0 ? last - 2 : 0);        // pieces 0, 1, 2 have landed
  }
  __builtin_amdgcn_s_barrier();
  if (wm == 1) __builtin_amdgcn_s_barrier();           // the lower half of the block runs one barrier behind

  const uint32_t sb = (uint32_t)(uintptr_t)(lds_ptr_t)smem;
  const uint32_t ra0 = sb + (uint32_t)((wm * 64 + r) * TROWB);
  const uint32_t rb0 = sb + (uint32_t)((wn * 32 + (SW ? 8 * (r >> 2) + (r & 3) : r)) * TROWB);
  const uint32_t ka = (uint32_t)(r & 7), kw = SW ? (uint32_t)((r & 3) | (((r >> 2) & 1) << 2)) : ka;
  const uint32_t xa0 = ((uint32_t)q ^ ka) << 4, xa1 = ((uint32_t)(4 + q) ^ ka) << 4;
  const uint32_t xw0 = ((uint32_t)q ^ kw) << 4, xw1 = ((uint32_t)(4 + q) ^ kw) << 4;
  constexpr int BN1 = SW ? 512 : 2048;               // second W fragment of a 32-column half: +4 rows (swapped form) / +16 rows

  u32x4_t a[8], b0[4], b1[4];                        // a[4 s + m'], b[2 s + n']
  if constexpr (KO_RD) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = u32x4_t{0x3f803f80u + lane, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}; asm volatile("" : "+v"(a[i])); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { b0[i] = a[i]; b1[i] = a[4 + i]; asm volatile("" : "+v"(b0[i]), "+v"(b1[i])); }
  }

  auto rd_a = [&](uint32_t base) {                   // base = address of the piece's row (wm * 64 + r)
    if constexpr (KO_RD) return;
    lds_rd<0>(a[0], base + xa0); lds_rd<2048>(a[1], base + xa0); lds_rd<4096>(a[2], base + xa0); lds_rd<6144>(a[3], base + xa0);
    lds_rd<0>(a[4], base + xa1); lds_rd<2048>(a[5], base + xa1); lds_rd<4096>(a[6], base + xa1); lds_rd<6144>(a[7], base + xa1);
  };
  auto rd_b = [&](u32x4_t (&b)[4], uint32_t base) {
    if constexpr (KO_RD) return;
    lds_rd<0>(b[0], base + xw0); lds_rd<BN1>(b[1], base + xw0);
    lds_rd<0>(b[2], base + xw1); lds_rd<BN1>(b[3], base + xw1);
  };
  auto mma_q = [&](int mi, int nj, u32x4_t (&b)[4]) {     // quadrant (mi, nj): 16 MFMAs, both 64-byte k-steps
    if constexpr (KO_MMA) { asm volatile("" : "+v"(a[0]), "+v"(b[0])); return; }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          if constexpr (SW) MMA<T>::step(acc[mi * 4 + m][nj * 2 + n], as_uint4(b[2 * s + n]), as_uint4(a[4 * s + m]));
          else MMA<T>::step(acc[mi * 4 + m][nj * 2 + n], as_uint4(a[4 * s + m]), as_uint4(b[2 * s + n]));
        }
  };
  // the rest of a phase's load segment: DMA of pieces 2P + 6 and 2P + 7, the wait that retires piece 2P + 4, the fragment reads
  // retired (every fragment register tied to the wait so no MFMA moves above it), the barrier
  auto stage = [&](int P, bool steady) {
    const int n0 = 2 * P + 6;
    if (steady) {
      issue(n0, n0 & 3); issue(n0 + 1, (n0 + 1) & 3);
      vm_wait<6>();
    } else {
      if (n0 < NP) issue(n0, n0 & 3);
      if (n0 + 1 < NP) issue(n0 + 1, (n0 + 1) & 3);
      const int last = (n0 + 1) < (NP - 1) ? (n0 + 1) : (NP - 1);
      vm_wait_pieces(last - (2 * P + 4) > 0 ? last - (2 * P + 4) : 0);
    }
    lds_wait8<0>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7]);
    lds_wait8<0>(b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  };
  auto close = [&]() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); };
  // stamps (instrumented build): slabs 4 and 5, per phase {load segment start, first barrier passed, MFMAs issued} -> entries 2 .. 13
  auto tile = [&](int t, bool steady) {
    const uint32_t bo = (uint32_t)(t & 1) * (4 * PIECE);
    const bool st = STAMP && (t == 4 || t == 5);
    const int sb_ = 2 + (t - 4) * 6;
    if (st) stamp(sb_ + 0);
    rd_b(b0, rb0 + bo + 1 * PIECE);
    rd_a(ra0 + bo + 0 * PIECE);
    rd_b(b1, rb0 + bo + 2 * PIECE);
    stage(2 * t, steady);
    if (st) stamp(sb_ + 1);
    mma_q(0, 0, b0);
    mma_q(0, 1, b1);
    if (st) stamp(sb_ + 2);
    close();
    if (st) stamp(sb_ + 3);
    rd_a(ra0 + bo + 3 * PIECE);
    stage(2 * t + 1, steady);
    if (st) stamp(sb_ + 4);
    mma_q(1, 1, b1);
    mma_q(1, 0, b0);
    if (st) stamp(sb_ + 5);
    close();
  };
  const int nsteady = nkt - 2;                         // slabs whose two phases both issue: 2 (2 t + 1) + 7 <= NP - 1
  int t = 0;
  stamp(1);
  for (; t < nsteady; ++t) tile(t, true);
  for (; t < nkt; ++t) tile(t, false);
  stamp(14);
  if (wm == 0) __builtin_amdgcn_s_barrier();           // the lagging half's last MFMA segment ends behind this one
}

// staging pointers of tile (bm, bn): LDS chunk c = tid + 512 i of a piece sits at (piece row c >> 3, physical chunk c & 7) and holds the
// logical chunk (c & 7) ^ key(piece row).  Piece rows: A pieces = [half of the block 0/1][64 rows], W pieces = [wave column 0..3][32 rows].
template <typename T, bool SW>
__device__ __forceinline__ void t256_stage_ptrs(const GemmP& p, const T* A, const T* W, int bm, int bn, int tid, const T* (&gp)[4][2]) {
  constexpr int VEC = Vec16<T>::VEC;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + 512 * i, rho = c >> 3, pc = c & 7;
    const int cha = pc ^ (rho & 7);
    const int chw = SW ? pc ^ ((rho & 3) | (((rho >> 3) & 1) << 2)) : cha;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int ar = bm * XBM + (rho >> 6) * 128 + h * 64 + (rho & 63); ar = ar < p.M ? ar : p.M - 1;
      int wr = bn * XBN + (rho >> 5) * 64 + h * 32 + (rho & 31); wr = wr < p.N ? wr : p.N - 1;
      gp[h ? 3 : 0][i] = A + (int64_t)ar * p.lda + cha * VEC;
      gp[h ? 2 : 1][i] = W + (int64_t)wr * p.ldw + chw * VEC;
    }
  }
}

// tile index (after the XCD remap) -> tile coordinates: XCD patches of GM tile rows (see gemm_tiled256_kernel)
__device__ __forceinline__ void t256_tile_coords(const GemmP& p, int v, int& bm, int& bn) {
  const int GM = p.gm;
  const int per = GM * p.tiles_n, grp = v / per, first = grp * GM;
  const int gsz = (p.tiles_m - first) < GM ? (p.tiles_m - first) : GM;
  const int in = v - grp * per;
  bm = first + in % gsz;
  bn = in / gsz;
}
__device__ __forceinline__ int xcd_remap(int bid, int n) {      // bijective: the blocks of one XCD (bid % 8 equal) get consecutive indices
  const int qn = n >> 3, rn = n & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + idx;
}

template <typename T, int ACT, bool SW, bool POSTS = false>
__device__ __forceinline__ void t256_epilogue(const GemmP& p, f32x4 (&acc)[8][4], int bm, int bn, int wave, int lane, int z, int wz, unsigned char* smem,
                                              float2* mr_s) {
  const int wm = wave >> 2, wn = wave & 3;
  if constexpr (SW) {
    const int rb0_ = bm * XBM + wm * 128, cb0 = bn * XBN + wn * 64;
    const float2* mrl = mr_s + wm * 128;
    const bool res = p.res != nullptr, ln = p.ln_mr != nullptr, st = p.stats_out != nullptr;    // launch_tiled admits these five forms only
    bool done = false;
    if constexpr (POSTS) {
     if (p.post) {         // launch_tiled admits exactly these post-op forms on the swapped-operand kernels (the phased kernel only)
      done = true;
      if constexpr (ACT == SL_ACT_GELU) {
        tile_epilogue_sw<ACT, EPI_AUX | EPI_DROP>(p, acc, rb0_, cb0, lane, z, wz, mrl);        // FFN1 forward: mid = dropout(gelu(pre)), pre kept
      } else {
        if (p.post == SL_POST_DROPOUT) tile_epilogue_sw<ACT, EPI_RES | EPI_DROP>(p, acc, rb0_, cb0, lane, z, wz, mrl);   // h = residual + dropout(sublayer)
        else if (p.post == SL_POST_GELU_BWD) tile_epilogue_sw<ACT, EPI_GBWD>(p, acc, rb0_, cb0, lane, z, wz, mrl);
        else tile_epilogue_sw<ACT, EPI_SBWD>(p, acc, rb0_, cb0, lane, z, wz, mrl);
      }
     }
    }
    if (done) return;
    if (p.aux) tile_epilogue_sw<ACT, EPI_AUX>(p, acc, rb0_, cb0, lane, z, wz, mrl);
    else if (ln) tile_epilogue_sw<ACT, EPI_LN>(p, acc, rb0_, cb0, lane, z, wz, mrl);
    else if (st) tile_epilogue_sw<ACT, EPI_RES | EPI_STATS>(p, acc, rb0_, cb0, lane, z, wz, mrl);
    else if (res) tile_epilogue_sw<ACT, EPI_RES>(p, acc, rb0_, cb0, lane, z, wz, mrl);
    else tile_epilogue_sw<ACT, 0>(p, acc, rb0_, cb0, lane, z, wz, mrl);
  } else {
    if constexpr (ACT != SL_ACT_SILU_MUL) {
      // the LDS-turned rows epilogue uses 16 KiB per wave of the piece slots (every DMA has landed: the last phases wait vmcnt(0))
      if (!p.direct_epi && tile_epilogue_rows<T, ACT, 8, POSTS>(p, acc, bm * XBM + wm * 128, bn * XBN + wn * 64, lane, z, wz, (float*)smem + wave * 4096,
                                                        sizeof(T) == 2 && p.ln_mr ? mr_s + wm * 128 : nullptr)) return;
    }
    tile_epilogue_g<T, ACT, 8, 4>(p, acc, bm * XBM + wm * 128, bn * XBN + wn * 64, lane >> 4, lane & 15, z, wz);
  }
}

template <typename T, int ACT, bool SW = false, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_tiled256p_kernel(GemmP p) {
  constexpr bool STAMP = (DBG & 8) != 0;
  static_assert(!SW || (sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL), "the swapped-operand form is the bf16 store epilogue");
  constexpr int BK = TROWB / (int)sizeof(T);
  constexpr int PIECE = 128 * TROWB;            // 16 KiB
  // one LDS object (a second one beside an LDS-DMA target can cost a vmcnt(0) per k-step, guide §5 item 4a): 8 piece slots + {mean, rstd}
  __shared__ __attribute__((aligned(16))) unsigned char smem[8 * PIECE + XBM * 8];
  float2* mr_s = (float2*)(smem + 8 * PIECE);   // instrumented build: the stamps of lane 0 of waves 0 and 4 live here (no fold in that build)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if constexpr (STAMP) {
    if ((wave & 3) == 0 && lane == 0) ((uint32_t*)mr_s)[(wave >> 2) * 32] = (uint32_t)__builtin_amdgcn_s_memtime();
  }
  int bm, bn;
  t256_tile_coords(p, xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n), bm, bn);
  const int z = blockIdx.y;
  int64_t a_off; int wz;
  if (!resolve_group(p, z, bm, a_off, wz, XBM)) return;
  const T* A = (const T*)p.A + a_off;
  const T* W = (const T*)p.W + (int64_t)wz * p.sW + p.wx;
  if (p.grp_ext && bn * XBN >= p.N) return;
  const T* gp[4][2];     // [PA0, PB0, PB1, PA1][i]
  t256_stage_ptrs<T, SW>(p, A, W, bm, bn, tid, gp);

  f32x4 acc[8][4];
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (sizeof(T) == 2 && p.ln_mr && tid < XBM) {   // LayerNorm fold: this tile's {mean, rstd} pairs wait in LDS for the epilogue
    int row = bm * XBM + tid;                     // (before the first DMA: the compiler drains vmcnt for this load's use)
    row = row < p.M ? row : p.M - 1;
    mr_s[tid] = ((const float2*)p.ln_mr)[row];
  }
  t256_mainloop<T, SW, DBG>(smem, gp, 0, p.K / BK, wave, lane, acc, (uint32_t*)mr_s);
  if constexpr (STAMP) {
    if ((wave & 3) == 0 && lane < 32 && p.stamp) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      p.stamp[((int64_t)(blockIdx.y * gridDim.x + blockIdx.x) * 2 + (wave >> 2)) * 32 + lane] = ((const uint32_t*)mr_s)[(wave >> 2) * 32 + lane];
    }
  }
  t256_epilogue<T, ACT, SW, DBG == 0>(p, acc, bm, bn, wave, lane, z, wz, smem, mr_s);
}

// ----------------------------------------------------------------------------------------------
// Stream-K form of the kernel above (un-grouped bf16 / fp32 products whose 256^2 tiles do not fill the chip evenly: KD windows of
// 2-8 k rows, the per-rank KD regime of a few hundred rows, weight gradients of 16-64 tiles under K = 8 000, decode projections).
// The tiles' K slabs form one sequence of tiles x slabs units, cut into `gridDim.x` equal contiguous ranges, one per block (one
// block per CU); a block walks its range from the top down, tile segment by tile segment, each segment through the main loop above.
//   * a segment that is a whole tile: the usual epilogue;
//   * a segment that does not reach its tile's last slab (only a block's FIRST segment can be one): the accumulators go to the
//     block's slot of the workspace as fp32 (16-byte write-through stores), every wave drains, one lane raises the block's flag;
//   * a segment that ends its tile but does not start it (only a block's LAST segment): the tile's other segments belong to the
//     blocks just below, which produced them first thing — the owner polls their flags (one lane, relaxed, s_sleep), takes ONE
//     agent-scope acquire, adds the partial sums in descending block order (a fixed order: results are reproducible, though not
//     bit-identical to the unsplit kernel), clears the flags and runs the epilogue.
// Waiting is only ever for work that was started before the waiter's own: no cycle, and with at most one block per CU resident
// (130 KiB of LDS) every block of a grid of <= #CUs blocks is resident or becomes resident as soon as any kernel's block retires.
// Workspace (caller-owned, zero-initialised once): [flags: 1 KiB][gridDim.x slots of 256 KiB].  Guide §6 Guideline 16 (R1).
// ----------------------------------------------------------------------------------------------
constexpr size_t SK_FLAG_BYTES = 1024, SK_SLOT_BYTES = (size_t)XBM * XBN * 4;

template <typename T, int ACT, bool SW>
__global__ __launch_bounds__(512, 2) void gemm_tiled256sk_kernel(GemmP p, unsigned char* ws) {
  static_assert(!SW || (sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL), "the swapped-operand form is the bf16 store epilogue");
  constexpr int BK = TROWB / (int)sizeof(T);
  constexpr int PIECE = 128 * TROWB;
  __shared__ __attribute__((aligned(16))) unsigned char smem[8 * PIECE + XBM * 8];
  float2* mr_s = (float2*)(smem + 8 * PIECE);
  typedef __attribute__((address_space(1))) unsigned int gu32;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = gridDim.x, nkt = p.K / BK, nt = p.tiles_m * p.tiles_n;
  const int vb = xcd_remap(blockIdx.x, G);                 // blocks of one XCD take consecutive ranges: their tiles share A / W slabs in its L2
  const int64_t U = (int64_t)nt * nkt;
  const int64_t u_lo = U * vb / G;
  int64_t u_hi = U * (vb + 1) / G;
  gu32* flags = (gu32*)ws;
  float* slots = (float*)(ws + SK_FLAG_BYTES);
  const T* A = (const T*)p.A;
  const T* W = (const T*)p.W;
  bool first = true;
  while (u_hi > u_lo) {
    const int tile = (int)((u_hi - 1) / nkt);
    const int64_t t0 = (int64_t)tile * nkt;
    const int s1 = (int)(u_hi - t0), s0 = (int)((u_lo > t0 ? u_lo : t0) - t0);
    int bm, bn;
    t256_tile_coords(p, tile, bm, bn);               // unit order = the XCD-patch tile order: the blocks of one XCD (consecutive vb) work on neighbouring tiles
    const T* gp[4][2];
    t256_stage_ptrs<T, SW>(p, A, W, bm, bn, tid, gp);
    f32x4 acc[8][4];
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!first) __builtin_amdgcn_s_barrier();              // the previous segment's epilogue may still be turning rows through LDS in another wave
    first = false;
    t256_mainloop<T, SW, 0>(smem, gp, (int64_t)s0 * BK, s1 - s0, wave, lane, acc, nullptr);
    if (s1 < nkt) {
      // partial sums -> this block's slot, in register order: [wave][fragment][lane] x 16 bytes, write-through (sc1)
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(slots + (int64_t)vb * (XBM * XBN), 0, (int)SK_SLOT_BYTES, 0x00020000);
      const int off = (wave * 32 * 64 + lane) * 16;
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[m][n]), rs, off + (m * 4 + n) * 1024, 0, 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // EVERY storing wave drains, then the workgroup's barrier, then ONE flag store
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flags + vb, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (s0 > 0) {
        // owner: the rest of this tile sits in the slots of the blocks below, down to the one that holds the tile's first slab
        int v_first = vb - 1;
        while (U * v_first / G > t0) --v_first;
        if (wave == 0) {
          for (int v = vb - 1; v >= v_first; --v) {
            if (lane == 0) {
              while (__hip_atomic_load(flags + v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(8);
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        for (int v = vb - 1; v >= v_first; --v) {
          const f32x4* src = (const f32x4*)(slots + (int64_t)v * (XBM * XBN)) + wave * 32 * 64 + lane;
#pragma unroll
          for (int m = 0; m < 8; m += 2) {       // eight fragments (32 registers) in flight at a time: all 32 at once would need 128, fewer leaves the
            f32x4 t8[2][4];                        // read latency-bound (guide: >= 8 loads per lane outstanding on a handed-off tile)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int n = 0; n < 4; ++n) t8[h][n] = __builtin_nontemporal_load(src + ((m + h) * 4 + n) * 64);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
              for (int n = 0; n < 4; ++n) acc[m + h][n] += t8[h][n];
            __builtin_amdgcn_sched_barrier(0);
          }
        }
        if (tid < vb - v_first) __hip_atomic_store(flags + (vb - 1 - tid), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed: ready for the next launch
      }
      t256_epilogue<T, ACT, SW>(p, acc, bm, bn, wave, lane, 0, 0, smem, mr_s);
    }
    u_hi = t0 + s0;
  }
}

// ----------------------------------------------------------------------------------------------
// skinny kernel (M <= 16*MT): HBM-bound weight streaming for decode.
//   block = RF 16-row weight fragments x NW waves; wave w takes 64-byte k-steps w, w+NW, ... with U steps
//   of loads in flight; W fragments go global -> VGPR -> MFMA (A operand), the M activation rows are the
//   B operand (re-read from L2); partial sums meet once in LDS.
//   PACKED: W is stored fragment-major [ceil(N/16)][K/KSTEP][64 lanes][16 B] (sl_pack_weight layout), so
//   every wave-level weight load is one contiguous 1 KiB (measured +25..45 % over row-major 16 x 64 B).
//   Epilogues: NONE/GELU (+bias, +residual), SILU_MUL (fragment pairs = gate, up), ROPE_KV (fragment
//   pairs = the two rotate_half halves of a head; q is written rotated, k/v go straight into the cache).
//   fuse_rms: the RMSNorm gain is pre-folded into W and the per-row rsqrt(mean(x^2)+eps) is computed
//   from the x fragments the block loads anyway, then applied to the accumulators.
// ----------------------------------------------------------------------------------------------

template <typename T, int MT, int ACT, int RF, int NW, int U, bool PACKED, bool KCONT = false>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmP p, SkinnyX sx) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int KSTEP = MMA<T>::KSTEP;
  constexpr int RB = 16 * RF;
  constexpr bool PAIRS = (ACT == SL_ACT_SILU_MUL || ACT == SL_ACT_ROPE_KV);
  static_assert(!PAIRS || RF % 2 == 0, "pair epilogues need an even number of fragments");
  __shared__ float red[NW][RB][MT * 16 + 1];
  __shared__ float red_ss[NW][4][MT * 16];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int z = blockIdx.y;
  const int n0 = blockIdx.x * RB;
  const int nks_full = p.K / KSTEP;  // full 64-byte steps

  const T* A = (const T*)p.A + (int64_t)z * p.sA;
  const T* W = (const T*)p.W + (int64_t)z * p.sW;

  const T* wp[RF];
  int64_t wstep;  // elements between consecutive k-steps of one fragment row
  if constexpr (PACKED) {
    const int nfrag = (p.N + 15) >> 4;
    wstep = 64 * VEC;
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      int fi = (n0 >> 4) + f; fi = fi < nfrag ? fi : nfrag - 1;
      wp[f] = W + (int64_t)fi * nks_full * (64 * VEC) + lane * VEC;
    }
  } else {
    wstep = KSTEP;
#pragma unroll
    for (int f = 0; f < RF; ++f) {
      int wr = n0 + f * 16 + r; wr = wr < p.N ? wr : p.N - 1;
      wp[f] = W + (int64_t)wr * p.ldw + q * VEC;
    }
  }
  const T* xp[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    int xr = t * 16 + r; xr = xr < p.M ? xr : p.M - 1;
    xp[t] = A + (int64_t)xr * p.lda + q * VEC;
  }

  f32x4 acc[RF][MT];
  float ss[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    ss[t] = 0.f;
#pragma unroll
    for (int f = 0; f < RF; ++f) acc[f][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  const bool fuse = sx.fuse_rms != 0;

  auto sumsq = [&](const uint4& u, float& s) {
    float e[VEC];
    Vec16<T>::unpack(u, e);
#pragma unroll
    for (int j = 0; j < VEC; ++j) s = fmaf(e[j], e[j], s);
  };

  // K split over the block's waves: interleaved 64-byte steps (w, w+NW, ...) or one contiguous slice per wave
  int ks, kend;
  constexpr int KSTR = KCONT ? 1 : NW;
  if constexpr (KCONT) {
    const int per = (nks_full + NW - 1) / NW;
    ks = wave * per;
    kend = (ks + per) < nks_full ? (ks + per) : nks_full;
  } else {
    ks = wave;
    kend = nks_full;
  }
  for (; ks + (U - 1) * KSTR < kend; ks += U * KSTR) {
    uint4 fw[U][RF], fx[U][MT];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t kk = ks + u * KSTR;
#pragma unroll
      for (int f = 0; f < RF; ++f) fw[u][f] = ld_nt16(wp[f] + kk * wstep);
#pragma unroll
      for (int t = 0; t < MT; ++t) fx[u][t] = *(const uint4*)(xp[t] + kk * KSTEP);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int t = 0; t < MT; ++t) MMA<T>::step(acc[f][t], fw[u][f], fx[u][t]);
      if (fuse) {
#pragma unroll
        for (int t = 0; t < MT; ++t) sumsq(fx[u][t], ss[t]);
      }
    }
  }
  for (; ks < kend; ks += KSTR) {
    uint4 fw[RF], fx[MT];
#pragma unroll
    for (int f = 0; f < RF; ++f) fw[f] = ld_nt16(wp[f] + (int64_t)ks * wstep);
#pragma unroll
    for (int t = 0; t < MT; ++t) fx[t] = *(const uint4*)(xp[t] + (int64_t)ks * KSTEP);
#pragma unroll
    for (int f = 0; f < RF; ++f)
#pragma unroll
      for (int t = 0; t < MT; ++t) MMA<T>::step(acc[f][t], fw[f], fx[t]);
    if (fuse) {
#pragma unroll
      for (int t = 0; t < MT; ++t) sumsq(fx[t], ss[t]);
    }
  }
  if constexpr (!PACKED) {
    // K tail (K % KSTEP != 0): one predicated step, taken by the wave whose turn it is
    if (nks_full * KSTEP < p.K && wave == (KCONT ? NW - 1 : nks_full % NW)) {
      const int64_t k = (int64_t)nks_full * KSTEP;
      const bool ok = (k + q * VEC) < p.K;
      uint4 fw[RF], fx[MT];
#pragma unroll
      for (int f = 0; f < RF; ++f) fw[f] = ok ? *(const uint4*)(wp[f] + k) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int t = 0; t < MT; ++t) fx[t] = ok ? *(const uint4*)(xp[t] + k) : make_uint4(0, 0, 0, 0);
#pragma unroll
      for (int f = 0; f < RF; ++f)
#pragma unroll
        for (int t = 0; t < MT; ++t) MMA<T>::step(acc[f][t], fw[f], fx[t]);
      if (fuse) {
#pragma unroll
        for (int t = 0; t < MT; ++t) sumsq(fx[t], ss[t]);
      }
    }
  }

  // D[weight row 4q+i][x row r]  ->  red[wave][f*16 + 4q+i][t*16 + r]
#pragma unroll
  for (int f = 0; f < RF; ++f)
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) red[wave][f * 16 + q * 4 + i][t * 16 + r] = acc[f][t][i];
  if (fuse) {
#pragma unroll
    for (int t = 0; t < MT; ++t) red_ss[wave][q][t * 16 + r] = ss[t];
  }
  __syncthreads();

  void* Cb = p.out_f32 ? (void*)((float*)p.C + (int64_t)z * p.sC) : (void*)((T*)p.C + (int64_t)z * p.sC);
  const T* bias = p.bias ? (const T*)p.bias + (int64_t)z * p.sBias : nullptr;
  const void* Rb = p.res ? (const void*)((const T*)p.res + (int64_t)z * p.sR) : nullptr;
  auto row_scale = [&](int m) -> float {
    if (!fuse) return 1.0f;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red_ss[w][0][m] + red_ss[w][1][m] + red_ss[w][2][m] + red_ss[w][3][m];
    return rsqrtf(s / (float)p.K + sx.eps);
  };
  auto rsum = [&](int n, int m) -> float {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += red[w][n][m];
    return v;
  };

  if constexpr (PAIRS) {
    constexpr int NP = RF / 2;
    for (int o = tid; o < NP * 16 * p.M; o += NW * 64) {
      const int n = o & 15, pr = (o >> 4) % NP, m = o / (16 * NP);
      const float rs = row_scale(m);
      float a = rsum((2 * pr) * 16 + n, m) * rs, b = rsum((2 * pr + 1) * 16 + n, m) * rs;
      if constexpr (ACT == SL_ACT_SILU_MUL) {
        const int ocol = ((n0 >> 4) / 2 + pr) * 16 + n;
        if (ocol >= (p.N >> 1)) continue;
        if (bias) { a += to_f32(bias[n0 + (2 * pr) * 16 + n]); b += to_f32(bias[n0 + (2 * pr + 1) * 16 + n]); }
        store_out<T>(p, Cb, Rb, m, ocol, silu(a) * b);
      } else {  // ROPE_KV: global fragment gf; a head is 8 fragments (D = 128)
        const int gf = (n0 >> 4) + 2 * pr;
        if (gf * 16 >= p.N) continue;
        const int hh = gf >> 3, j = (gf & 7) >> 1;
        const int pos = sx.pos[m];
        if (hh < sx.nh + sx.nkv) {
          const int d = j * 16 + n;  // a = x[d], b = x[d + 64]
          const float c = sx.cos[(int64_t)pos * 64 + d], s = sx.sin[(int64_t)pos * 64 + d];
          const float o1 = a * c - b * s, o2 = b * c + a * s;
          if (hh < sx.nh) {
            T* dst = (T*)Cb + (int64_t)m * p.ldc + hh * 128 + d;
            dst[0] = from_f32<T>(o1); dst[64] = from_f32<T>(o2);
          } else {
            T* dst = (T*)sx.kc + (((int64_t)sx.seq[m] * sx.nkv + (hh - sx.nh)) * sx.max_ctx + pos) * 128 + d;
            dst[0] = from_f32<T>(o1); dst[64] = from_f32<T>(o2);
          }
        } else {  // v rows are in natural order: fragments 2j, 2j+1
          const int d = (gf & 7) * 16 + n;
          T* dst = (T*)sx.vc + (((int64_t)sx.seq[m] * sx.nkv + (hh - sx.nh - sx.nkv)) * sx.max_ctx + pos) * 128 + d;
          dst[0] = from_f32<T>(a); dst[16] = from_f32<T>(b);
        }
      }
    }
  } else {
    for (int o = tid; o < RB * p.M; o += NW * 64) {
      const int n = o % RB, m = o / RB;
      const int col = n0 + n;
      if (col >= p.N) continue;
      float v = rsum(n, m) * row_scale(m);
      if (bias) v += to_f32(bias[col]);
      if constexpr (ACT == SL_ACT_GELU) v = gelu_act<T>(v);
      store_out<T>(p, Cb, Rb, m, col, v);
    }
  }
}

// ----------------------------------------------------------------------------------------------
// host dispatch
// ----------------------------------------------------------------------------------------------
// ----------------------------------------------------------------------------------------------
// stream-K admission (sl_gemm_ex_args.sk_ws): returns the grid size, 0 = keep one block per tile
// ----------------------------------------------------------------------------------------------
static int sk_cu_count() {
  static int cus[SL_MAX_DEVICES] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SL_MAX_DEVICES) return 256;
  if (!cus[dev]) {
    int n = 0;
    cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cus[dev];
}
extern "C" size_t sl_gemm_streamk_workspace_bytes(void) { return SK_FLAG_BYTES + 256 * SK_SLOT_BYTES; }

static int sk_grid(const GemmP& p, int batch, int bk, size_t ws_bytes) {
  const int mode = sl_env().stream_k;          // SL_STREAM_K: 0 = never, 1 = rule (default), 2 = whenever the form allows
  if (!mode || p.ta || p.tw || p.grp || batch != 1 || p.K % bk || p.ln_mr || p.stats_out || p.amax_val || p.aux || p.N < 192 || p.post || p.colsum) return 0;
  const int cus = sk_cu_count() < 256 ? sk_cu_count() : 256;
  const int64_t tm = (p.M + XBM - 1) / XBM, tn = (p.N + XBN - 1) / XBN, nt = tm * tn, nkt = p.K / bk;
  if (nkt < 8 || tm * XBM * 4 > (int64_t)p.M * 5 + 4 * XBM) return 0;      // short reductions; rows padded by more than a quarter (+ one tile)
  int G = cus;
  if (nt * 4 < G) G = (int)(nt * 4);                                        // a tile's slabs go to at most ~4 blocks: the owner reads the others' sums serially
  if (nt * nkt < (int64_t)G * 4) return 0;
  if (ws_bytes < SK_FLAG_BYTES + (size_t)G * SK_SLOT_BYTES) return 0;
  if (mode == 2) return G;
  // Measured (profiles/r04_f_streamk.txt): cutting K across blocks gives up what the XCD tile patches buy — neighbouring blocks no longer
  // stream the SAME slabs, so every block pulls its 64 KiB per slab from beyond L2 (3 200 x 3 072 x 16 384: 156 tiles, all 256 CUs busy,
  // 2.6 GB through the Infinity Cache in 310 us = its bandwidth; 0.94 x the one-block-per-tile time) — and a tile's fp32 hand-off costs
  // 10-20 us.  It pays only where a few tiles sit under a long reduction (400 x 3 072 x 16 384: 24 tiles, 1.6 x).
  return (nt * 8 <= cus && nkt >= 128) ? G : 0;
}

// ----------------------------------------------------------------------------------------------
// Plain split-K for products of FEW tiles (round 5): the per-rank KD window (634 LLM rows) has N = 3 072 products of 36 big / 120
// small tiles under K = 3 072 ... 16 384 — a third of the chip busy (634 x 3 072 x 8 192: 425 TF/s against the vendor library's 729,
// x 16 384: 294 against 672; profiles/r05_b_gemm_vs_vendor_pad_rule.txt).  The reduction is cut into S equal runs of whole slabs and
// the S partial products run as ONE BATCHED launch of the ordinary tiled kernels (batch index = K run: A and W advance by K / S
// columns, fp32 partial tiles go to the caller's workspace), so blocks of a run still share their slabs in L2 — which the stream-K form
// above gives up — and a second launch adds the runs in run order and applies bias / residual / rounding: deterministic, no atomics.
// Costs S x M x N x 4 bytes written and read back, which is why it is for few tiles only.
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int S, int64_t slab, GemmP p) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i >= (int64_t)p.M * p.N) return;
  const int64_t row = i / p.N;
  const int col = (int)(i - row * p.N);          // N % 4 == 0 (admission): the four values are one row's
  f32x4 acc = *(const f32x4*)(part + i);
  for (int z = 1; z < S; ++z) {
    const f32x4 v = *(const f32x4*)(part + (int64_t)z * slab + i);
    acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3];
  }
  float v4[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) v4[e] = acc[e] + (p.bias ? to_f32(((const T*)p.bias)[col + e]) : 0.f);
  if (p.post == SL_POST_SILU_MUL_BWD) {
    const int64_t o = row * p.post_ld + 32 * (col >> 4) + (col & 15);
    T* op = (T*)p.C + row * p.ldc + 32 * (col >> 4) + (col & 15);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float dg, du;
      post_silu_bwd<T>(v4[e], to_f32(((const T*)p.post_in)[o + e]), to_f32(((const T*)p.post_in)[o + 16 + e]), dg, du);
      op[e] = from_f32<T>(dg); op[16 + e] = from_f32<T>(du);
    }
    return;
  }
  if (p.post) post_apply<T, 4>(p, row, col, v4);
#pragma unroll
  for (int e = 0; e < 4; ++e) store_out<T>(p, p.C, p.res, row, col + e, v4[e]);
}

static int splitk_runs(const GemmP& p, int batch, int bk, size_t ws_bytes) {
  if (p.colsum) return 0;        // column sums are taken in the tile epilogues (one adder per wave and column), not in the reduce pass
  if (!sl_env().split_k || p.ta || p.tw || p.grp || batch != 1 || p.K % bk || p.ln_mr || p.stats_out || p.amax_val || p.aux || p.N < 128 || (p.N & 3)) return 0;
  const int64_t t128 = (int64_t)((p.M + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN);
  const int nkt = p.K / bk;
  if (t128 > 256 || nkt < 32) return 0;          // above half the 512 slots (two blocks per CU) the chip is busy enough; short reductions
  int S = (int)(512 / t128);
  if (S > 8) S = 8;
  if (S > nkt / 12) S = nkt / 12;                // every run keeps >= 12 slabs (768 k) behind its prologue
  while (S > 1 && (nkt % S || (size_t)S * p.M * p.N * sizeof(float) > ws_bytes)) --S;
  return S >= 2 ? S : 0;
}

// both operands K-major (weight gradients): gemm_tiled_tt_kernel, with the reduction cut into S runs of whole slabs when the tiles alone
// leave CUs idle (two blocks per CU: 512 slots) and the caller supplied a workspace
template <typename T>
static bool tt_ok(const GemmP& p, int batch) {
  if (sizeof(T) != 2 || !sl_env().wgrad_tr || !p.ta || !p.tw || batch != 1 || p.grp || p.aux || p.ln_mr || p.stats_out || p.amax_val) return false;
  return p.M % TBM == 0 && p.N % TBN == 0 && p.lda % 8 == 0 && p.ldw % 8 == 0 && p.K >= 128 && p.wx == 0 && p.cx == 0 && p.rx == 0;
}

template <typename T>
static int launch_tt(GemmP& p, hipStream_t st, void* sk_ws, size_t sk_ws_bytes) {
  p.tiles_m = p.M / TBM;
  p.tiles_n = p.N / TBN;
  const int nt = p.tiles_m * p.tiles_n, nkt = (p.K + 63) / 64;
  int S = 1;
  const size_t ws = sk_ws && sk_ws_bytes > SK_FLAG_BYTES ? sk_ws_bytes - SK_FLAG_BYTES : 0;
  if (ws && nt < 512 && !(p.N & 3)) {
    S = 512 / nt;
    const int smax = sl_env().tt_max_splits > 0 ? sl_env().tt_max_splits : 8;
    if (S > smax) S = smax;
    if (S > nkt / 12) S = nkt / 12;
    while (S > 1 && (size_t)S * p.M * p.N * sizeof(float) > ws) --S;
    if (S < 1) S = 1;
  }
  const int spr = (nkt + S - 1) / S;
  S = (nkt + spr - 1) / spr;
  if (S == 1) {
    hipLaunchKernelGGL((gemm_tiled_tt_kernel<SL_ACT_NONE>), dim3(nt, 1), dim3(256), 0, st, p, spr);
    SL_CHECK_LAUNCH("gemm_tiled_tt");
    return 0;
  }
  float* part = (float*)((unsigned char*)sk_ws + SK_FLAG_BYTES);
  GemmP q = p;
  q.C = part; q.ldc = p.N; q.sC = (int64_t)p.M * p.N; q.out_f32 = 1;
  q.bias = nullptr; q.sBias = 0; q.res = nullptr; q.ldr = 0; q.sR = 0; q.res_f32 = 0;
  hipLaunchKernelGGL((gemm_tiled_tt_kernel<SL_ACT_NONE>), dim3(nt, S), dim3(256), 0, st, q, spr);
  SL_CHECK_LAUNCH("gemm_tiled_tt (K runs)");
  const int64_t vecs = ((int64_t)p.M * p.N + 3) / 4;
  hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, st, part, S, (int64_t)p.M * p.N, p);
  SL_CHECK_LAUNCH("splitk_reduce");
  return 0;
}

template <typename T, int ACT>
static int launch_tiled(GemmP& p, int batch, hipStream_t st, void* sk_ws = nullptr, size_t sk_ws_bytes = 0) {
  constexpr int BK_ = TROWB / (int)sizeof(T);
  if constexpr (ACT == SL_ACT_NONE && sizeof(T) == 2) {
    if (tt_ok<T>(p, batch)) return launch_tt<T>(p, st, sk_ws, sk_ws_bytes);
  }
  if constexpr (ACT == SL_ACT_NONE) {
    if (sk_ws) {
      const int S = splitk_runs(p, batch, BK_, sk_ws_bytes > SK_FLAG_BYTES ? sk_ws_bytes - SK_FLAG_BYTES : 0);
      if (S) {
        float* part = (float*)((unsigned char*)sk_ws + SK_FLAG_BYTES);      // behind the stream-K flags (which stay zero)
        GemmP q = p;
        q.K = p.K / S; q.sA = q.K; q.sW = q.K;                                // run z reads columns [z K/S, (z+1) K/S) of A and W
        q.C = part; q.ldc = p.N; q.sC = (int64_t)p.M * p.N; q.out_f32 = 1;
        q.bias = nullptr; q.sBias = 0; q.res = nullptr; q.ldr = 0; q.sR = 0; q.res_f32 = 0;
        q.post = 0; q.drop_thr24 = 0; q.post_in = nullptr; q.colsum = nullptr;      // the reduce pass applies them
        SL_TRY((launch_tiled<T, SL_ACT_NONE>(q, S, st)));
        const int64_t vecs = ((int64_t)p.M * p.N + 3) / 4;
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)((vecs + 255) / 256)), dim3(256), 0, st, part, S, (int64_t)p.M * p.N, p);
        SL_CHECK_LAUNCH("splitk_reduce");
        return 0;
      }
    }
  }
  if (sk_ws) {
    const int G = sk_grid(p, batch, BK_, sk_ws_bytes);
    if (G) {
      p.tiles_m = (p.M + XBM - 1) / XBM;
      p.tiles_n = (p.N + XBN - 1) / XBN;
      if constexpr (sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL) {
        const bool al = !(p.N & 7) && !(p.ldc & 7) && !((uintptr_t)p.C & 15) && (!p.res || (!(p.ldr & 7) && !((uintptr_t)p.res & 15)));
        if (al && !p.out_f32 && !p.res_f32 && !p.direct_epi && !sl_env().no_swap_epilogue) {
          hipLaunchKernelGGL((gemm_tiled256sk_kernel<T, ACT, true>), dim3(G), dim3(512), 0, st, p, (unsigned char*)sk_ws);
          SL_CHECK_LAUNCH("gemm_tiled256sk (swapped operands)");
          return 0;
        }
      }
      hipLaunchKernelGGL((gemm_tiled256sk_kernel<T, ACT, false>), dim3(G), dim3(512), 0, st, p, (unsigned char*)sk_ws);
      SL_CHECK_LAUNCH("gemm_tiled256sk");
      return 0;
    }
  }
  // large products: 256^2 tiles once they alone give every CU >= 2 tiles (ragged batches: sized by the largest group)
  if (!p.ta && !p.tw && p.K % BK_ == 0 && !p.grp_ext && g_disable_glds == 0 && !sl_env().disable_t256 && !((p.post || p.colsum) && !sl_env().t256_phased)) {
    const int famM = (p.grp || batch != 1) ? p.M : sl_family_rows(p.M);      // the tile family follows the pinned rows (common.h sl_family_rows)
    const int64_t t256 = (int64_t)((famM + XBM - 1) / XBM) * ((p.N + XBN - 1) / XBN) * batch;
    const int64_t min_tiles = sl_env().t256_min_tiles;   // tuning switches
    const int min_k = sl_env().t256_min_k;
    // rows padded to 256 vs to 128: short (grouped) products such as the 123-row projector would half-fill the big tile
    const int64_t m128 = (int64_t)((famM + TBM - 1) / TBM) * TBM, m256 = (int64_t)((famM + XBM - 1) / XBM) * XBM;
    // Mid-size products (KD windows: M = 2-8 k rows): neither tile count fills the chip evenly, so the choice is made on whole
    // rounds of tiles — 256 slots of one 256^2 tile per CU against 512 slots of 128^2 tiles (two blocks per CU, each at ~0.85 of
    // the big tile's rate per flop): 5072 x 3072 is 240 big tiles = one round (1.09 PF/s; 960 small ones = two rounds, 0.99),
    // 3200 x 5120 is 260 big tiles = two rounds, the second almost empty (0.63 PF/s; small tiles 0.98).  tools/sweep_t256.py.
    bool by_rounds = false;
    if (t256 < min_tiles && !p.grp && batch == 1) {
      const int64_t t128 = (int64_t)((famM + TBM - 1) / TBM) * ((p.N + TBN - 1) / TBN);
      const int64_t r256 = (t256 + 255) / 256, r128 = (t128 + 511) / 512;
      by_rounds = (double)r256 * (XBM * XBN) * 0.85 < (double)r128 * 2.0 * (TBM * TBN);
    }
    // rows padded to 256: the whole-rounds comparison above already prices the padding (it counts TILES), so a product it picks may pad
    // freely — 634 x 16 384 x 3 072 (the per-rank KD window) is 192 big tiles = one round against 640 small ones = two; only products
    // admitted by their tile count alone (grouped / batched: the 123-row projector) keep the 1/8 bound
    const bool pad_ok = m256 <= m128 + m128 / 8 || (by_rounds && sl_env().t256_by_rounds_pad);
    if ((t256 >= min_tiles || by_rounds) && p.N >= 192 && p.K >= min_k && pad_ok) {   // 1024: with the row epilogue the big tile also wins at K = 1024..1536 (+10..20 %)
      p.tiles_m = (p.M + XBM - 1) / XBM;
      p.tiles_n = (p.N + XBN - 1) / XBN;
      const int phased = sl_env().t256_phased;   // 0: round-3 one-barrier-per-slab loop (A/B)
      if constexpr (sizeof(T) == 2 && ACT != SL_ACT_SILU_MUL) {
        // swapped-operand form (register epilogue, 16-byte stores): plain bf16 stores on 8-element aligned rows, one of the forms
        // {bias}, {bias, residual}, {LayerNorm fold}, {bias, residual, row statistics}, {bias, pre-activation copy}
        const bool al = !(p.N & 7) && !(p.ldc & 7) && !(p.sC & 7) && !((uintptr_t)p.C & 15) &&
                        (!p.res || (!(p.ldr & 7) && !(p.sR & 7) && !((uintptr_t)p.res & 15)));
        const bool form = p.aux ? (!p.ln_mr && !p.stats_out && !p.res && !((uintptr_t)p.aux & 15))        // {bias, pre-activation copy}: the training forward's FFN1
                                : !p.ln_mr ? (!p.stats_out || p.res) : (!p.res && !p.stats_out);
        // post-ops on the swapped-operand kernels: dropout in {bias, pre-activation copy, GELU} or {bias, residual}; GELU' (+ colsum_out) and SwiGLU' on
        // the plain product; operand rows 16-byte aligned.  Everything else with a post-op takes the LDS-turned rows epilogue.
        bool post_ok = !p.post && !p.colsum;      // (the phased kernel only: the one-barrier-per-slab A/B form keeps the plain epilogues)
        if (p.post == SL_POST_DROPOUT) post_ok = !p.colsum && !p.ln_mr && !p.stats_out && (ACT == SL_ACT_GELU ? (p.aux && !p.res) : (p.res && !p.aux));
        if (p.post == SL_POST_GELU_BWD || p.post == SL_POST_SILU_MUL_BWD)
          post_ok = ACT == SL_ACT_NONE && !(p.post_ld & 7) && !((uintptr_t)p.post_in & 15) && (p.post == SL_POST_GELU_BWD || !p.colsum);
        if (al && form && !p.grp && !p.out_f32 && !p.res_f32 && !p.amax_val && !p.direct_epi && !sl_env().no_swap_epilogue && post_ok && (phased || (!p.post && !p.colsum))) {
#ifdef SL_GEMM_DEBUG
          if constexpr (ACT == SL_ACT_NONE) {       // instrumented / knocked-out builds (tools/gemm_stamps.py, tools/gemm_knockout.py), never in the product .so
            const int ko = sl_env().gemm_ko;
            if (phased && !p.ln_mr && (sl_env().gemm_stamp_ptr || ko)) {
              p.stamp = (uint32_t*)(uintptr_t)sl_env().gemm_stamp_ptr;
              const dim3 g(p.tiles_m * p.tiles_n, batch);
              if (p.stamp) hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 8>), g, dim3(512), 0, st, p);
              else if (ko == 1) hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 1>), g, dim3(512), 0, st, p);
              else if (ko == 2) hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 2>), g, dim3(512), 0, st, p);
              else if (ko == 3) hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 3>), g, dim3(512), 0, st, p);
              else if (ko == 4) hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 4>), g, dim3(512), 0, st, p);
              else hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true, 6>), g, dim3(512), 0, st, p);
              SL_CHECK_LAUNCH("gemm_tiled256 (debug)");
              return 0;
            }
          }
#endif
          if (phased) hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, true>), dim3(p.tiles_m * p.tiles_n, batch), dim3(512), 0, st, p);
          else hipLaunchKernelGGL((gemm_tiled256_kernel<T, ACT, true>), dim3(p.tiles_m * p.tiles_n, batch), dim3(512), 0, st, p);
          SL_CHECK_LAUNCH("gemm_tiled256 (swapped operands)");
          return 0;
        }
      }
      if (phased) hipLaunchKernelGGL((gemm_tiled256p_kernel<T, ACT, false>), dim3(p.tiles_m * p.tiles_n, batch), dim3(512), 0, st, p);
      else hipLaunchKernelGGL((gemm_tiled256_kernel<T, ACT>), dim3(p.tiles_m * p.tiles_n, batch), dim3(512), 0, st, p);
      SL_CHECK_LAUNCH("gemm_tiled256");
      return 0;
    }
  }
  p.tiles_m = (p.M + TBM - 1) / TBM;
  p.tiles_n = (p.N + TBN - 1) / TBN;
  dim3 grid(p.tiles_m * p.tiles_n, batch);
  constexpr int BK = TROWB / (int)sizeof(T);
  if ((p.post || p.colsum) && !(!p.ta && !p.tw && p.K % BK == 0 && (!p.grp_ext || p.grp_kslab) && !g_disable_glds)) {
    sl_set_error("sl_gemm_ex: post_op / colsum_out need whole 128-byte K slabs (K %% %d == 0) and the LDS-DMA kernels (SL_DISABLE_GLDS unset)", BK);
    return SL_ERR_UNSUPPORTED;
  }
  if (!p.ta && !p.tw && p.K % BK == 0 && !p.grp_ext && g_disable_glds == 2)
    hipLaunchKernelGGL((gemm_tiled_glds_kernel<T, ACT, false>), grid, dim3(256), 0, st, p);
  else if (!p.ta && !p.tw && p.K % BK == 0 && (!p.grp_ext || p.grp_kslab) && !g_disable_glds)   // per-group K: only the register path handles K tails (groups_ext = 2: the caller vouches for whole slabs)
    hipLaunchKernelGGL((gemm_tiled_glds_kernel<T, ACT, true>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((gemm_tiled_kernel<T, ACT>), grid, dim3(256), 0, st, p);
  SL_CHECK_LAUNCH("gemm_tiled");
  return 0;
}

template <typename T, int MT, int ACT, int RF, int NW, int U, bool PACKED, bool KCONT = false>
static int launch_skinny_cfg(GemmP& p, const SkinnyX& sx, int batch, hipStream_t st) {
  dim3 grid((p.N + 16 * RF - 1) / (16 * RF), batch);
  hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, ACT, RF, NW, U, PACKED, KCONT>), grid, dim3(NW * 64), 0, st, p, sx);
  SL_CHECK_LAUNCH("gemm_skinny");
  return 0;
}

// Structure per shape, from the sweeps in tools/tune_skinny.hip (MI355X, bf16, packed weights):
//   M <= 16:  gate/up, lm_head (>= 1024 fragments): 4 fragments x 4 waves x 4 steps in flight   (5.3 / 6.2 TB/s)
//             qkv (320 fragments):                   2 fragments x 8 waves x 4 steps            (2.9 TB/s)
//             N = hidden (o, down; 192 fragments):   1 fragment x 16 waves x 2 steps            (2.7 / 3.3 TB/s)
//             (2 x 16 x 4 and contiguous K slices looked 5-12 % better in the bare sweep but were slower inside the
//              real kernel with its fused-norm / pair epilogues: decode step 2.19 vs 1.98 ms)
//   M 17..64: activations are re-read per fragment from L2, so fewer fragments per wave and wide K splits win.
template <typename T, int MT, int ACT, bool PACKED>
static int launch_skinny_mt(GemmP& p, const SkinnyX& sx, int batch, hipStream_t st) {
  constexpr bool PAIRS = (ACT == SL_ACT_SILU_MUL || ACT == SL_ACT_ROPE_KV);
  const int nfrag = (p.N + 15) / 16 * batch;
  if constexpr (MT == 1) {
    if constexpr (PACKED && sizeof(T) == 2 && !PAIRS) {
      // N = hidden (o, down) at M <= 8: three or four 64-byte steps of loads in flight per wave, chosen so the wave's share
      // of K divides evenly (o: 6 steps = 2 x 3, down: 16 = 4 x 4): 1.575 -> 1.49 ms per decode step at M = 1, +3 % at M = 16
      // where the x fragments crowd the loads.  The same sweep over the qkv and gate/up structures (2x16x3, 2x8x6, 2x8x3;
      // 4x4x6, 4x4x3, 2x8x3, 2x8x6, 4x8x3) moved nothing (profiles/r04_r_skinny_small_m.txt).  SL_SKINNY_ALT=1: old structure.
      if (nfrag < 256 && sl_family_rows(p.M) <= 8 && !(sl_env().skinny_alt & 1)) {
        const int per_wave = p.K / 32 / 16;
        if (per_wave % 3 == 0) return launch_skinny_cfg<T, MT, ACT, 1, 16, 3, PACKED>(p, sx, batch, st);
        return launch_skinny_cfg<T, MT, ACT, 1, 16, 4, PACKED>(p, sx, batch, st);
      }
    }
    if (nfrag >= 1024) return launch_skinny_cfg<T, MT, ACT, 4, 4, 4, PACKED>(p, sx, batch, st);
    if (nfrag >= 256 || PAIRS) return launch_skinny_cfg<T, MT, ACT, 2, 8, 4, PACKED>(p, sx, batch, st);
    if constexpr (!PAIRS) return launch_skinny_cfg<T, MT, ACT, 1, 16, 2, PACKED>(p, sx, batch, st);
  } else if constexpr (MT == 2) {
    if (nfrag >= 1024) return launch_skinny_cfg<T, MT, ACT, 2, 8, 4, PACKED>(p, sx, batch, st);
    if (nfrag >= 256 || PAIRS) return launch_skinny_cfg<T, MT, ACT, 2, 16, 2, PACKED>(p, sx, batch, st);
    if constexpr (!PAIRS) return launch_skinny_cfg<T, MT, ACT, 1, 16, 2, PACKED>(p, sx, batch, st);
  } else {
    if (nfrag >= 1024) return launch_skinny_cfg<T, MT, ACT, 4, 8, 2, PACKED>(p, sx, batch, st);
    if (nfrag >= 256 || PAIRS) return launch_skinny_cfg<T, MT, ACT, 2, 16, 2, PACKED>(p, sx, batch, st);
    if constexpr (!PAIRS) return launch_skinny_cfg<T, MT, ACT, 1, 16, 2, PACKED>(p, sx, batch, st);
  }
  return 0;
}

template <typename T, int ACT>
static int launch_skinny(GemmP& p, const SkinnyX& sx, int batch, bool packed, hipStream_t st) {
  if (packed) {
    if constexpr (ACT == SL_ACT_GELU) {
      sl_set_error("sl_gemm: packed weights are not built with the GELU epilogue");
      return SL_ERR_UNSUPPORTED;
    } else {
      if (sl_family_rows(p.M) <= 16) return launch_skinny_mt<T, 1, ACT, true>(p, sx, batch, st);
      if (sl_family_rows(p.M) <= 32) return launch_skinny_mt<T, 2, ACT, true>(p, sx, batch, st);
      return launch_skinny_mt<T, 4, ACT, true>(p, sx, batch, st);
    }
  }
  if constexpr (ACT == SL_ACT_ROPE_KV) {
    sl_set_error("sl_gemm: the ROPE_KV epilogue needs packed weights");
    return SL_ERR_UNSUPPORTED;
  } else {
    if (sl_family_rows(p.M) <= 16) return launch_skinny_mt<T, 1, ACT, false>(p, sx, batch, st);
    if (sl_family_rows(p.M) <= 32) return launch_skinny_mt<T, 2, ACT, false>(p, sx, batch, st);
    return launch_skinny_mt<T, 4, ACT, false>(p, sx, batch, st);
  }
}

template <typename T>
static int gemm_typed(const sl_gemm_args* a, GemmP& p, const SkinnyX& sx, hipStream_t st, void* sk_ws, size_t sk_ws_bytes) {
  const bool skinny = sl_family_rows(a->M) <= 64 && !p.ta && !p.tw && !p.aux && !p.res_f32 && !p.grp && !p.ln_mr && !p.stats_out && !p.post && !p.colsum;  // backward features and the LayerNorm fold live in the tiled kernel
  const bool packed = a->w_layout == SL_W_PACKED;
  if (!skinny && (packed || a->act == SL_ACT_ROPE_KV || sx.fuse_rms)) {
    sl_set_error("sl_gemm: packed weights / ROPE_KV / fused RMSNorm need plain operands (no transposes / groups), M=%d", a->M);
    return SL_ERR_UNSUPPORTED;
  }
  switch (a->act) {
    case SL_ACT_NONE: return skinny ? launch_skinny<T, SL_ACT_NONE>(p, sx, a->batch, packed, st) : launch_tiled<T, SL_ACT_NONE>(p, a->batch, st, sk_ws, sk_ws_bytes);
    case SL_ACT_GELU: return skinny ? launch_skinny<T, SL_ACT_GELU>(p, sx, a->batch, packed, st) : launch_tiled<T, SL_ACT_GELU>(p, a->batch, st, sk_ws, sk_ws_bytes);
    case SL_ACT_SILU_MUL: return skinny ? launch_skinny<T, SL_ACT_SILU_MUL>(p, sx, a->batch, packed, st) : launch_tiled<T, SL_ACT_SILU_MUL>(p, a->batch, st, sk_ws, sk_ws_bytes);
    case SL_ACT_ROPE_KV: return launch_skinny<T, SL_ACT_ROPE_KV>(p, sx, a->batch, packed, st);
  }
  sl_set_error("sl_gemm: unknown act %d", a->act);
  return SL_ERR_ARG;
}

// true when a plain (M, N, K) product of this dtype is served by one of the LDS-DMA tiled kernels, whose rows epilogue carries the
// LayerNorm fold (ln_* / stats_out)
// shapes for which the training tapes may hand a product its post-ops (train_tape.hip fuse_ok): the tiled kernels' row range, whole K slabs
bool sl_gemm_post_ok(int64_t M, int N, int K, int dtype) {
  return sl_family_rows((int)(M > 0x7fffffff ? 0x7fffffff : M)) > 64 && M > 64 && N % 16 == 0 && K % (dtype == SL_F32 ? 32 : 64) == 0 && sl_env().disable_glds == 0;
}

bool sl_gemm_rows_epilogue_ok(int M, int N, int K, int dtype) {
  // any row count: a product that carries ln_* / stats_out is kept on the tiled kernels even below 65 rows (gemm_typed), so that the fold is a
  // property of the MODEL — a short utterance encoded alone takes the same epilogues, hence the same bits, as inside a batch
  if (dtype != SL_BF16 || M <= 0 || (N & 63) || sl_env().disable_glds != 0 || sl_env().direct_epilogue != 0) return false;
  return K % (TROWB / 2) == 0;
}

int sl_gemm_impl(const sl_gemm_args* a, const sl_gemm_fused* fx, const sl_gemm_ex_args* ex, hipStream_t st) {
  SL_CHECK_ARG(a != nullptr, "sl_gemm: null args");
  g_disable_glds = sl_env().disable_glds;   // 1: register staging, 2: glds with compiler-visible LDS reads
  SL_CHECK_ARG(a->M > 0 && a->N > 0 && a->K > 0 && a->batch > 0, "sl_gemm: bad shape M=%d N=%d K=%d batch=%d", a->M, a->N, a->K, a->batch);
  SL_CHECK_ARG(a->dtype == SL_F32 || a->dtype == SL_BF16, "sl_gemm: bad dtype %d", a->dtype);
  const int vec = a->dtype == SL_F32 ? 4 : 8;
  // a K-contiguous (non-transposed) operand is read in 16-byte chunks along K; transposed ones along the output index
  const bool both_t = ex && ex->trans_a && ex->trans_w;
  SL_CHECK_ARG(both_t || a->K % vec == 0, "sl_gemm: K=%d must be a multiple of %d", a->K, vec);
  SL_CHECK_ARG(a->lda % vec == 0 && a->strideA % vec == 0, "sl_gemm: lda/strideA must keep rows 16-byte aligned");
  SL_CHECK_ARG(((uintptr_t)a->A & 15) == 0 && ((uintptr_t)a->W & 15) == 0, "sl_gemm: A and W must be 16-byte aligned");
  if (a->w_layout == SL_W_PACKED) {
    SL_CHECK_ARG(a->K % (4 * vec) == 0, "sl_gemm: packed weights need K %% %d == 0", 4 * vec);
  } else {
    SL_CHECK_ARG(a->w_layout == SL_W_ROWMAJOR, "sl_gemm: unknown w_layout %d", a->w_layout);
    SL_CHECK_ARG(a->ldw % vec == 0 && a->strideW % vec == 0, "sl_gemm: ldw/strideW must keep rows 16-byte aligned");
  }
  if (a->act == SL_ACT_SILU_MUL) SL_CHECK_ARG(a->N % 32 == 0, "sl_gemm: SILU_MUL needs N %% 32 == 0 (16-row gate/up blocks)");
  GemmP p;
  p.A = a->A; p.lda = a->lda; p.sA = a->strideA;
  p.W = a->W; p.ldw = a->ldw; p.sW = a->strideW;
  p.C = a->C; p.ldc = a->ldc; p.sC = a->strideC;
  p.bias = a->bias; p.sBias = a->strideBias;
  p.res = a->residual; p.ldr = a->ldr; p.sR = a->strideR;
  p.M = a->M; p.N = a->N; p.K = a->K; p.out_f32 = a->out_f32;
  p.tiles_m = p.tiles_n = 0;
  p.ta = p.tw = 0; p.aux = nullptr; p.res_f32 = 0; p.grp = nullptr; p.w_mod = 1; p.cx = p.rx = p.wx = 0; p.grp_ext = 0; p.grp_kslab = 0;
  p.post = 0; p.drop_thr24 = 0; p.drop_scale = 1.f; p.drop_seed = 0; p.drop_ld = 0; p.post_in = nullptr; p.post_ld = 0; p.colsum = nullptr;
  p.stamp = nullptr; p.amax_val = nullptr; p.amax_idx = nullptr; p.ln_mr = nullptr; p.ln_u = nullptr; p.ln_c = nullptr; p.stats_out = nullptr;
  const int direct_epi = sl_env().direct_epilogue;
  p.direct_epi = direct_epi;
  const int gm_env = sl_env().gemm_gm;
  p.gm = gm_env;
  if (ex) {
    p.ta = ex->trans_a; p.tw = ex->trans_w; p.aux = ex->aux_out; p.res_f32 = ex->residual_f32;
    p.grp = ex->groups; p.w_mod = ex->w_mod > 0 ? ex->w_mod : 1;
    p.grp_ext = ex->groups && ex->groups_ext;
    p.grp_kslab = ex->groups && ex->groups_ext == 2;
    SL_CHECK_ARG(!(p.ta || p.tw || p.aux) || a->act != SL_ACT_SILU_MUL, "sl_gemm_ex: transposed operands / aux_out are not combined with SILU_MUL");
    SL_CHECK_ARG(!p.res_f32 || a->out_f32, "sl_gemm_ex: residual_f32 needs out_f32");
    SL_CHECK_ARG(!(p.ta || p.tw) || a->w_layout == SL_W_ROWMAJOR, "sl_gemm_ex: transposed operands need row-major storage");
    if (ex->ln_mr || ex->ln_u || ex->ln_c || ex->stats_out) {
      // both sides of the LayerNorm fold live in the rows epilogue of the LDS-DMA tiled kernels (bf16, 4-column vectors)
      SL_CHECK_ARG(sl_gemm_rows_epilogue_ok(a->M, a->N, a->K, a->dtype) && a->batch == 1 && !ex->groups && !ex->trans_a && !ex->trans_w && !ex->aux_out &&
                       a->act != SL_ACT_SILU_MUL && a->w_layout == SL_W_ROWMAJOR && a->ldc % 4 == 0 && ((uintptr_t)a->C & 7) == 0 &&
                       (!a->residual || (a->ldr % 4 == 0 && ((uintptr_t)a->residual & 7) == 0)),
                   "sl_gemm_ex: ln_* / stats_out need a plain bf16 row-major product the LDS-DMA tiled kernels take (M=%d N=%d K=%d), 4-element aligned rows", a->M, a->N, a->K);
      SL_CHECK_ARG((!ex->ln_mr && !ex->ln_u && !ex->ln_c) || (ex->ln_mr && ex->ln_u && ex->ln_c && !a->bias),
                   "sl_gemm_ex: the LayerNorm fold needs ln_mr, ln_u and ln_c together (the bias is inside ln_c)");
      p.ln_mr = ex->ln_mr; p.ln_u = ex->ln_u; p.ln_c = ex->ln_c; p.stats_out = ex->stats_out;
    }
    if (ex->post_op || ex->colsum_out) {
      SL_CHECK_ARG(ex->post_op >= SL_POST_NONE && ex->post_op <= SL_POST_SILU_MUL_BWD, "sl_gemm_ex: unknown post_op %d", ex->post_op);
      SL_CHECK_ARG(!ex->trans_a && !ex->trans_w && !ex->groups && !ex->ln_mr && !ex->ln_u && !ex->ln_c && !ex->stats_out && !ex->amax_val && !ex->amax_idx &&
                       a->batch == 1 && a->w_layout == SL_W_ROWMAJOR && a->act != SL_ACT_SILU_MUL && a->act != SL_ACT_ROPE_KV && a->M > 64,
                   "sl_gemm_ex: post_op / colsum_out need one plain row-major product on the tiled kernels (M > 64, no transposes / groups / ln_* / stats_out / amax_*)");
      SL_CHECK_ARG(ex->drop_p >= 0.f && ex->drop_p < 1.f, "sl_gemm_ex: drop_p %f outside [0, 1)", (double)ex->drop_p);
      if (ex->post_op == SL_POST_DROPOUT) SL_CHECK_ARG(ex->drop_p > 0.f && ex->drop_ld >= a->N, "sl_gemm_ex: SL_POST_DROPOUT needs drop_p > 0 and drop_ld >= N");
      if (ex->post_op == SL_POST_GELU_BWD)
        SL_CHECK_ARG(ex->post_in && ex->post_ld >= a->N && !a->residual && !a->bias && a->act == SL_ACT_NONE && !ex->aux_out && !a->out_f32 && (ex->drop_p == 0.f || ex->drop_ld >= a->N),
                     "sl_gemm_ex: SL_POST_GELU_BWD needs post_in (M, N), the plain epilogue (no bias / residual / act / aux_out) and an output in the storage type");
      if (ex->post_op == SL_POST_SILU_MUL_BWD)
        SL_CHECK_ARG(ex->post_in && ex->post_ld >= 2 * (int64_t)a->N && a->ldc >= 2 * (int64_t)a->N && a->N % 16 == 0 && !a->residual && !a->bias && a->act == SL_ACT_NONE &&
                         !ex->aux_out && !a->out_f32 && !ex->colsum_out && ex->drop_p == 0.f,
                     "sl_gemm_ex: SL_POST_SILU_MUL_BWD needs post_in = gu (M, 2 N), ldc >= 2 N, N %% 16 == 0 and the plain epilogue");
      p.post = ex->post_op; p.post_in = ex->post_in; p.post_ld = ex->post_ld; p.colsum = ex->colsum_out;
      if (ex->drop_p > 0.f && ex->post_op != SL_POST_NONE) {
        p.drop_thr24 = (uint32_t)((double)ex->drop_p * 16777216.0);
        p.drop_scale = 1.0f / (1.0f - ex->drop_p);
        p.drop_seed = ex->drop_seed; p.drop_ld = ex->drop_ld;
      }
    }
    if (ex->amax_val || ex->amax_idx) {
      SL_CHECK_ARG(ex->amax_val && ex->amax_idx && a->act == SL_ACT_NONE && a->batch == 1 && !ex->groups && !ex->trans_a && !ex->trans_w && !ex->aux_out &&
                       !a->residual && sl_family_rows(a->M) > 64 && a->w_layout == SL_W_ROWMAJOR,
                   "sl_gemm_ex: amax_val / amax_idx (fused row-wise top-1) need both pointers, the plain epilogue without residual, one "
                   "un-grouped row-major product and M > 64 (M=%d act=%d batch=%d)", a->M, a->act, a->batch);
      p.amax_val = ex->amax_val; p.amax_idx = ex->amax_idx;
    }
  }
  SL_CHECK_ARG(p.amax_val || a->C, "sl_gemm: null C");
  if (sl_env().gemm_log)     // SL_GEMM_LOG=1: one line per product on stderr (tools/kd_gemm_shapes.py turns a KD window's lines into a per-shape table)
    fprintf(stderr, "SLGEMM M=%d N=%d K=%d batch=%d act=%d ta=%d tw=%d res=%d resf32=%d outf32=%d aux=%d grp=%d bias=%d packed=%d dt=%d\n", a->M, a->N, a->K, a->batch, a->act,
            p.ta, p.tw, a->residual != nullptr, p.res_f32, a->out_f32, p.aux != nullptr, p.grp ? (p.grp_ext ? 2 : 1) : 0, a->bias != nullptr, a->w_layout == SL_W_PACKED, a->dtype);
  SkinnyX sx;
  memset(&sx, 0, sizeof(sx));
  if (fx) {
    sx.fuse_rms = fx->fuse_rms; sx.eps = fx->rms_eps;
    sx.cos = fx->rope_cos; sx.sin = fx->rope_sin; sx.pos = fx->tok_pos; sx.seq = fx->tok_seq;
    sx.kc = fx->k_cache; sx.vc = fx->v_cache; sx.nh = fx->n_heads; sx.nkv = fx->n_kv_heads; sx.max_ctx = fx->max_ctx;
    sx.rstd_in = fx->rstd_in; sx.rstd_out = fx->rstd_out; sx.norm_out = fx->norm_out; sx.norm_gain = fx->norm_gain;
    SL_CHECK_ARG((!fx->norm_out && !fx->norm_gain) || (fx->norm_out && fx->norm_gain && fx->rstd_out && !a->out_f32), "sl_gemm: norm_out needs norm_gain, rstd_out and an output in the storage type");
  }
  if (a->act == SL_ACT_ROPE_KV) {
    SL_CHECK_ARG(fx && fx->rope_cos && fx->rope_sin && fx->tok_pos && fx->tok_seq && fx->k_cache && fx->v_cache,
                 "sl_gemm: ROPE_KV epilogue needs the sl_gemm_fused tables");
    SL_CHECK_ARG(a->N == (fx->n_heads + 2 * fx->n_kv_heads) * 128 && a->batch == 1, "sl_gemm: ROPE_KV expects N = (n_heads + 2 n_kv) * 128");
  }
  // packed weights with more than g_stream_min_m rows: LDS-staged streaming kernel (gemm_stream.hip)
  if (a->w_layout == SL_W_PACKED && sl_family_rows(a->M) > stream_min_m() && a->batch == 1 && !ex && a->act != SL_ACT_GELU &&
      a->K % (a->dtype == SL_F32 ? 32 : 64) == 0)
    return sl_gemm_stream_launch(p, sx, a->dtype, a->act, fx ? fx->split_ws : nullptr, fx ? fx->split_ws_bytes : 0, st);
  SL_CHECK_ARG(a->w_layout != SL_W_PACKED || sl_family_rows(a->M) <= 64, "sl_gemm: packed weights with M=%d > 64 need batch 1 and K %% 64 == 0", a->M);
  SL_CHECK_ARG(!sx.rstd_in && !sx.rstd_out && !sx.norm_out, "sl_gemm: rstd_in / rstd_out / norm_out are features of the streaming path (M > %d rows, packed weights)", stream_min_m());
  void* sk_ws = ex ? ex->sk_ws : nullptr;
  const size_t sk_ws_bytes = ex ? ex->sk_ws_bytes : 0;
  SL_CHECK_ARG(!sk_ws || ((uintptr_t)sk_ws & 15) == 0, "sl_gemm_ex: sk_ws must be 16-byte aligned");
  if (a->dtype == SL_F32) return gemm_typed<float>(a, p, sx, st, sk_ws, sk_ws_bytes);
  return gemm_typed<bf16_t>(a, p, sx, st, sk_ws, sk_ws_bytes);
}

extern "C" int32_t sl_gemm_split_count(int32_t M, int32_t N, int32_t K, int32_t dtype) {
  M = sl_family_rows(M);
  if (M <= stream_min_m() || M <= 0 || N <= 0 || K <= 0 || K % (dtype == SL_F32 ? 32 : 64) != 0) return 1;
  return sl_gemm_stream_splits(M, N, K, dtype);
}

extern "C" size_t sl_gemm_split_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t dtype) {
  if (sl_family_rows(M) <= stream_min_m() || M <= 0 || N <= 0 || K <= 0) return 0;
  return sl_gemm_stream_ws_bytes(M, N, K, dtype);
}

extern "C" int32_t sl_gemm_ln_fold_ok(int32_t M, int32_t N, int32_t K, int32_t dtype) { return sl_gemm_rows_epilogue_ok(M, N, K, dtype) ? 1 : 0; }

extern "C" int sl_gemm(const sl_gemm_args* a, sl_stream stream) { return sl_gemm_impl(a, nullptr, nullptr, (hipStream_t)stream); }

extern "C" int sl_gemm_ex(const sl_gemm_args* a, const sl_gemm_ex_args* ex, sl_stream stream) {
  SL_CHECK_ARG(ex != nullptr, "sl_gemm_ex: null ex args");
  return sl_gemm_impl(a, nullptr, ex, (hipStream_t)stream);
}

extern "C" int sl_gemm_fused_decode(const sl_gemm_args* a, const sl_gemm_fused* fx, sl_stream stream) {
  SL_CHECK_ARG(fx != nullptr, "sl_gemm_fused_decode: null fused args");
  return sl_gemm_impl(a, fx, nullptr, (hipStream_t)stream);
}
