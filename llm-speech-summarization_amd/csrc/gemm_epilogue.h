// gemm_epilogue.h — the epilogues shared by the tiled GEMM kernels (gemm.hip: 128-tile kernels; gemm_tt.hip: weight-gradient kernel;
// gemm256.hip: 256-tile kernels).  Device code only; every including translation unit instantiates what its kernels use.
#pragma once
#include "common.h"
#include "gemm_internal.h"

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

// ----------------------------------------------------------------------------------------------
// LDS reads of the LDS-DMA kernels
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
