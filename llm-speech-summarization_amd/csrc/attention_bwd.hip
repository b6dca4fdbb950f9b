// attention_bwd.hip — flash-style attention BACKWARD on MFMA for packed variable-length sequences (KD step):
// the S x S probabilities are recomputed tile by tile from q, k and the forward's log-sum-exp, never stored.
//   HuBERT / Whisper: head_dim 64, bidirectional, optional dropout of the probabilities (hf:models/hubert/modeling_hubert.py:234-259)
//   Llama (frozen, data gradients only): head_dim 128, causal, GQA (hf:models/llama/modeling_llama.py:191-213)
// With S = scale q k^T, P = softmax(S), Pd = dropout(P), O = Pd v and delta_i = sum_d dO_id O_id:
//   dV = Pd^T dO,   dPd = dO v^T,   dS = P o (mask/(1-p) o dPd - delta),   dQ = scale dS k,   dK = scale dS^T q.
// Two kernels, no atomics (bitwise reproducible):
//   * attn_bwd_dkdv_kernel — one block per (sequence, kv head, 64 keys); a wave keeps 16 keys' K / V fragments and their dK / dV
//     accumulators in registers and sweeps the query tiles of every query head of its GQA group.  S and dP are computed with
//     the QUERY on the accumulator row and the key on the lane (S = Q K^T), so that P and dS, packed, are directly the A
//     operands of dV += Pd^T dO and dK += dS^T Q (fp32: the accumulator as it stands; bf16: two 16-query fragments interleaved,
//     contraction slot 8a+i <-> query 4a+i of the even fragment, 8a+4+i <-> the odd one, and the transposed Q / dO images are
//     stored with their columns in that order).
//   * attn_bwd_dq_kernel — one block per (sequence, head, 64 queries); a wave keeps 16 queries' q / dO fragments, lse and delta
//     and sweeps the key tiles: S^T = K Q^T puts the key on the accumulator row, dS^T packed is the B operand of
//     dQ^T += K^T dS^T (K^T image with permuted columns, as above), the epilogue stores 4 consecutive dims per lane.
// Both run on the dtype-generic 16x16 MFMA step (bf16 16x16x32 / exact-fp32 16x16x4), so the fp32 parity mode and the bf16
// mode share the code; tiles are staged through XOR-swizzled LDS.
#include "common.h"

struct AttnBwdP {
  const void* q; int64_t q_rs, q_hs;
  const void* k; int64_t k_rs, k_hs;
  const void* v; int64_t v_rs, v_hs;
  const void* o; int64_t o_rs, o_hs;
  const void* dout; int64_t do_rs, do_hs;
  void* dq; int64_t dq_rs, dq_hs;
  void* dk; int64_t dk_rs, dk_hs;
  void* dv; int64_t dv_rs, dv_hs;
  const float* lse;
  float* delta;
  const int32_t* cu_q; const int32_t* cu_k; const int32_t* klen;
  int64_t n_tok_q;
  int n_heads, n_kv;
  float scale;
  uint32_t drop_thr;
  float drop_scale;
  uint64_t drop_seed;
};

namespace {

// swizzled byte offset of 16-byte chunk `ch` of row `row`; rows hold CPR chunks (power of two)
template <int CPR>
__device__ __forceinline__ int swz(int row, int ch) {
  constexpr int MASK = (CPR < 16 ? CPR : 16) - 1;
  return row * (CPR * 16) + ((ch ^ (row & MASK)) << 4);
}

// column of tile row `row` inside a transposed image: bf16 interleaves the two 16-row fragments of a 32-row group so that
// the 8 contraction slots a lane supplies (8a .. 8a+7) are rows {4a..4a+3} of the even fragment then of the odd one
template <typename T>
__device__ __forceinline__ int tpos(int row) {
  if constexpr (sizeof(T) == 2) return (row & ~31) | (((row >> 2) & 3) << 3) | (((row >> 4) & 1) << 2) | (row & 3);
  else return row;
}

// element j of a 16-byte chunk (no pointer punning: a uint4 store read back through float* is undefined under TBAA and
// miscompiled the fp32 instantiation)
template <typename T>
__device__ __forceinline__ T chunk_elem(const uint4& u, int j) {
  const uint32_t w[4] = {u.x, u.y, u.z, u.w};
  if constexpr (sizeof(T) == 2) return T{(uint16_t)(w[j >> 1] >> (16 * (j & 1)))};
  else return __builtin_bit_cast(float, w[j]);
}

template <typename T> __device__ __forceinline__ float exp_scaled(float x) {   // exp(x)
  if constexpr (sizeof(T) == 2) return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
  else return expf(x);
}

// pack accumulator fragments into the A / B operand of one k-step over the tile's rows
template <typename T, int NFR>
__device__ __forceinline__ uint4 pack_step(const f32x4 (&x)[NFR], int ks) {
  if constexpr (sizeof(T) == 2) {
    return make_uint4(pack2_bf16(x[2 * ks][0], x[2 * ks][1]), pack2_bf16(x[2 * ks][2], x[2 * ks][3]),
                      pack2_bf16(x[2 * ks + 1][0], x[2 * ks + 1][1]), pack2_bf16(x[2 * ks + 1][2], x[2 * ks + 1][3]));
  } else {
    // copy the elements to scalars first: hipcc (ROCm 7.2) miscompiles __builtin_bit_cast applied directly to an ext_vector
    // ELEMENT (x[ks][i]) — every cast reads element 0 (verified in isolation on the .s)
    const float f0 = x[ks][0], f1 = x[ks][1], f2 = x[ks][2], f3 = x[ks][3];
    return make_uint4(__float_as_uint(f0), __float_as_uint(f1), __float_as_uint(f2), __float_as_uint(f3));
  }
}

// bf16: the B (or A) operand of a product that contracts over the ROWS of a row-major LDS tile — 8 contraction slots of lane (r, a): rows
// base + 4a + {0..3} (slots 8a .. 8a+3) and base + 16 + 4a + {0..3} (slots 8a+4 .. 8a+7) at column 16 n + r — gathered straight from the
// row-major image by two ds_read_b64_tr_b16 (within a 16-lane group lane 4 qq + pp addresses row qq's 8-byte piece pp of the 32 bytes that
// hold the 16 columns; the instruction hands lane r the four rows' values of column r).  The slot order is the one pack_step gives the
// other operand.  Round 5: replaces a second, transposed LDS image of every tile, which cost 16 two-byte stores per staged chunk with up to
// 16-way bank conflicts (its rows are 64-128 bytes: lanes that differ in the source chunk land on one bank).
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_b_t;
typedef __attribute__((address_space(3))) void* lds_ptr_b_t;
template <int CPR>
__device__ __forceinline__ void tr_issue(uint32_t img, int row_base, int n, int lane, u32x2_b_t& lo, u32x2_b_t& hi) {
  const int r = lane & 15, a = lane >> 4, qq = r >> 2, pp = r & 3;
  const int r0 = row_base + 4 * a + qq, r1 = r0 + 16;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(img + (uint32_t)(swz<CPR>(r0, 2 * n + (pp >> 1)) + 8 * (pp & 1))) : "memory");
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(img + (uint32_t)(swz<CPR>(r1, 2 * n + (pp >> 1)) + 8 * (pp & 1))) : "memory");
}
// one wait for a group of fragments in flight (every register tied to it, so that no product moves above the wait): a wait per fragment
// exposed the LDS latency sixteen times per tile (llama shape 134 -> 130 us per layer, hubert shape 212 -> 207, tools/time_attn_bwd.py)
template <int NG>
__device__ __forceinline__ void tr_wait(u32x2_b_t (&lo)[NG], u32x2_b_t (&hi)[NG]) {
  static_assert(NG == 4 || NG == 8, "groups of 4 or 8 fragments");
  if constexpr (NG == 4)
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]));
  else
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[0]), "+v"(hi[0]), "+v"(lo[1]), "+v"(hi[1]), "+v"(lo[2]), "+v"(hi[2]), "+v"(lo[3]), "+v"(hi[3]), "+v"(lo[4]),
                 "+v"(hi[4]), "+v"(lo[5]), "+v"(hi[5]), "+v"(lo[6]), "+v"(hi[6]), "+v"(lo[7]), "+v"(hi[7]));
}

// delta[t][h] = sum_d dO[t][h][d] * O[t][h][d]
template <typename T, int D>
__global__ __launch_bounds__(256) void attn_bwd_delta_kernel(AttnBwdP p) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int LPP = D / VEC;          // lanes per (token, head) pair
  const int64_t pair = ((int64_t)blockIdx.x * 256 + threadIdx.x) / LPP;
  const int sub = threadIdx.x % LPP;
  const int64_t n_pairs = p.n_tok_q * p.n_heads;
  float acc = 0.f;
  if (pair < n_pairs) {
    const int64_t t = pair / p.n_heads;
    const int h = (int)(pair % p.n_heads);
    const uint4 uo = *(const uint4*)((const T*)p.o + t * p.o_rs + (int64_t)h * p.o_hs + sub * VEC);
    const uint4 ud = *(const uint4*)((const T*)p.dout + t * p.do_rs + (int64_t)h * p.do_hs + sub * VEC);
    float fo[VEC], fd[VEC];
    Vec16<T>::unpack(uo, fo);
    Vec16<T>::unpack(ud, fd);
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc += fo[j] * fd[j];
  }
#pragma unroll
  for (int o = LPP / 2; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if (pair < n_pairs && sub == 0) p.delta[pair] = acc;
}

// KF = 16-key fragments per wave (round 6: 2, a block owns 128 keys): every Q / dO fragment read from LDS — row-major for S and dP, transposed for
// dV and dK — now feeds KF products instead of one.  With one fragment per wave the kernel was LDS-bound (per tile and wave 16 ds_read_b128 +
// 32 ds_read_b64_tr for 32 MFMAs: ~256 LDS cycles per wave, four waves on one LDS port, against 512 matrix-core cycles per SIMD).
template <typename T, int D, bool CAUSAL, int TQ, int KF>
__device__ __forceinline__ void attn_bwd_dkdv_body(const AttnBwdP& p, const int bx, const int by, const int bz) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int SZ = (int)sizeof(T);
  constexpr int KSTEP = MMA<T>::KSTEP;
  constexpr int KS_D = D / KSTEP;       // k-steps across the head dim (S, dP)
  constexpr int KS_Q = TQ / KSTEP;      // k-steps across the queries of a tile (dV, dK)
  constexpr int NQF = TQ / 16;          // 16-query fragments per tile
  constexpr int NF = D / 16;            // 16-wide fragments of the head dim
  constexpr int CPR = D * SZ / 16;      // chunks per row of the row-major tiles
  constexpr int CPT = TQ * SZ / 16;     // chunks per row of the transposed images
  constexpr int TILE_B = TQ * D * SZ;
  static_assert(TQ % KSTEP == 0 && CPT == 4 * KS_Q, "tile shape");
  constexpr bool TR = sizeof(T) == 2;   // bf16: transposed operands are read out of the row-major images (tr_frag); fp32 keeps transposed images
  constexpr int NIMG = TR ? 2 : 4;
  __shared__ __attribute__((aligned(16))) unsigned char smem[NIMG * TILE_B + 2 * TQ * 4];
  unsigned char* Qs = smem;
  unsigned char* dOs = smem + TILE_B;
  unsigned char* QT = smem + (TR ? 0 : 2) * TILE_B;      // (unused when TR)
  unsigned char* dOT = smem + (TR ? 0 : 3) * TILE_B;
  float* lse_s = (float*)(smem + NIMG * TILE_B);
  float* dl_s = lse_s + TQ;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, qd = lane >> 4;
  const int seq = bz, kvh = by;
  const int rep = p.n_heads / p.n_kv;
  const int q0 = p.cu_q[seq], qlen = p.cu_q[seq + 1] - q0;
  const int klen = p.klen[seq];
  constexpr int BKEYS = 64 * KF;        // keys per block
  const int key0 = bx * BKEYS;
  if (key0 >= klen) return;
  const int shift = klen - qlen;        // causal: key j visible to query i iff j <= i + shift
  const int64_t krow0 = p.cu_k[seq];

  // this wave's KF x 16 keys: K and V fragments (B operands of S = Q K^T and dP = dO V^T)
  int kj[KF];
  uint4 kf[KF][KS_D], vf[KF][KS_D];
#pragma unroll
  for (int f = 0; f < KF; ++f) {
    kj[f] = key0 + (wave * KF + f) * 16 + r;
    const int kc = kj[f] < klen ? kj[f] : klen - 1;
    const T* kp = (const T*)p.k + (krow0 + kc) * p.k_rs + (int64_t)kvh * p.k_hs + qd * VEC;
    const T* vp = (const T*)p.v + (krow0 + kc) * p.v_rs + (int64_t)kvh * p.v_hs + qd * VEC;
#pragma unroll
    for (int s = 0; s < KS_D; ++s) { kf[f][s] = *(const uint4*)(kp + s * KSTEP); vf[f][s] = *(const uint4*)(vp + s * KSTEP); }
  }
  f32x4 dk[KF][NF], dv[KF][NF];
#pragma unroll
  for (int f = 0; f < KF; ++f)
#pragma unroll
    for (int n = 0; n < NF; ++n) { dk[f][n] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[f][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }

  int qt_first = 0;
  if (CAUSAL) { const int i0 = key0 - shift; qt_first = i0 > 0 ? i0 / TQ : 0; }   // first query that sees a key of this block
  const int n_qt = (qlen + TQ - 1) / TQ;

  // One thread stages one 16-byte column chunk of two consecutive rows of Q and of dO per tile ((TQ / 2) x CPR = 256 pairs in every
  // instantiation).  The NEXT tile's rows are requested before the current tile's products start (round 5: the loop used to load, store
  // and compute in sequence, with only other blocks to cover the global-load latency).
  static_assert((TQ / 2) * CPR == 256, "one (row pair, chunk) per thread");
  const int n_tiles = n_qt > qt_first ? rep * (n_qt - qt_first) : 0;
  const int prow = 2 * (tid / CPR), pch = tid % CPR;
  uint4 uq0, uq1, ud0, ud1;
  float lse_r = 0.f, dl_r = 0.f;
  auto gload = [&](int it) {
    const int h = it / (n_qt - qt_first), qt = qt_first + it % (n_qt - qt_first);
    const int head = kvh * rep + h, qt0 = qt * TQ;
    const T* qb = (const T*)p.q + (int64_t)head * p.q_hs;
    const T* dob = (const T*)p.dout + (int64_t)head * p.do_hs;
    int qi0 = qt0 + prow, qi1 = qi0 + 1;
    qi0 = qi0 < qlen ? qi0 : qlen - 1; qi1 = qi1 < qlen ? qi1 : qlen - 1;
    uq0 = *(const uint4*)(qb + (int64_t)(q0 + qi0) * p.q_rs + pch * VEC); uq1 = *(const uint4*)(qb + (int64_t)(q0 + qi1) * p.q_rs + pch * VEC);
    ud0 = *(const uint4*)(dob + (int64_t)(q0 + qi0) * p.do_rs + pch * VEC); ud1 = *(const uint4*)(dob + (int64_t)(q0 + qi1) * p.do_rs + pch * VEC);
    if (tid < TQ) {
      const int qi = qt0 + tid;
      const int64_t idx = ((int64_t)q0 + (qi < qlen ? qi : qlen - 1)) * p.n_heads + head;
      lse_r = p.lse[idx];
      dl_r = p.delta[idx];
    }
  };
  if (n_tiles > 0) gload(0);
  for (int it = 0; it < n_tiles; ++it) {
    const int h = it / (n_qt - qt_first), qt = qt_first + it % (n_qt - qt_first);
    const int head = kvh * rep + h;
    {
      const int qt0 = qt * TQ;
      __syncthreads();   // the previous tile's LDS reads are done
      // row-major images (A operands of S / dP; bf16: also the source of the transposed B operands of dK / dV, tr_frag).  fp32: transposed
      // images beside them — the two rows a thread holds are neighbours there, so an element pair goes out as one 8-byte store
      {
        const int row = prow, ch = pch;
        *(uint4*)(Qs + swz<CPR>(row, ch)) = uq0; *(uint4*)(Qs + swz<CPR>(row + 1, ch)) = uq1;
        *(uint4*)(dOs + swz<CPR>(row, ch)) = ud0; *(uint4*)(dOs + swz<CPR>(row + 1, ch)) = ud1;
        if constexpr (!TR) {
          const int pos = tpos<T>(row);        // even, and tpos(row + 1) = pos + 1
#pragma unroll
          for (int j = 0; j < VEC; ++j) {
            const int d = ch * VEC + j;
            const int off = swz<CPT>(d, pos / VEC) + (pos % VEC) * SZ;
            *(float2*)(QT + off) = make_float2(chunk_elem<float>(uq0, j), chunk_elem<float>(uq1, j));
            *(float2*)(dOT + off) = make_float2(chunk_elem<float>(ud0, j), chunk_elem<float>(ud1, j));
          }
        }
      }
      if (tid < TQ) { lse_s[tid] = lse_r; dl_s[tid] = dl_r; }
      if (it + 1 < n_tiles) gload(it + 1);      // in flight under this tile's products
      __syncthreads();

      // S = Q K^T and dP = dO V^T for TQ queries x this wave's 16 keys: lane (key r, a = qd) holds queries n*16 + 4a + i
      f32x4 s[KF][NQF], dp[KF][NQF];
#pragma unroll
      for (int n = 0; n < NQF; ++n) {
#pragma unroll
        for (int f = 0; f < KF; ++f) { s[f][n] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[f][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int ks = 0; ks < KS_D; ++ks) {
          const uint4 aq = *(const uint4*)(Qs + swz<CPR>(n * 16 + r, ks * 4 + qd));
          const uint4 ad = *(const uint4*)(dOs + swz<CPR>(n * 16 + r, ks * 4 + qd));
#pragma unroll
          for (int f = 0; f < KF; ++f) {
            MMA<T>::step(s[f][n], aq, kf[f][ks]);
            MMA<T>::step(dp[f][n], ad, vf[f][ks]);
          }
        }
      }
      // probabilities (recomputed from the forward's lse), dropout mask, dS.  The mask's index is ((query row * heads + head) << 16) | key: over a
      // tile's queries its upper half takes at most two values, so the inner hash is taken twice per tile instead of once per score
      const int64_t a_first = ((int64_t)q0 + qt0) * p.n_heads + head;
      const uint32_t hi_base = (uint32_t)(a_first >> 16);
      uint32_t inner0 = 0, inner1 = 0;
      if (p.drop_thr) { inner0 = drop_inner(hi_base, p.drop_seed); inner1 = drop_inner(hi_base + 1, p.drop_seed); }
#pragma unroll
      for (int n = 0; n < NQF; ++n) {
        const f32x4 ls = *(const f32x4*)(lse_s + n * 16 + 4 * qd);
        const f32x4 dl = *(const f32x4*)(dl_s + n * 16 + 4 * qd);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int qi = qt0 + n * 16 + 4 * qd + i;
          const int64_t a_q = ((int64_t)q0 + qi) * p.n_heads + head;            // (rows past qlen: masked by `ok`, any mask bit will do)
          const uint32_t inner = (uint32_t)(a_q >> 16) == hi_base ? inner0 : inner1;
#pragma unroll
          for (int f = 0; f < KF; ++f) {
            bool ok = kj[f] < klen && qi < qlen;
            if (CAUSAL) ok = ok && (kj[f] <= qi + shift);
            float pv = ok ? exp_scaled<T>(s[f][n][i] * p.scale - ls[i]) : 0.f;
            float dpv = dp[f][n][i];
            float pd = pv;
            if (p.drop_thr) {
              const uint32_t lo = ((uint32_t)a_q << 16) | (uint32_t)kj[f];
              const bool keep = dropout_keep_lo(lo, inner, p.drop_thr);
              pd = keep ? pv * p.drop_scale : 0.f;
              dpv = keep ? dpv * p.drop_scale : 0.f;
            }
            s[f][n][i] = pd;                                    // Pd  -> dV += Pd^T dO
            dp[f][n][i] = pv * (dpv - dl[i]) * p.scale;         // dS (scale folded) -> dK += dS^T Q
          }
        }
      }
#pragma unroll
      for (int ks = 0; ks < KS_Q; ++ks) {
        uint4 apd[KF], ads[KF];
#pragma unroll
        for (int f = 0; f < KF; ++f) { apd[f] = pack_step<T, NQF>(s[f], ks); ads[f] = pack_step<T, NQF>(dp[f], ks); }
        if constexpr (TR) {
          const uint32_t img_d = (uint32_t)(uintptr_t)(lds_ptr_b_t)dOs, img_q = (uint32_t)(uintptr_t)(lds_ptr_b_t)Qs;
#pragma unroll
          for (int n0 = 0; n0 < NF; n0 += 4) {      // four dim fragments of dO and of Q in flight, one wait, eight products
            u32x2_b_t lo[8], hi[8];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              tr_issue<CPR>(img_d, 32 * ks, n0 + g, lane, lo[g], hi[g]);
              tr_issue<CPR>(img_q, 32 * ks, n0 + g, lane, lo[4 + g], hi[4 + g]);
            }
            tr_wait<8>(lo, hi);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
              for (int f = 0; f < KF; ++f) {
                MMA<T>::step(dv[f][n0 + g], apd[f], make_uint4(lo[g].x, lo[g].y, hi[g].x, hi[g].y));
                MMA<T>::step(dk[f][n0 + g], ads[f], make_uint4(lo[4 + g].x, lo[4 + g].y, hi[4 + g].x, hi[4 + g].y));
              }
          }
        } else {
#pragma unroll
          for (int n = 0; n < NF; ++n) {
            const uint4 bd = *(const uint4*)(dOT + swz<CPT>(n * 16 + r, ks * 4 + qd));
            const uint4 bq = *(const uint4*)(QT + swz<CPT>(n * 16 + r, ks * 4 + qd));
#pragma unroll
            for (int f = 0; f < KF; ++f) {
              MMA<T>::step(dv[f][n], apd[f], bd);
              MMA<T>::step(dk[f][n], ads[f], bq);
            }
          }
        }
      }
    }
  }
  // epilogue: lane (dim col r, a = qd) holds keys 4a + i of this wave, dim n*16 + r
  T* dkb = (T*)p.dk + (int64_t)kvh * p.dk_hs;
  T* dvb = (T*)p.dv + (int64_t)kvh * p.dv_hs;
#pragma unroll
  for (int f = 0; f < KF; ++f)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int kk = key0 + (wave * KF + f) * 16 + 4 * qd + i;
      if (kk >= klen) continue;
      T* dkr = dkb + (krow0 + kk) * p.dk_rs;
      T* dvr = dvb + (krow0 + kk) * p.dv_rs;
#pragma unroll
      for (int n = 0; n < NF; ++n) {
        dkr[n * 16 + r] = from_f32<T>(dk[f][n][i]);
        dvr[n * 16 + r] = from_f32<T>(dv[f][n][i]);
      }
    }
}

// QF = 16-query fragments per wave (round 6: 2, a block owns 128 queries): every K / V fragment read from LDS feeds QF products
template <typename T, int D, bool CAUSAL, int TQ, int KF>
__global__ __launch_bounds__(256) void attn_bwd_dkdv_kernel(AttnBwdP p) {
  attn_bwd_dkdv_body<T, D, CAUSAL, TQ, KF>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

template <typename T, int D, bool CAUSAL, int TK, int QF>
__device__ __forceinline__ void attn_bwd_dq_body(const AttnBwdP& p, const int bx, const int by, const int bz) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int SZ = (int)sizeof(T);
  constexpr int KSTEP = MMA<T>::KSTEP;
  constexpr int KS_D = D / KSTEP;
  constexpr int KS_K = TK / KSTEP;      // k-steps across the keys of a tile (dQ)
  constexpr int NKF = TK / 16;
  constexpr int NF = D / 16;
  constexpr int CPR = D * SZ / 16;
  constexpr int CPT = TK * SZ / 16;
  constexpr int TILE_B = TK * D * SZ;
  constexpr bool TR = sizeof(T) == 2;   // bf16: K^T fragments are read out of the row-major K image (tr_frag)
  __shared__ __attribute__((aligned(16))) unsigned char smem[(TR ? 2 : 3) * TILE_B];
  unsigned char* Ks = smem;
  unsigned char* Vs = smem + TILE_B;
  unsigned char* KT = smem + (TR ? 0 : 2) * TILE_B;     // (unused when TR)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, qd = lane >> 4;
  const int seq = bz, head = by;
  const int kvh = head / (p.n_heads / p.n_kv);
  const int q0 = p.cu_q[seq], qlen = p.cu_q[seq + 1] - q0;
  const int klen = p.klen[seq];
  constexpr int BQ = 64 * QF;           // queries per block
  const int qt0 = bx * BQ;
  if (qt0 >= qlen) return;
  const int shift = klen - qlen;
  const int64_t krow0 = p.cu_k[seq];
  const T* kb = (const T*)p.k + krow0 * p.k_rs + (int64_t)kvh * p.k_hs;
  const T* vb = (const T*)p.v + krow0 * p.v_rs + (int64_t)kvh * p.v_hs;

  // this wave's 16 queries: q and dO fragments (B operands of S^T = K Q^T and dP^T = V dO^T), lse and delta of query r
  int qi[QF];
  uint4 qf[QF][KS_D], dof[QF][KS_D];
  float lse_q[QF], dl_q[QF];
  int64_t a_q[QF];
  uint32_t drop_in[QF];
  f32x4 dq[QF][NF];
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    qi[f] = qt0 + (wave * QF + f) * 16 + r;
    const int qc = qi[f] < qlen ? qi[f] : qlen - 1;
    const T* qp = (const T*)p.q + (int64_t)(q0 + qc) * p.q_rs + (int64_t)head * p.q_hs + qd * VEC;
    const T* dp_ = (const T*)p.dout + (int64_t)(q0 + qc) * p.do_rs + (int64_t)head * p.do_hs + qd * VEC;
#pragma unroll
    for (int s = 0; s < KS_D; ++s) { qf[f][s] = *(const uint4*)(qp + s * KSTEP); dof[f][s] = *(const uint4*)(dp_ + s * KSTEP); }
    const int64_t lidx = ((int64_t)q0 + qc) * p.n_heads + head;
    lse_q[f] = p.lse[lidx]; dl_q[f] = p.delta[lidx];
    // dropout mask index = ((query row * heads + head) << 16) | key: its upper half is a constant of this lane's query
    a_q[f] = lidx;
    drop_in[f] = p.drop_thr ? drop_inner((uint32_t)(a_q[f] >> 16), p.drop_seed) : 0u;
#pragma unroll
    for (int n = 0; n < NF; ++n) dq[f][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  int n_kt = (klen + TK - 1) / TK;
  if (CAUSAL) {
    const int last = qt0 + BQ - 1 + shift;   // largest key index any query of this block sees
    const int lim = last < 0 ? 0 : last / TK + 1;
    n_kt = lim < n_kt ? lim : n_kt;
  }
  // staging as in the dK / dV kernel: a thread holds NP (row pair, chunk) items of K and V, the next key tile's rows are requested before the
  // current tile's products start, and the transposed K image is written two rows (one 4- / 8-byte store) at a time
  constexpr int NP = (TK / 2) * CPR / 256;
  static_assert(NP * 256 == (TK / 2) * CPR && NP >= 1, "whole (row pair, chunk) items per thread");
  uint4 uk0[NP], uk1[NP], uv0[NP], uv1[NP];
  auto gload = [&](int kt) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = tid + 256 * i, row = 2 * (c / CPR), ch = c % CPR;
      int k0r = kt * TK + row, k1r = k0r + 1;
      k0r = k0r < klen ? k0r : klen - 1; k1r = k1r < klen ? k1r : klen - 1;
      uk0[i] = *(const uint4*)(kb + (int64_t)k0r * p.k_rs + ch * VEC); uk1[i] = *(const uint4*)(kb + (int64_t)k1r * p.k_rs + ch * VEC);
      uv0[i] = *(const uint4*)(vb + (int64_t)k0r * p.v_rs + ch * VEC); uv1[i] = *(const uint4*)(vb + (int64_t)k1r * p.v_rs + ch * VEC);
    }
  };
  if (n_kt > 0) gload(0);
  for (int kt = 0; kt < n_kt; ++kt) {
    const int key0 = kt * TK;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int c = tid + 256 * i, row = 2 * (c / CPR), ch = c % CPR;
      *(uint4*)(Ks + swz<CPR>(row, ch)) = uk0[i]; *(uint4*)(Ks + swz<CPR>(row + 1, ch)) = uk1[i];
      *(uint4*)(Vs + swz<CPR>(row, ch)) = uv0[i]; *(uint4*)(Vs + swz<CPR>(row + 1, ch)) = uv1[i];
      if constexpr (!TR) {
        const int pos = tpos<T>(row);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          const int d = ch * VEC + j;
          const int off = swz<CPT>(d, pos / VEC) + (pos % VEC) * SZ;
          *(float2*)(KT + off) = make_float2(chunk_elem<float>(uk0[i], j), chunk_elem<float>(uk1[i], j));
        }
      }
    }
    if (kt + 1 < n_kt) gload(kt + 1);      // in flight under this tile's products
    __syncthreads();

    // S^T = K Q^T, dP^T = V dO^T: lane (query r, a = qd) holds keys n*16 + 4a + i
    f32x4 s[QF][NKF], dp[QF][NKF];
#pragma unroll
    for (int n = 0; n < NKF; ++n) {
#pragma unroll
      for (int f = 0; f < QF; ++f) { s[f][n] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[f][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < KS_D; ++ks) {
        const uint4 ak = *(const uint4*)(Ks + swz<CPR>(n * 16 + r, ks * 4 + qd));
        const uint4 av = *(const uint4*)(Vs + swz<CPR>(n * 16 + r, ks * 4 + qd));
#pragma unroll
        for (int f = 0; f < QF; ++f) {
          MMA<T>::step(s[f][n], ak, qf[f][ks]);
          MMA<T>::step(dp[f][n], av, dof[f][ks]);
        }
      }
    }
#pragma unroll
    for (int f = 0; f < QF; ++f)
#pragma unroll
      for (int n = 0; n < NKF; ++n)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int kj = key0 + n * 16 + 4 * qd + i;
          bool ok = kj < klen && qi[f] < qlen;
          if (CAUSAL) ok = ok && (kj <= qi[f] + shift);
          const float pv = ok ? exp_scaled<T>(s[f][n][i] * p.scale - lse_q[f]) : 0.f;
          float dpv = dp[f][n][i];
          if (p.drop_thr) {
            const bool keep = dropout_keep_lo(((uint32_t)a_q[f] << 16) | (uint32_t)kj, drop_in[f], p.drop_thr);
            dpv = keep ? dpv * p.drop_scale : 0.f;
          }
          dp[f][n][i] = pv * (dpv - dl_q[f]) * p.scale;
        }
    // dQ^T += K^T dS^T
#pragma unroll
    for (int ks = 0; ks < KS_K; ++ks) {
      uint4 bds[QF];
#pragma unroll
      for (int f = 0; f < QF; ++f) bds[f] = pack_step<T, NKF>(dp[f], ks);
      if constexpr (TR) {
        const uint32_t img_k = (uint32_t)(uintptr_t)(lds_ptr_b_t)Ks;
#pragma unroll
        for (int n0 = 0; n0 < NF; n0 += 4) {
          u32x2_b_t lo[4], hi[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) tr_issue<CPR>(img_k, 32 * ks, n0 + g, lane, lo[g], hi[g]);
          tr_wait<4>(lo, hi);
#pragma unroll
          for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int f = 0; f < QF; ++f) MMA<T>::step(dq[f][n0 + g], make_uint4(lo[g].x, lo[g].y, hi[g].x, hi[g].y), bds[f]);
        }
      } else {
#pragma unroll
        for (int n = 0; n < NF; ++n) {
          const uint4 ak = *(const uint4*)(KT + swz<CPT>(n * 16 + r, ks * 4 + qd));
#pragma unroll
          for (int f = 0; f < QF; ++f) MMA<T>::step(dq[f][n], ak, bds[f]);
        }
      }
    }
  }
  // epilogue: lane (query r, a = qd) holds dims n*16 + 4a + i
#pragma unroll
  for (int f = 0; f < QF; ++f) {
    if (qi[f] >= qlen) continue;
    T* dqr = (T*)p.dq + (int64_t)(q0 + qi[f]) * p.dq_rs + (int64_t)head * p.dq_hs + 4 * qd;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      if constexpr (sizeof(T) == 2) {
        *(uint2*)(dqr + n * 16) = make_uint2(pack2_bf16(dq[f][n][0], dq[f][n][1]), pack2_bf16(dq[f][n][2], dq[f][n][3]));
      } else {
        *(f32x4*)(dqr + n * 16) = dq[f][n];
      }
    }
  }
}

template <typename T, int D, bool CAUSAL, int TK, int QF>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnBwdP p) {
  attn_bwd_dq_body<T, D, CAUSAL, TK, QF>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Both passes in ONE launch (round 6): the first n_dkdv blocks are the dK / dV blocks, the rest the dQ blocks.  The two passes are independent (they
// read the same q / k / v / dO / lse / delta and write disjoint outputs); in the per-rank KD window (2 sequences) each alone fills a third of the
// chip — 80 dK / dV blocks for Llama's 8 kv heads — and they ran back to back.  Same block bodies, same bits.  The LDS of the two bodies adds up
// (static arrays: 33-49 KiB), which is why the large launches of a 16-sample window, where each pass fills the chip by itself, keep their own kernels.
template <typename T, int D, bool CAUSAL, int TQ, int TK>
__global__ __launch_bounds__(256) void attn_bwd_both_kernel(AttnBwdP p, int kx, int ky, int n_dkdv, int qx, int qy) {
  int b = blockIdx.x;
  if (b < n_dkdv) {
    const int bx = b % kx; b /= kx;
    attn_bwd_dkdv_body<T, D, CAUSAL, TQ, 1>(p, bx, b % ky, b / ky);
  } else {
    b -= n_dkdv;
    const int bx = b % qx; b /= qx;
    attn_bwd_dq_body<T, D, CAUSAL, TK, 1>(p, bx, b % qy, b / qy);
  }
}

template <typename T, int D, bool CAUSAL>
int launch_attn_bwd(const sl_attn_bwd_args* a, const AttnBwdP& p, hipStream_t st) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int LPP = D / VEC;
  const int64_t pairs = a->n_tok_q * a->n_heads;
  hipLaunchKernelGGL((attn_bwd_delta_kernel<T, D>), dim3((unsigned)((pairs * LPP + 255) / 256)), dim3(256), 0, st, p);
  SL_CHECK_LAUNCH("attn_bwd_delta");
  // tile sizes keep every kernel inside 64 KiB of static LDS
  constexpr int TQ = sizeof(T) == 2 ? (D == 64 ? 64 : 32) : (D == 64 ? 32 : 16);
  constexpr int TK = sizeof(T) == 2 ? 64 : 32;
  // keys (queries) per block: 128 in bf16 — two 16-row fragments per wave share every LDS operand read; fp32 (parity mode) keeps one.
  // SL_ATTN_BWD_KF=1: the one-fragment kernels (A/B; same bits)
  // Measured (profiles/r06_e_attn_bwd_ab.txt): the two-fragment form needs 320-360 registers, one wave per SIMD.  Llama's shape (head_dim 128, causal)
  // gains where its 128-key blocks still fill the chip (16 x 200: 140 -> 113 us per layer); HuBERT's (head_dim 64: 201 -> 238 us) and the per-rank
  // window's few blocks (2 x 317: 83 -> 99 us) lose the second wave that hid their latencies.  SL_ATTN_BWD_KF = 1 / 2 forces a form.
  if constexpr (sizeof(T) == 2) {
    const int64_t blocks2 = (int64_t)((a->max_klen + 127) / 128) * a->n_kv_heads * a->nseq;
    const int kf = sl_env().attn_bwd_kf;
    if (kf == 2 || (kf != 1 && D == 128 && blocks2 >= 192)) {
      constexpr int TK2 = D == 64 ? 64 : 32;      // (head_dim 128 with two query fragments per wave: 64-key tiles spill)
      hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, D, CAUSAL, TQ, 2>), dim3((a->max_klen + 127) / 128, a->n_kv_heads, a->nseq), dim3(256), 0, st, p);
      SL_CHECK_LAUNCH("attn_bwd_dkdv");
      hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, CAUSAL, TK2, 2>), dim3((a->max_qlen + 127) / 128, a->n_heads, a->nseq), dim3(256), 0, st, p);
      SL_CHECK_LAUNCH("attn_bwd_dq");
      return 0;
    }
  }
  if constexpr (sizeof(T) == 2) {
    const int kx = (a->max_klen + 63) / 64, qx = (a->max_qlen + 63) / 64;
    const int64_t n_dkdv = (int64_t)kx * a->n_kv_heads * a->nseq, n_dq = (int64_t)qx * a->n_heads * a->nseq;
    if (sl_env().attn_bwd_both == 2 || (sl_env().attn_bwd_both && n_dkdv + n_dq <= 1024)) {      // neither pass fills the chip (256 CUs x 2-4 blocks) by itself; 2 = always (A/B)
      hipLaunchKernelGGL((attn_bwd_both_kernel<T, D, CAUSAL, TQ, TK>), dim3((unsigned)(n_dkdv + n_dq)), dim3(256), 0, st, p, kx, a->n_kv_heads, (int)n_dkdv, qx,
                         a->n_heads);
      SL_CHECK_LAUNCH("attn_bwd_both");
      return 0;
    }
  }
  hipLaunchKernelGGL((attn_bwd_dkdv_kernel<T, D, CAUSAL, TQ, 1>), dim3((a->max_klen + 63) / 64, a->n_kv_heads, a->nseq), dim3(256), 0, st, p);
  SL_CHECK_LAUNCH("attn_bwd_dkdv");
  hipLaunchKernelGGL((attn_bwd_dq_kernel<T, D, CAUSAL, TK, 1>), dim3((a->max_qlen + 63) / 64, a->n_heads, a->nseq), dim3(256), 0, st, p);
  SL_CHECK_LAUNCH("attn_bwd_dq");
  return 0;
}

}  // namespace

extern "C" int sl_attn_bwd(const sl_attn_bwd_args* a, sl_stream stream) {
  SL_CHECK_ARG(a && a->q && a->k && a->v && a->out && a->d_out && a->dq && a->dk && a->dv && a->lse && a->delta && a->cu_q && a->cu_k && a->klen,
               "sl_attn_bwd: null pointer");
  SL_CHECK_ARG(a->nseq > 0 && a->max_qlen > 0 && a->max_klen > 0 && a->n_tok_q > 0 && a->n_heads > 0 && a->n_kv_heads > 0 &&
                   a->n_heads % a->n_kv_heads == 0, "sl_attn_bwd: bad shape");
  SL_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f, "sl_attn_bwd: dropout_p=%f outside [0, 1)", (double)a->dropout_p);
  const int vec = a->dtype == SL_F32 ? 4 : 8;
  const int64_t strides[] = {a->q_row_stride, a->q_head_stride, a->k_row_stride, a->k_head_stride, a->v_row_stride, a->v_head_stride,
                             a->o_row_stride, a->o_head_stride, a->do_row_stride, a->do_head_stride};
  for (int64_t s : strides) SL_CHECK_ARG(s % vec == 0, "sl_attn_bwd: input strides must keep 16-byte alignment");
  SL_CHECK_ARG(a->dq_row_stride % 4 == 0 && a->dq_head_stride % 4 == 0 && ((uintptr_t)a->dq & 15) == 0,
               "sl_attn_bwd: dq rows / heads must start on 4-element boundaries of a 16-byte aligned buffer");
  AttnBwdP p;
  p.q = a->q; p.q_rs = a->q_row_stride; p.q_hs = a->q_head_stride;
  p.k = a->k; p.k_rs = a->k_row_stride; p.k_hs = a->k_head_stride;
  p.v = a->v; p.v_rs = a->v_row_stride; p.v_hs = a->v_head_stride;
  p.o = a->out; p.o_rs = a->o_row_stride; p.o_hs = a->o_head_stride;
  p.dout = a->d_out; p.do_rs = a->do_row_stride; p.do_hs = a->do_head_stride;
  p.dq = a->dq; p.dq_rs = a->dq_row_stride; p.dq_hs = a->dq_head_stride;
  p.dk = a->dk; p.dk_rs = a->dk_row_stride; p.dk_hs = a->dk_head_stride;
  p.dv = a->dv; p.dv_rs = a->dv_row_stride; p.dv_hs = a->dv_head_stride;
  p.lse = a->lse; p.delta = a->delta;
  p.cu_q = a->cu_q; p.cu_k = a->cu_k; p.klen = a->klen;
  p.n_tok_q = a->n_tok_q;
  p.n_heads = a->n_heads; p.n_kv = a->n_kv_heads; p.scale = a->scale;
  p.drop_thr = 0; p.drop_scale = 1.f; p.drop_seed = a->dropout_seed;
  if (a->dropout_p > 0.f) {
    p.drop_thr = (uint32_t)((double)a->dropout_p * 16777216.0);
    p.drop_scale = 1.0f / (1.0f - a->dropout_p);
  }
  hipStream_t st = (hipStream_t)stream;
  SL_DISPATCH_DTYPE(a->dtype, T, {
    if (a->head_dim == 64) return a->causal ? launch_attn_bwd<T, 64, true>(a, p, st) : launch_attn_bwd<T, 64, false>(a, p, st);
    if (a->head_dim == 128) return a->causal ? launch_attn_bwd<T, 128, true>(a, p, st) : launch_attn_bwd<T, 128, false>(a, p, st);
    sl_set_error("sl_attn_bwd: head_dim %d not built (64, 128)", a->head_dim);
    return SL_ERR_UNSUPPORTED;
  });
}
