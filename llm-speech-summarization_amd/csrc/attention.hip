// attention.hip — flash-style attention forward on MFMA for packed variable-length sequences.
//   HuBERT: head_dim 64, bidirectional, no mask (hf:models/hubert/modeling_hubert.py:234-259)
//   Llama prefill: head_dim 128, causal, GQA (hf:models/llama/modeling_llama.py:191-213)
// Block = 4 waves x 16 query rows; 64-key K and V^T tiles in XOR-swizzled LDS; S = Q.K^T and O += P.V
// both run on the dtype-generic 16x16 MFMA step (bf16: 16x16x32, fp32: exact 16x16x4), softmax state
// (running max / sum) in fp32 registers, scores never leave the CU.
#include <stdlib.h>
#include "common.h"

struct AttnP {
  const void* q; int64_t q_rs, q_hs;
  const void* k; int64_t k_rs, k_hs;
  const void* v; int64_t v_rs, v_hs;
  void* o; int64_t o_rs, o_hs;
  const int32_t* cu_q; const int32_t* cu_k; const int32_t* klen;
  int n_heads, n_kv;
  float scale;
  uint32_t drop_thr;     // training mode: keep iff u24(hash) >= drop_thr (0 = no dropout); see sl_attn_args.dropout_p
  float drop_scale;
  uint64_t drop_seed;
  float* lse;            // training mode: log-sum-exp per (query row, head), natural log (NULL = not wanted)
};

// swizzled byte offset of 16-byte chunk `ch` of row `row`; rows hold `cpr` chunks (8, 16 or 32)
template <int CPR>
__device__ __forceinline__ int swz_off(int row, int ch) {
  constexpr int MASK = (CPR < 16 ? CPR : 16) - 1;
  return row * (CPR * 16) + ((ch ^ (row & MASK)) << 4);
}

template <typename T, int D, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnP p) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int SZ = (int)sizeof(T);
  constexpr int KSTEP = MMA<T>::KSTEP;
  constexpr int KS_D = D / KSTEP;     // k-steps across the head dim (S = Q.K^T)
  constexpr int KS_P = 64 / KSTEP;    // k-steps across the 64 keys of a tile (O += P.V)
  constexpr int NF_O = D / 16;        // output column fragments
  constexpr int CPR_K = D * SZ / 16;  // 16-byte chunks per K row
  constexpr int CPR_V = 64 * SZ / 16; // chunks per V^T row / P row
  constexpr int K_BYTES = 64 * D * SZ, V_BYTES = D * 64 * SZ, P_BYTES = 16 * 64 * SZ;
  __shared__ __attribute__((aligned(16))) unsigned char smem[K_BYTES + V_BYTES + 4 * P_BYTES];
  unsigned char* Ks = smem;
  unsigned char* Vt = smem + K_BYTES;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, qd = lane >> 4;
  unsigned char* Ps = smem + K_BYTES + V_BYTES + wave * P_BYTES;

  const int seq = blockIdx.z, head = blockIdx.y;
  const int kvh = head / (p.n_heads / p.n_kv);
  const int q0 = p.cu_q[seq], qlen = p.cu_q[seq + 1] - q0;
  const int klen = p.klen[seq];
  const int qt0 = blockIdx.x * 64;
  if (qt0 >= qlen) return;
  const int shift = klen - qlen;  // causal: key j visible to query i iff j <= i + shift

  const T* qb = (const T*)p.q + (int64_t)head * p.q_hs;
  const T* kb = (const T*)p.k + (int64_t)p.cu_k[seq] * p.k_rs + (int64_t)kvh * p.k_hs;
  const T* vb = (const T*)p.v + (int64_t)p.cu_k[seq] * p.v_rs + (int64_t)kvh * p.v_hs;

  // Q fragments for this wave's 16 rows
  uint4 qf[KS_D];
  {
    int qi = qt0 + wave * 16 + r;
    qi = qi < qlen ? qi : qlen - 1;
    const T* qp = qb + (int64_t)(q0 + qi) * p.q_rs + qd * VEC;
#pragma unroll
    for (int s = 0; s < KS_D; ++s) qf[s] = *(const uint4*)(qp + s * KSTEP);
  }

  f32x4 o[NF_O];
#pragma unroll
  for (int n = 0; n < NF_O; ++n) o[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[4], l_run[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { m_run[i] = -INFINITY; l_run[i] = 0.f; }

  int nkt = (klen + 63) >> 6;
  if (CAUSAL) {
    const int last = qt0 + 63 + shift;  // largest visible key index of this q tile
    const int lim = last < 0 ? 0 : (last >> 6) + 1;
    nkt = lim < nkt ? lim : nkt;
  }

  for (int kt = 0; kt < nkt; ++kt) {
    const int key0 = kt * 64;
    __syncthreads();  // previous tile's LDS reads are done
    // K tile: 64 rows x CPR_K chunks
#pragma unroll
    for (int i = 0; i < (64 * CPR_K) / 256; ++i) {
      const int c = tid + 256 * i, row = c / CPR_K, ch = c % CPR_K;
      int kr = key0 + row; kr = kr < klen ? kr : klen - 1;
      *(uint4*)(Ks + swz_off<CPR_K>(row, ch)) = *(const uint4*)(kb + (int64_t)kr * p.k_rs + ch * VEC);
    }
    // V tile, transposed on the way in: Vt[d][key]
#pragma unroll
    for (int i = 0; i < (64 * CPR_K) / 256; ++i) {
      const int c = tid + 256 * i, row = c / CPR_K, ch = c % CPR_K;
      int kr = key0 + row; kr = kr < klen ? kr : klen - 1;
      const uint4 u = *(const uint4*)(vb + (int64_t)kr * p.v_rs + ch * VEC);
      T e[VEC];
      *(uint4*)e = u;
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const int d = ch * VEC + j;
        *(T*)(Vt + swz_off<CPR_V>(d, row / VEC) + (row % VEC) * SZ) = e[j];
      }
    }
    __syncthreads();

    // S = Q.K^T for 16 rows x 64 keys
    f32x4 s[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      s[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS_D; ++ks) {
        const uint4 kfrag = *(const uint4*)(Ks + swz_off<CPR_K>(n * 16 + r, ks * 4 + qd));
        MMA<T>::step(s[n], qf[ks], kfrag);
      }
    }
    // scale, mask, online softmax.  Lane (c = r, qd) holds rows 4*qd + i, cols n*16 + c.
    float mx[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) mx[i] = -INFINITY;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int kj = key0 + n * 16 + r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int qi = qt0 + wave * 16 + qd * 4 + i;
        bool ok = kj < klen;
        if (CAUSAL) ok = ok && (kj <= qi + shift);
        const float v = ok ? s[n][i] * p.scale : -INFINITY;
        s[n][i] = v;
        mx[i] = fmaxf(mx[i], v);
      }
    }
    float alpha[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float m = mx[i];
      m = fmaxf(m, __shfl_xor(m, 8, 64)); m = fmaxf(m, __shfl_xor(m, 4, 64));
      m = fmaxf(m, __shfl_xor(m, 2, 64)); m = fmaxf(m, __shfl_xor(m, 1, 64));
      const float m_new = fmaxf(m_run[i], m);
      const float m_use = m_new == -INFINITY ? 0.f : m_new;
      alpha[i] = __expf(m_run[i] - m_use);
      m_run[i] = m_new;
      mx[i] = m_use;
    }
    float rs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      const int col = n * 16 + r;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float pv = __expf(s[n][i] - mx[i]);
        rs[i] += pv;                                    // the normaliser is the undropped sum (dropout follows the softmax)
        const int row = qd * 4 + i;
        if (p.drop_thr) {
          const int64_t qg = (int64_t)q0 + qt0 + wave * 16 + row;
          pv = dropout_keep((((qg * p.n_heads + head) << 16) | (int64_t)(key0 + col)), p.drop_seed, p.drop_thr) ? pv * p.drop_scale : 0.f;
        }
        *(T*)(Ps + swz_off<CPR_V>(row, col / VEC) + (col % VEC) * SZ) = from_f32<T>(pv);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t = rs[i];
      t += __shfl_xor(t, 8, 64); t += __shfl_xor(t, 4, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 1, 64);
      l_run[i] = l_run[i] * alpha[i] + t;
    }
#pragma unroll
    for (int n = 0; n < NF_O; ++n)
#pragma unroll
      for (int i = 0; i < 4; ++i) o[n][i] *= alpha[i];
    __syncthreads();  // P visible to the whole wave (and keeps the 4 waves in step)

    // O += P.V
#pragma unroll
    for (int ks = 0; ks < KS_P; ++ks) {
      const uint4 pf = *(const uint4*)(Ps + swz_off<CPR_V>(r, ks * 4 + qd));
#pragma unroll
      for (int n = 0; n < NF_O; ++n) {
        const uint4 vf = *(const uint4*)(Vt + swz_off<CPR_V>(n * 16 + r, ks * 4 + qd));
        MMA<T>::step(o[n], pf, vf);
      }
    }
  }

  // epilogue
  T* ob = (T*)p.o + (int64_t)head * p.o_hs;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int qi = qt0 + wave * 16 + qd * 4 + i;
    if (qi >= qlen) continue;
    const float inv = l_run[i] > 0.f ? 1.0f / l_run[i] : 0.f;
    T* orow = ob + (int64_t)(q0 + qi) * p.o_rs;
#pragma unroll
    for (int n = 0; n < NF_O; ++n) orow[n * 16 + r] = from_f32<T>(o[n][i] * inv);
    if (p.lse && r == 0) p.lse[((int64_t)q0 + qi) * p.n_heads + head] = l_run[i] > 0.f ? m_run[i] + logf(l_run[i]) : -INFINITY;
  }
}

// ----------------------------------------------------------------------------------------------
// bf16 fast path: transposed scores.  S^T = K.Q^T puts a QUERY on each lane column (lane (r, q) holds keys 4q..4q+3 of
// query r for every 16-key tile), so
//   * the softmax is lane-local (row max / sum: the lane's own registers + two cross-lane steps over q),
//   * exp(S) packed to bf16 IS the B operand of the second product — the 32 contraction slots of a 16x16x32 step are
//     assigned to keys as (slot 8q+j <-> key 4q+j of the even tile, slot 8q+4+j <-> key 4q+j of the odd tile), and the
//     V^T operand is gathered in that order by ds_read_b64_tr_b16 from a row-major V tile — no P tile in LDS, no
//     element-wise V transpose,
//   * O^T accumulates with the query on the lane too: the running rescale is one multiply per register.
// A wave owns QT x 16 queries (K and V fragments are read once per QT query tiles), a block 4 waves; K/V tiles of 64
// keys are prefetched global -> registers during the products of the previous tile and double-buffered in LDS (one
// barrier per tile).  V image: D = 64 (128-byte rows): chunk ^ (((row >> 1) & 3) << 1); D = 128: image (b) of the
// guide on rows with bits 2 and 3 of the key index swapped, so the two 4-key blocks a 32-lane half gathers lie 8 rows
// apart (both conflict-free for the transposed reads).
// ----------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_a_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_a_t;

template <int D>
__device__ __forceinline__ int v_img_off(int row, int ch) {
  if constexpr (D == 64) return 128 * row + 16 * (ch ^ (((row >> 1) & 3) << 1));
  else {
    const int pr = (row & ~12) | ((row & 4) << 1) | ((row & 8) >> 1);
    return 256 * pr + 16 * (ch ^ (((pr & 3) << 2) | ((pr >> 2) & 3)));
  }
}

// ST = 64-key tiles per staged block of keys.  Round 6 built ST = 2 for head_dim 64 (one barrier and one global -> LDS hand-over per 128 keys instead of
// per 64: the knock-outs of profiles/r06_i_attn_fwd_knockouts.txt put staging + barriers at 22 % of the kernel) and measured it EQUAL to ST = 1
// (1 132-1 154 against 1 133-1 146 us at 512 x 499, encoder pass 107.9 against 107.8 ms, profiles/r06_l_attn_fwd_st_ab.txt): the barrier count is not what
// the staging costs.  Default ST = 1; SL_ATTN_FWD_ST=2 selects the other form.
template <int D, bool CAUSAL, int QT, bool DROP = false, int ST = 1>
__global__ __launch_bounds__(256, 2) void attn_fwd_tr_kernel(AttnP p) {
  using T = bf16_t;
  constexpr int KS_D = D / 32;             // 32-wide steps across the head dim (S^T)
  constexpr int NF_O = D / 16;             // 16-wide output fragments
  constexpr int CPR = D / 8;               // 16-byte chunks per K / V row
  constexpr int TILE_B = 64 * D * 2;       // bytes of one 64-key tile
  constexpr int NLD = ST * (64 * CPR) / 256;   // chunks per thread per staged block of ST tiles
  constexpr int QB = QT * 64;              // queries per block
  __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][ST * TILE_B];   // [buffer][K | V][tile of the block]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int seq = blockIdx.z, head = blockIdx.y;
  const int kvh = head / (p.n_heads / p.n_kv);
  const int q0 = p.cu_q[seq], qlen = p.cu_q[seq + 1] - q0;
  const int klen = p.klen[seq];
  const int qt0 = blockIdx.x * QB;
  if (qt0 >= qlen) return;
  const int shift = klen - qlen;           // causal: key j visible to query i iff j <= i + shift
  const int qw0 = qt0 + wave * (QT * 16);  // this wave's first query
  const bool wave_on = qw0 < qlen;         // wave-uniform; idle waves still stage tiles and meet the barriers

  const T* qb = (const T*)p.q + (int64_t)head * p.q_hs;
  const T* kb = (const T*)p.k + (int64_t)p.cu_k[seq] * p.k_rs + (int64_t)kvh * p.k_hs;
  const T* vb = (const T*)p.v + (int64_t)p.cu_k[seq] * p.v_rs + (int64_t)kvh * p.v_hs;

  uint4 qf[QT][KS_D];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    int qi = qw0 + t * 16 + r;
    qi = qi < qlen ? qi : qlen - 1;
    const T* qp = qb + (int64_t)(q0 + qi) * p.q_rs + q * 8;
#pragma unroll
    for (int ks = 0; ks < KS_D; ++ks) qf[t][ks] = *(const uint4*)(qp + ks * 32);
  }

  f32x4 o[NF_O][QT];
#pragma unroll
  for (int n = 0; n < NF_O; ++n)
#pragma unroll
    for (int t = 0; t < QT; ++t) o[n][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m_run[QT], l_run[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) { m_run[t] = -INFINITY; l_run[t] = 0.f; }

  int nkt = (klen + 63) >> 6;
  if (CAUSAL) {
    const int last = qt0 + QB - 1 + shift;   // largest key index any query of this block sees
    const int lim = last < 0 ? 0 : (last >> 6) + 1;
    nkt = lim < nkt ? lim : nkt;
  }
  const float c = p.scale * 1.4426950408889634f;   // scores in the log2 domain: exp(x) = exp2(x log2 e)
  // training-mode dropout of the probabilities: per query tile of this lane the upper half of the mask index (one inner hash round, taken here
  // once instead of per score: a 32-bit integer multiply is a quarter-rate instruction) and the query's share of the lower half
  uint32_t drop_in[QT], drop_lo[QT];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    drop_in[t] = 0; drop_lo[t] = 0;
    if constexpr (DROP) {
      const int qi = qw0 + t * 16 + r;
      const int64_t a_q = ((int64_t)q0 + (qi < qlen ? qi : qlen - 1)) * p.n_heads + head;
      drop_in[t] = drop_inner((uint32_t)(a_q >> 16), p.drop_seed);
      drop_lo[t] = (uint32_t)a_q << 16;
    }
  }

  u32x4_t kreg[NLD], vreg[NLD];
  auto gload = [&](int sb) {      // staged block sb = tiles [sb ST, sb ST + ST)
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int cidx = tid + 256 * i, row = cidx / CPR, ch = cidx % CPR;
      int kr = sb * (64 * ST) + row; kr = kr < klen ? kr : klen - 1;
      kreg[i] = *(const u32x4_t*)(kb + (int64_t)kr * p.k_rs + ch * 8);
      vreg[i] = *(const u32x4_t*)(vb + (int64_t)kr * p.v_rs + ch * 8);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int cidx = tid + 256 * i, row = cidx / CPR, ch = cidx % CPR;
      *(u32x4_t*)(&smem[buf][0][0] + (row >> 6) * TILE_B + swz_off<CPR>(row & 63, ch)) = kreg[i];
      *(u32x4_t*)(&smem[buf][1][0] + (row >> 6) * TILE_B + v_img_off<D>(row & 63, ch)) = vreg[i];
    }
  };
  if (nkt > 0) { gload(0); sstore(0); }
  __syncthreads();

  const int qq = r >> 2, pp = r & 3;   // transposed read: lane 4 qq + pp of a 16-lane group addresses row qq, 8-byte piece pp
  const int nsb = (nkt + ST - 1) / ST;
  for (int sb = 0; sb < nsb; ++sb) {
#if !defined(SL_ATTN_KO) || SL_ATTN_KO != 4
    const int buf = sb & 1;
    if (sb + 1 < nsb) gload(sb + 1);
#else
    const int buf = 0;                      // knock-out: one staged block, no global loads / LDS stores / barriers in the loop
#endif
   for (int sub = 0; sub < ST; ++sub) {
    const int kt = sb * ST + sub, key0 = kt * 64;
    if (kt >= nkt) break;
    bool active = wave_on;
    if (CAUSAL) active = active && (key0 <= qw0 + QT * 16 - 1 + shift);
    if (active) {
      const unsigned char* Kt = &smem[buf][0][0] + sub * TILE_B;
      const uint32_t vbase = (uint32_t)(uintptr_t)(lds_ptr_a_t)(&smem[buf][1][0] + sub * TILE_B);
      // S^T = K.Q^T: lane (r, q) <- keys n*16 + 4q + i of query r
      f32x4 s[QT][4];
#pragma unroll
      for (int n = 0; n < 4; ++n) {
#pragma unroll
        for (int t = 0; t < QT; ++t) s[t][n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS_D; ++ks) {
          const uint4 kf = *(const uint4*)(Kt + swz_off<CPR>(n * 16 + r, ks * 4 + q));
#pragma unroll
#if !defined(SL_ATTN_KO) || SL_ATTN_KO != 3
          for (int t = 0; t < QT; ++t) MMA<T>::step(s[t][n], kf, qf[t][ks]);
#else
          for (int t = 0; t < QT; ++t) s[t][n][0] += __builtin_bit_cast(float, kf.x & 0x3f800000u) + __builtin_bit_cast(float, qf[t][ks].x & 0x3f800000u);      // knock-out: no S products
#endif
        }
      }
      bool need_mask = key0 + 64 > klen;
      if (CAUSAL) need_mask = need_mask || (key0 + 63 > qw0 + shift);
      uint4 pb[QT][2];
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        const int qi = qw0 + t * 16 + r;
        float mloc = -INFINITY;   // raw scores; the scale (> 0) is folded into the exponent's fma
        if (need_mask) {
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int kj = key0 + n * 16 + 4 * q + i;
              bool ok = kj < klen;
              if (CAUSAL) ok = ok && (kj <= qi + shift);
              const float v = ok ? s[t][n][i] : -INFINITY;
              s[t][n][i] = v;
              mloc = fmaxf(mloc, v);
            }
        } else {
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) mloc = fmaxf(mloc, s[t][n][i]);
        }
#if !defined(SL_ATTN_KO) || SL_ATTN_KO != 5
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
#endif
        const float m_new = fmaxf(m_run[t], mloc * c);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_use);
        m_run[t] = m_new;
        float ls = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#if !defined(SL_ATTN_KO) || SL_ATTN_KO != 1
            float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][n][i], c, -m_use));
#else
            float pv = __builtin_fmaf(s[t][n][i], c, -m_use);      // knock-out: no exp
#endif
            ls += pv;                                   // undropped normaliser
            if constexpr (DROP) {      // mask index ((query row * heads + head) << 16) | key: the inner hash round belongs to the query (drop_in[t])
              pv = dropout_keep_lo(drop_lo[t] | (uint32_t)(key0 + n * 16 + 4 * q + i), drop_in[t], p.drop_thr) ? pv * p.drop_scale : 0.f;
            }
            s[t][n][i] = pv;
          }
        l_run[t] = l_run[t] * alpha + ls;   // per-lane partial sum; the four q lanes of a query meet in the epilogue
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
          pb[t][kk] = make_uint4(pack2_bf16(s[t][2 * kk][0], s[t][2 * kk][1]), pack2_bf16(s[t][2 * kk][2], s[t][2 * kk][3]),
                                 pack2_bf16(s[t][2 * kk + 1][0], s[t][2 * kk + 1][1]), pack2_bf16(s[t][2 * kk + 1][2], s[t][2 * kk + 1][3]));
        // running rescale: skipped (exactly) while no query of the wave has a new maximum; scalar multiplies on purpose —
        // the compiler's v_pk_mul_f32 pairs cost more than two v_mul_f32 beside MFMAs (MI355X_MICROARCH issue-cost table)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
          for (int n = 0; n < NF_O; ++n)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float x = o[n][t][i];
              asm("v_mul_f32 %0, %1, %0" : "+v"(x) : "v"(alpha));
              o[n][t][i] = x;
            }
        }
      }
      // O^T += V^T.P^T: the V^T fragment of 16 dims x (two 16-key tiles) comes transposed out of the row-major V tile
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        u32x2_a_t lo[NF_O], hi[NF_O];
#pragma unroll
        for (int n = 0; n < NF_O; ++n) {
          const int r0 = (2 * kk) * 16 + 4 * q + qq, r1 = r0 + 16;
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo[n]) : "v"(vbase + (uint32_t)(v_img_off<D>(r0, 2 * n + (pp >> 1)) + 8 * (pp & 1))) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi[n]) : "v"(vbase + (uint32_t)(v_img_off<D>(r1, 2 * n + (pp >> 1)) + 8 * (pp & 1))) : "memory");
        }
#pragma unroll
        for (int n = 0; n < NF_O; ++n) {
          if (n == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo[0]), "+v"(hi[0]) : "n"(2 * (NF_O - 1) > 15 ? 15 : 2 * (NF_O - 1)));
          else if (n == NF_O - 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo[NF_O - 1]), "+v"(hi[NF_O - 1]));
          else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(lo[n]), "+v"(hi[n]) : "n"(2 * (NF_O - 1 - n)));
          const uint4 vf = make_uint4(lo[n].x, lo[n].y, hi[n].x, hi[n].y);
#pragma unroll
#if !defined(SL_ATTN_KO) || SL_ATTN_KO != 2
          for (int t = 0; t < QT; ++t) MMA<T>::step(o[n][t], vf, pb[t][kk]);
#else
          for (int t = 0; t < QT; ++t) o[n][t][0] += __builtin_bit_cast(float, vf.x & 0x3f800000u) + __builtin_bit_cast(float, pb[t][kk].x & 0x3f800000u);   // knock-out: no PV products
#endif
        }
      }
    }
   }
#if !defined(SL_ATTN_KO) || SL_ATTN_KO != 4
    if (sb + 1 < nsb) sstore(buf ^ 1);
    __syncthreads();
#endif
  }

  if (!wave_on) return;
  T* ob = (T*)p.o + (int64_t)head * p.o_hs;
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    float l = l_run[t];
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const int qi = qw0 + t * 16 + r;
    if (qi >= qlen) continue;
    const float inv = l > 0.f ? 1.0f / l : 0.f;
    if (p.lse && q == 0)   // m_run is in the log2 domain (scores scaled by scale * log2 e)
      p.lse[((int64_t)q0 + qi) * p.n_heads + head] = l > 0.f ? (m_run[t] + log2f(l)) * 0.6931471805599453f : -INFINITY;
    T* orow = ob + (int64_t)(q0 + qi) * p.o_rs + 4 * q;
#pragma unroll
    for (int n = 0; n < NF_O; ++n)
      *(uint2*)(orow + n * 16) = make_uint2(pack2_bf16(o[n][t][0] * inv, o[n][t][1] * inv), pack2_bf16(o[n][t][2] * inv, o[n][t][3] * inv));
  }
}

template <typename T, int D, bool CAUSAL>
static int launch_attn(const sl_attn_args* a, hipStream_t st) {
  AttnP p;
  p.q = a->q; p.q_rs = a->q_row_stride; p.q_hs = a->q_head_stride;
  p.k = a->k; p.k_rs = a->k_row_stride; p.k_hs = a->k_head_stride;
  p.v = a->v; p.v_rs = a->v_row_stride; p.v_hs = a->v_head_stride;
  p.o = a->out; p.o_rs = a->o_row_stride; p.o_hs = a->o_head_stride;
  p.cu_q = a->cu_q; p.cu_k = a->cu_k; p.klen = a->klen;
  p.n_heads = a->n_heads; p.n_kv = a->n_kv_heads; p.scale = a->scale;
  p.drop_thr = 0; p.drop_scale = 1.f; p.drop_seed = a->dropout_seed;
  p.lse = a->lse;
  if (a->dropout_p > 0.f) {
    p.drop_thr = (uint32_t)((double)a->dropout_p * 16777216.0);
    p.drop_scale = 1.0f / (1.0f - a->dropout_p);
  }
  if constexpr (sizeof(T) == 2) {
    // bf16: transposed-score kernel (8-byte output vectors need 4-element strides / an 8-byte aligned base)
    const int generic = sl_env().attn_generic;
    if (!generic && a->o_row_stride % 4 == 0 && a->o_head_stride % 4 == 0 && ((uintptr_t)a->out & 7) == 0) {
      if (p.drop_thr) {   // training mode (HuBERT's attention dropout): one variant (32 queries per wave) with the mask applied to
                          // the packed probabilities; other shapes take the generic kernel, which applies the same mask
        if constexpr (D == 64 && !CAUSAL) {
          dim3 grid((a->max_qlen + 127) / 128, a->n_heads, a->nseq);
          if (sl_env().attn_fwd_st == 2) hipLaunchKernelGGL((attn_fwd_tr_kernel<D, CAUSAL, 2, true, 2>), grid, dim3(256), 0, st, p);
          else hipLaunchKernelGGL((attn_fwd_tr_kernel<D, CAUSAL, 2, true>), grid, dim3(256), 0, st, p);
          SL_CHECK_LAUNCH("attn_fwd_tr");
          return 0;
        }
      } else {
        const int qt_env = sl_env().attn_qt;   // tuning switch
        if constexpr (D == 64) {
          // 64 queries per wave where the sequences are long enough to fill such blocks: K / V fragments read once per 4 query tiles
          if (qt_env ? qt_env == 4 : a->max_qlen > 192) {
            dim3 grid((a->max_qlen + 255) / 256, a->n_heads, a->nseq);
            if (sl_env().attn_fwd_st == 2) hipLaunchKernelGGL((attn_fwd_tr_kernel<D, CAUSAL, 4, false, 2>), grid, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((attn_fwd_tr_kernel<D, CAUSAL, 4>), grid, dim3(256), 0, st, p);
            SL_CHECK_LAUNCH("attn_fwd_tr");
            return 0;
          }
        }
        constexpr int QT = 2;
        dim3 grid((a->max_qlen + QT * 64 - 1) / (QT * 64), a->n_heads, a->nseq);
        hipLaunchKernelGGL((attn_fwd_tr_kernel<D, CAUSAL, QT>), grid, dim3(256), 0, st, p);
        SL_CHECK_LAUNCH("attn_fwd_tr");
        return 0;
      }
    }
  }
  dim3 grid((a->max_qlen + 63) / 64, a->n_heads, a->nseq);
  hipLaunchKernelGGL((attn_fwd_kernel<T, D, CAUSAL>), grid, dim3(256), 0, st, p);
  SL_CHECK_LAUNCH("attn_fwd");
  return 0;
}

extern "C" int sl_attn_fwd(const sl_attn_args* a, sl_stream stream) {
  SL_CHECK_ARG(a && a->q && a->k && a->v && a->out && a->cu_q && a->cu_k && a->klen, "sl_attn_fwd: null pointer");
  SL_CHECK_ARG(a->nseq > 0 && a->max_qlen > 0 && a->n_heads > 0 && a->n_kv_heads > 0 && a->n_heads % a->n_kv_heads == 0,
               "sl_attn_fwd: bad shape");
  SL_CHECK_ARG(a->dropout_p >= 0.f && a->dropout_p < 1.f, "sl_attn_fwd: dropout_p=%f outside [0, 1)", (double)a->dropout_p);
  const int vec = a->dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(a->q_row_stride % vec == 0 && a->k_row_stride % vec == 0 && a->v_row_stride % vec == 0 && a->q_head_stride % vec == 0 &&
                   a->k_head_stride % vec == 0 && a->v_head_stride % vec == 0,
               "sl_attn_fwd: strides must keep 16-byte alignment");
  hipStream_t st = (hipStream_t)stream;
  SL_DISPATCH_DTYPE(a->dtype, T, {
    if (a->head_dim == 64) return a->causal ? launch_attn<T, 64, true>(a, st) : launch_attn<T, 64, false>(a, st);
    if (a->head_dim == 128) return a->causal ? launch_attn<T, 128, true>(a, st) : launch_attn<T, 128, false>(a, st);
    sl_set_error("sl_attn_fwd: head_dim %d not built (64, 128)", a->head_dim);
    return SL_ERR_UNSUPPORTED;
  });
}
