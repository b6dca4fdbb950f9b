// attention.hip — flash-style attention forward on MFMA for packed variable-length sequences.
//   HuBERT: head_dim 64, bidirectional, no mask (hf:models/hubert/modeling_hubert.py:234-259)
//   Llama prefill: head_dim 128, causal, GQA (hf:models/llama/modeling_llama.py:191-213)
// Block = 4 waves x 16 query rows; 64-key K and V^T tiles in XOR-swizzled LDS; S = Q.K^T and O += P.V
// both run on the dtype-generic 16x16 MFMA step (bf16: 16x16x32, fp32: exact 16x16x4), softmax state
// (running max / sum) in fp32 registers, scores never leave the CU.
#include "common.h"

struct AttnP {
  const void* q; int64_t q_rs, q_hs;
  const void* k; int64_t k_rs, k_hs;
  const void* v; int64_t v_rs, v_hs;
  void* o; int64_t o_rs, o_hs;
  const int32_t* cu_q; const int32_t* cu_k; const int32_t* klen;
  int n_heads, n_kv;
  float scale;
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
        const float pv = __expf(s[n][i] - mx[i]);
        rs[i] += pv;
        const int row = qd * 4 + i;
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
  dim3 grid((a->max_qlen + 63) / 64, a->n_heads, a->nseq);
  hipLaunchKernelGGL((attn_fwd_kernel<T, D, CAUSAL>), grid, dim3(256), 0, st, p);
  SL_CHECK_LAUNCH("attn_fwd");
  return 0;
}

extern "C" int sl_attn_fwd(const sl_attn_args* a, sl_stream stream) {
  SL_CHECK_ARG(a && a->q && a->k && a->v && a->out && a->cu_q && a->cu_k && a->klen, "sl_attn_fwd: null pointer");
  SL_CHECK_ARG(a->nseq > 0 && a->max_qlen > 0 && a->n_heads > 0 && a->n_kv_heads > 0 && a->n_heads % a->n_kv_heads == 0,
               "sl_attn_fwd: bad shape");
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
