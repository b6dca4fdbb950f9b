// llama_ops.hip — the small Llama-side kernels: embedding gather, RoPE + KV-cache append, one-token GQA
// attention against the cache, greedy token selection.  All HBM/latency-bound; 16-byte accesses, fp32 math.
#include <stdlib.h>

#include "common.h"

// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_gather_kernel(const T* __restrict__ table, const int32_t* __restrict__ ids,
                                                           T* __restrict__ out, int64_t n, int cols) {
  constexpr int VEC = Vec16<T>::VEC;
  const int cpr = cols / VEC;
  const int64_t total = n * cpr;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / cpr;
    const int ch = (int)(i % cpr);
    *(uint4*)(out + row * cols + ch * VEC) = *(const uint4*)(table + (int64_t)ids[row] * cols + ch * VEC);
  }
}

extern "C" int sl_embed_gather(const void* table, const int32_t* ids, void* out, int64_t n, int32_t cols, int32_t dtype,
                               sl_stream stream) {
  SL_CHECK_ARG(table && ids && out && n >= 0 && cols > 0, "sl_embed_gather: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(cols % vec == 0, "sl_embed_gather: cols=%d must be a multiple of %d", cols, vec);
  if (n == 0) return 0;
  const int64_t total = n * (cols / vec);
  const unsigned grid = (unsigned)(ceil_div64(total, 256) < 4096 ? ceil_div64(total, 256) : 4096);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((embed_gather_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)table, ids, (T*)out, n, cols);
  });
  SL_CHECK_LAUNCH("embed_gather");
  return 0;
}

// ----------------------------------------------------------------------------------------------
// RoPE + KV append.  One thread owns chunk j of the first half of a head and the matching chunk of the
// second half (rotate_half pairs d with d + D/2).
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_append_kernel(T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
                                                             const int32_t* __restrict__ tok_seq, const int32_t* __restrict__ tok_pos,
                                                             const float* __restrict__ cosT, const float* __restrict__ sinT, int64_t n_tok,
                                                             int nh, int nkv, int D, int max_ctx) {
  constexpr int VEC = Vec16<T>::VEC;
  const int half = D / 2, cph = half / VEC;  // chunks per half head
  const int heads = nh + 2 * nkv;
  const int64_t total = n_tok * heads * cph;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int j = (int)(i % cph);
    const int h = (int)((i / cph) % heads);
    const int64_t t = i / ((int64_t)cph * heads);
    T* src = qkv + t * (int64_t)heads * D + (int64_t)h * D + j * VEC;
    const int pos = tok_pos[t];
    uint4 u1 = *(const uint4*)src, u2 = *(const uint4*)(src + half);
    if (h < nh + nkv) {
      float a[VEC], b[VEC], c[VEC], s[VEC], o1[VEC], o2[VEC];
      Vec16<T>::unpack(u1, a);
      Vec16<T>::unpack(u2, b);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        c[e] = cosT[(int64_t)pos * half + j * VEC + e];
        s[e] = sinT[(int64_t)pos * half + j * VEC + e];
        o1[e] = a[e] * c[e] - b[e] * s[e];  // x*cos + rotate_half(x)*sin, first half: -x2*sin
        o2[e] = b[e] * c[e] + a[e] * s[e];  // second half: +x1*sin
      }
      u1 = Vec16<T>::pack(o1);
      u2 = Vec16<T>::pack(o2);
    }
    if (h < nh) {
      *(uint4*)src = u1;
      *(uint4*)(src + half) = u2;
    } else {
      const bool isk = h < nh + nkv;
      const int kvh = isk ? h - nh : h - nh - nkv;
      T* dst = (isk ? kc : vc) + (((int64_t)tok_seq[t] * nkv + kvh) * max_ctx + pos) * D + j * VEC;
      *(uint4*)dst = u1;
      *(uint4*)(dst + half) = u2;
    }
  }
}

extern "C" int sl_rope_kv_append(void* qkv, void* k_cache, void* v_cache, const int32_t* tok_seq, const int32_t* tok_pos,
                                 const float* cos, const float* sin, int64_t n_tok, int32_t n_heads, int32_t n_kv, int32_t D,
                                 int32_t max_ctx, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(qkv && k_cache && v_cache && tok_seq && tok_pos && cos && sin, "sl_rope_kv_append: null pointer");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(D % (2 * vec) == 0, "sl_rope_kv_append: head_dim=%d must be a multiple of %d", D, 2 * vec);
  if (n_tok == 0) return 0;
  const int64_t total = n_tok * (n_heads + 2 * n_kv) * (D / 2 / vec);
  const unsigned grid = (unsigned)(ceil_div64(total, 256) < 8192 ? ceil_div64(total, 256) : 8192);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((rope_kv_append_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (T*)qkv, (T*)k_cache, (T*)v_cache,
                       tok_seq, tok_pos, cos, sin, n_tok, n_heads, n_kv, D, max_ctx);
  });
  SL_CHECK_LAUNCH("rope_kv_append");
  return 0;
}

// ----------------------------------------------------------------------------------------------
// One-token attention against the cache (decode).  Block = (kv head, sequence); the REP query heads
// that share the kv head are processed together so K and V are streamed once.
//   phase 1: scores  — a 16-lane group per key, each lane D/16 dims, xor-shuffle reduce
//   phase 2: softmax — one wave per query head over the LDS score row
//   phase 3: P.V     — 16 key groups x 16 dim chunks, LDS reduce over key groups
// ----------------------------------------------------------------------------------------------
template <typename T, int D, int REP>
__global__ __launch_bounds__(256) void attn_decode_kernel(const T* __restrict__ q, int64_t q_stride, const T* __restrict__ kc,
                                                          const T* __restrict__ vc, T* __restrict__ out, const int32_t* __restrict__ ctx_len,
                                                          int ctx_add, int nh, int nkv, int max_ctx, float scale) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int EPL = D / 16;       // elements per lane of a 16-lane group (8 for D=128)
  constexpr int CPLN = EPL / VEC;   // 16-byte chunks per lane: 1 (bf16) or 2 (f32)
  extern __shared__ __attribute__((aligned(16))) float dsm[];
  const int kvh = blockIdx.x, b = blockIdx.y;
  const int n_keys = ctx_len[b] + ctx_add;
  float* sc = dsm;                          // [REP][max_ctx]
  float* red = dsm + REP * max_ctx;         // [16][REP][D]
  float* inv_sum = red + 16 * REP * D;      // [REP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int grp = lane >> 4, gl = lane & 15;

  const T* kbase = kc + ((int64_t)b * nkv + kvh) * max_ctx * D;
  const T* vbase = vc + ((int64_t)b * nkv + kvh) * max_ctx * D;

  // phase 1
  float qr[REP][EPL];
#pragma unroll
  for (int h = 0; h < REP; ++h) {
    const T* qp = q + (int64_t)b * q_stride + (int64_t)(kvh * REP + h) * D + gl * EPL;
#pragma unroll
    for (int c = 0; c < CPLN; ++c) Vec16<T>::unpack(*(const uint4*)(qp + c * VEC), &qr[h][c * VEC]);
  }
  for (int key0 = 0; key0 < n_keys; key0 += 16) {
    const int key = key0 + wave * 4 + grp;
    const int kk = key < n_keys ? key : n_keys - 1;
    float kf[EPL];
#pragma unroll
    for (int c = 0; c < CPLN; ++c) Vec16<T>::unpack(*(const uint4*)(kbase + (int64_t)kk * D + gl * EPL + c * VEC), &kf[c * VEC]);
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      float d = 0.f;
#pragma unroll
      for (int e = 0; e < EPL; ++e) d = fmaf(qr[h][e], kf[e], d);
      d += __shfl_xor(d, 8, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 1, 64);
      if (gl == 0 && key < n_keys) sc[h * max_ctx + key] = d * scale;
    }
  }
  __syncthreads();
  // phase 2
  for (int h = wave; h < REP; h += 4) {
    float m = -INFINITY;
    for (int k = lane; k < n_keys; k += 64) m = fmaxf(m, sc[h * max_ctx + k]);
    m = wave_max(m);
    float s = 0.f;
    for (int k = lane; k < n_keys; k += 64) {
      const float p = __expf(sc[h * max_ctx + k] - m);
      sc[h * max_ctx + k] = p;
      s += p;
    }
    s = wave_sum(s);
    if (lane == 0) inv_sum[h] = 1.0f / s;
  }
  __syncthreads();
  // phase 3
  {
    const int kg = tid >> 4, dc = tid & 15;
    float acc[REP][EPL];
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc[h][e] = 0.f;
    for (int key = kg; key < n_keys; key += 16) {
      float vf[EPL];
#pragma unroll
      for (int c = 0; c < CPLN; ++c) Vec16<T>::unpack(*(const uint4*)(vbase + (int64_t)key * D + dc * EPL + c * VEC), &vf[c * VEC]);
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        // bf16 mode: probabilities are rounded to the storage dtype before P.V as HF eager does
        const float p = to_f32(from_f32<T>(sc[h * max_ctx + key] * inv_sum[h]));
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[h][e] = fmaf(p, vf[e], acc[h][e]);
      }
    }
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
      for (int e = 0; e < EPL; ++e) red[(kg * REP + h) * D + dc * EPL + e] = acc[h][e];
  }
  __syncthreads();
  for (int o = tid; o < REP * D; o += 256) {
    const int h = o / D, d = o % D;
    float s = 0.f;
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) s += red[(kg * REP + h) * D + d];
    out[(int64_t)b * nh * D + (int64_t)(kvh * REP + h) * D + d] = from_f32<T>(s);
  }
}

template <typename T, int REP>
static int launch_attn_decode(const void* q, int64_t q_stride, const void* kc, const void* vc, void* out, const int32_t* ctx_len,
                              int ctx_add, int B, int nh, int nkv, int max_ctx, float scale, hipStream_t st) {
  constexpr int D = 128;
  const size_t lds = ((size_t)REP * max_ctx + 16 * REP * D + REP) * sizeof(float);
  SL_CHECK_ARG(lds <= 160 * 1024, "sl_attn_decode: max_ctx=%d needs %zu B of LDS (> 160 KiB)", max_ctx, lds);
  auto kern = attn_decode_kernel<T, D, REP>;
  if (lds > 64 * 1024) SL_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(nkv, B), dim3(256), lds, st, (const T*)q, q_stride, (const T*)kc, (const T*)vc, (T*)out, ctx_len, ctx_add,
                     nh, nkv, max_ctx, scale);
  SL_CHECK_LAUNCH("attn_decode");
  return 0;
}

int sl_attn_decode_impl(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache, void* out, const int32_t* ctx_len,
                        int ctx_add, int32_t B, int32_t n_heads, int32_t n_kv, int32_t D, int32_t max_ctx, float scale, int32_t dtype,
                        hipStream_t st) {
  SL_CHECK_ARG(q && k_cache && v_cache && out && ctx_len && B > 0, "sl_attn_decode: bad arguments");
  SL_CHECK_ARG(D == 128, "sl_attn_decode: head_dim %d not built (Llama family uses 128)", D);
  SL_CHECK_ARG(n_kv > 0 && n_heads % n_kv == 0, "sl_attn_decode: n_heads %% n_kv != 0");
  const int rep = n_heads / n_kv;
  SL_DISPATCH_DTYPE(dtype, T, {
    switch (rep) {
      case 1: return launch_attn_decode<T, 1>(q, q_stride, k_cache, v_cache, out, ctx_len, ctx_add, B, n_heads, n_kv, max_ctx, scale, st);
      case 2: return launch_attn_decode<T, 2>(q, q_stride, k_cache, v_cache, out, ctx_len, ctx_add, B, n_heads, n_kv, max_ctx, scale, st);
      case 3: return launch_attn_decode<T, 3>(q, q_stride, k_cache, v_cache, out, ctx_len, ctx_add, B, n_heads, n_kv, max_ctx, scale, st);
      case 4: return launch_attn_decode<T, 4>(q, q_stride, k_cache, v_cache, out, ctx_len, ctx_add, B, n_heads, n_kv, max_ctx, scale, st);
      default: sl_set_error("sl_attn_decode: n_heads/n_kv=%d not built (1..4)", rep); return SL_ERR_UNSUPPORTED;
    }
  });
}

extern "C" int sl_attn_decode(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache, void* out,
                              const int32_t* ctx_len, int32_t B, int32_t n_heads, int32_t n_kv, int32_t D, int32_t max_ctx, float scale,
                              int32_t dtype, sl_stream stream) {
  return sl_attn_decode_impl(q, q_stride, k_cache, v_cache, out, ctx_len, 0, B, n_heads, n_kv, D, max_ctx, scale, dtype, (hipStream_t)stream);
}

// ----------------------------------------------------------------------------------------------
// greedy selection: one block per sequence row
// ----------------------------------------------------------------------------------------------
struct EosList { int ids[8]; int n; };

__global__ __launch_bounds__(1024) void greedy_select_kernel(const float* __restrict__ logits, int V, EosList eos, int pad_id, int use_eos,
                                                             int advance_ctx, int32_t* __restrict__ unfinished, int32_t* __restrict__ ctx_len,
                                                             int32_t* __restrict__ gen_count, int32_t* __restrict__ finish_len,
                                                             int32_t* __restrict__ next_ids, int32_t* __restrict__ out_ids, int max_new,
                                                             const int32_t* __restrict__ forced, const int32_t* __restrict__ row_limit) {
  __shared__ float smax[16];
  __shared__ int sidx[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)b * V;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  if (forced) {   // sampling mode: the token was drawn by sample_rows_kernel; only the bookkeeping below runs
    if (tid == 0) {
      int tok = forced[b];
      const int unf = unfinished[b];
      if (use_eos) tok = unf ? tok : pad_id;
      const int n = gen_count[b];
      if (n < max_new) out_ids[(int64_t)b * max_new + n] = tok;
      gen_count[b] = n + 1;
      next_ids[b] = tok;
      if (use_eos && unf) {
        bool is_eos = false;
        for (int e = 0; e < eos.n; ++e) is_eos |= (tok == eos.ids[e]);
        if (row_limit) is_eos |= (n + 1 >= row_limit[b]);       // the row's own token budget is spent: finished like a row that emitted EOS
        if (is_eos) { unfinished[b] = 0; finish_len[b] = n + 1; }
      }
      if (advance_ctx) ctx_len[b] += 1;
    }
    return;
  }
  auto upd = [&](float v, int i) {
    if (v > best || (v == best && i < bi)) { best = v; bi = i; }   // NaN never wins, like a strict '>' scan
  };
  if ((V & 3) == 0) {  // 16-byte loads, four in flight per thread: the scan is latency-bound otherwise
    const float4* row4 = (const float4*)row;
    const int n4 = V >> 2;
    for (int i0 = tid; i0 < n4; i0 += 4096) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 1024;
        v[u] = i < n4 ? row4[i] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (i0 + u * 1024 >= n4) continue;
        const int i = (i0 + u * 1024) * 4;
        upd(v[u].x, i); upd(v[u].y, i + 1); upd(v[u].z, i + 2); upd(v[u].w, i + 3);
      }
    }
  } else {
    for (int i = tid; i < V; i += 1024) upd(row[i], i);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(best, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) { smax[wave] = best; sidx[wave] = bi; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 16; ++w)
      if (smax[w] > best || (smax[w] == best && sidx[w] < bi)) { best = smax[w]; bi = sidx[w]; }
    if (bi == 0x7fffffff) bi = 0;
    int tok = bi;
    int unf = unfinished[b];
    if (use_eos) tok = unf ? tok : pad_id;   // hf:generation/utils.py:2928-2929
    const int n = gen_count[b];
    if (n < max_new) out_ids[(int64_t)b * max_new + n] = tok;
    gen_count[b] = n + 1;
    next_ids[b] = tok;
    if (use_eos && unf) {
      bool is_eos = false;
      for (int e = 0; e < eos.n; ++e) is_eos |= (tok == eos.ids[e]);
      if (row_limit) is_eos |= (n + 1 >= row_limit[b]);
      if (is_eos) { unfinished[b] = 0; finish_len[b] = n + 1; }
    }
    if (advance_ctx) ctx_len[b] += 1;
  }
}

int sl_greedy_select_impl(const float* logits, int32_t B, int32_t V, const int32_t* eos_ids, int32_t n_eos, int32_t pad_id,
                          int32_t use_eos, int32_t advance_ctx, int32_t* unfinished, int32_t* ctx_len, int32_t* gen_count,
                          int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new, hipStream_t st, const int32_t* row_limit) {
  SL_CHECK_ARG(logits && unfinished && ctx_len && gen_count && finish_len && next_ids && out_ids && B > 0 && V > 0,
               "sl_greedy_select: bad arguments");
  SL_CHECK_ARG(n_eos >= 0 && n_eos <= 8, "sl_greedy_select: at most 8 eos ids");
  EosList e;
  e.n = n_eos;
  for (int i = 0; i < 8; ++i) e.ids[i] = i < n_eos ? eos_ids[i] : -1;
  hipLaunchKernelGGL(greedy_select_kernel, dim3(B), dim3(1024), 0, st, logits, V, e, pad_id, use_eos, advance_ctx, unfinished, ctx_len,
                     gen_count, finish_len, next_ids, out_ids, max_new, (const int32_t*)nullptr, row_limit);
  SL_CHECK_LAUNCH("greedy_select");
  return 0;
}

// greedy selection over the [group][B] partial maxima a fused lm_head left (gemm.hip tile_argmax): block = 32 rows x 32 group
// stripes (a wave = two stripes of 32 consecutive rows: two 128-byte segments per load), eight groups' values and columns in
// flight per thread — the scan is latency-bound (2 004 groups of 8 bytes per row at vocab 128 256), not bandwidth-bound; the
// stripes meet in LDS in stripe order and the bookkeeping of greedy_select_kernel follows.  Compare rule everywhere: larger
// value, then lower column.
__global__ __launch_bounds__(1024) void greedy_select_partial_kernel(const float* __restrict__ pv, const int32_t* __restrict__ pi, int n_groups, int B,
                                                                     EosList eos, int pad_id, int use_eos, int advance_ctx,
                                                                     int32_t* __restrict__ unfinished, int32_t* __restrict__ ctx_len,
                                                                     int32_t* __restrict__ gen_count, int32_t* __restrict__ finish_len,
                                                                     int32_t* __restrict__ next_ids, int32_t* __restrict__ out_ids, int max_new,
                                                                     const int32_t* __restrict__ row_limit) {
  constexpr int ROWS = 32, STRIPES = 32, U = 8;
  __shared__ float sv[STRIPES][ROWS];
  __shared__ int si[STRIPES][ROWS];
  const int lr = threadIdx.x & (ROWS - 1), stripe = threadIdx.x / ROWS;
  const int b = blockIdx.x * ROWS + lr;
  const int bc = b < B ? b : B - 1;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int g0 = stripe; g0 < n_groups; g0 += STRIPES * U) {
    float v[U];
    int ix[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int g = g0 + u * STRIPES;
      const int64_t at = (int64_t)(g < n_groups ? g : n_groups - 1) * B + bc;
      v[u] = pv[at];
      ix[u] = pi[at];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (g0 + u * STRIPES >= n_groups) continue;
      if (v[u] > best || (v[u] == best && ix[u] < bi)) { best = v[u]; bi = ix[u]; }
    }
  }
  sv[stripe][lr] = best;
  si[stripe][lr] = bi;
  __syncthreads();
  if (stripe != 0 || b >= B) return;
  for (int s2 = 1; s2 < STRIPES; ++s2)
    if (sv[s2][lr] > best || (sv[s2][lr] == best && si[s2][lr] < bi)) { best = sv[s2][lr]; bi = si[s2][lr]; }
  if (bi == 0x7fffffff) bi = 0;
  int tok = bi;
  const int unf = unfinished[b];
  if (use_eos) tok = unf ? tok : pad_id;   // hf:generation/utils.py:2928-2929
  const int n = gen_count[b];
  if (n < max_new) out_ids[(int64_t)b * max_new + n] = tok;
  gen_count[b] = n + 1;
  next_ids[b] = tok;
  if (use_eos && unf) {
    bool is_eos = false;
    for (int e = 0; e < eos.n; ++e) is_eos |= (tok == eos.ids[e]);
    if (row_limit) is_eos |= (n + 1 >= row_limit[b]);
    if (is_eos) { unfinished[b] = 0; finish_len[b] = n + 1; }
  }
  if (advance_ctx) ctx_len[b] += 1;
}

int sl_greedy_select_partial_impl(const float* amax_val, const int32_t* amax_idx, int32_t n_groups, int32_t B, const int32_t* eos_ids, int32_t n_eos,
                                  int32_t pad_id, int32_t use_eos, int32_t advance_ctx, int32_t* unfinished, int32_t* ctx_len, int32_t* gen_count,
                                  int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new, hipStream_t st, const int32_t* row_limit) {
  SL_CHECK_ARG(amax_val && amax_idx && unfinished && ctx_len && gen_count && finish_len && next_ids && out_ids && B > 0 && n_groups > 0,
               "sl_greedy_select_partial: bad arguments");
  SL_CHECK_ARG(n_eos >= 0 && n_eos <= 8, "sl_greedy_select_partial: at most 8 eos ids");
  EosList e;
  e.n = n_eos;
  for (int i = 0; i < 8; ++i) e.ids[i] = i < n_eos ? eos_ids[i] : -1;
  hipLaunchKernelGGL(greedy_select_partial_kernel, dim3((B + 31) / 32), dim3(1024), 0, st, amax_val, amax_idx, n_groups, B, e, pad_id, use_eos, advance_ctx,
                     unfinished, ctx_len, gen_count, finish_len, next_ids, out_ids, max_new, row_limit);
  SL_CHECK_LAUNCH("greedy_select_partial");
  return 0;
}

extern "C" int sl_greedy_select_partial(const float* amax_val, const int32_t* amax_idx, int32_t n_groups, int32_t B, const int32_t* eos_ids, int32_t n_eos,
                                        int32_t pad_id, int32_t use_eos, int32_t advance_ctx, int32_t* unfinished, int32_t* ctx_len, int32_t* gen_count,
                                        int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new, sl_stream stream) {
  return sl_greedy_select_partial_impl(amax_val, amax_idx, n_groups, B, eos_ids, n_eos, pad_id, use_eos, advance_ctx, unfinished, ctx_len, gen_count,
                                       finish_len, next_ids, out_ids, max_new, (hipStream_t)stream, nullptr);
}

// ----------------------------------------------------------------------------------------------
// sampled selection (hf:generation/utils.py:2911-2923 with do_sample: logits processors then multinomial):
//   TemperatureLogitsWarper (scores / T), TopKLogitsWarper (keep scores >= the k-th largest), TopPLogitsWarper (on the
//   top-k-filtered distribution: drop token i iff the probability mass of the tokens ranked ABOVE it is >= top_p; at least
//   one token survives), then one draw from the renormalised survivors.  One block per row; thresholds by 4-pass radix
//   selection over the order-preserving 32-bit image of the logits (by count for top-k, by probability mass for top-p), so no
//   sort and no O(V) scratch; the draw is the inverse CDF in INDEX order at u = hash(seed, row, step) — reproducible, and the
//   test restates it on the host.  Tokens of equal logit share their fate (all kept or all dropped).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ord_key(float x) {
  const uint32_t u = __float_as_uint(x);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(1024) void sample_rows_kernel(const float* __restrict__ logits, int V, float inv_temp, int top_k, float top_p, uint64_t seed,
                                                           const int32_t* __restrict__ gen_count, int32_t* __restrict__ choice,
                                                           const int32_t* __restrict__ row_ids) {
  __shared__ float redf[16];
  __shared__ int hcnt[256];
  __shared__ unsigned long long hmass_fx[256];   // probability mass per bin in 2^-40 fixed point: integer adds commute, so the
                                                 // bin totals (and the threshold compare below) are the same on every run
  __shared__ uint32_t s_prefix;
  __shared__ float s_above;
  __shared__ float part[1024];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* row = logits + (int64_t)b * V;
  auto block_sum = [&](float v) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) redf[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < 16; ++w) t += redf[w];
    return t;
  };
  // row maximum (numerical stability of the exponentials)
  float m = -INFINITY;
  for (int i = tid; i < V; i += 1024) m = fmaxf(m, row[i]);
  m = wave_max(m);
  if (lane == 0) redf[wave] = m;
  __syncthreads();
  m = redf[0];
  for (int w = 1; w < 16; ++w) m = fmaxf(m, redf[w]);
  __syncthreads();
  auto prob = [&](float x) { return __expf((x - m) * inv_temp); };

  // radix selection: MODE 0 = by count (k-th largest key), MODE 1 = by mass (first key, from the top, at which the mass of
  // the strictly larger keys is still < limit).  `floor_key`: only keys >= floor_key take part.  Returns the selected key.
  auto radix_select = [&](int mode, float limit, uint32_t floor_key) -> uint32_t {
    uint32_t prefix = 0;
    float above = 0.f;     // count / mass of keys larger than every key matching the current prefix
    for (int pass = 0; pass < 4; ++pass) {
      const int shift = 24 - 8 * pass;
      const uint32_t hi_mask = pass == 0 ? 0u : (0xFFFFFFFFu << (shift + 8));
      for (int i = tid; i < 256; i += 1024) { hcnt[i] = 0; hmass_fx[i] = 0ull; }
      __syncthreads();
      for (int i = tid; i < V; i += 1024) {
        const float x = row[i];
        const uint32_t k = ord_key(x);
        if (k < floor_key || (k & hi_mask) != prefix) continue;
        const int bin = (k >> shift) & 255;
        if (mode == 0) atomicAdd(&hcnt[bin], 1);
        else atomicAdd(&hmass_fx[bin], (unsigned long long)(prob(x) * 1099511627776.0f));   // prob <= 1: < 2^57 over a 128 k row
      }
      __syncthreads();
      if (tid == 0) {
        float run = above;
        int sel = -1;
        for (int bin = 255; bin >= 0; --bin) {
          const float w = mode == 0 ? (float)hcnt[bin] : (float)hmass_fx[bin] * (1.0f / 1099511627776.0f);
          if (mode == 0 ? (hcnt[bin] > 0 && run + w >= limit) : (hmass_fx[bin] > 0ull && run + w >= limit)) { sel = bin; break; }
          run += w;
        }
        if (sel < 0) {           // the limit is never reached (rounding / k >= candidates): take the smallest populated bin
          run = above;
          for (int bin = 255; bin >= 0; --bin) {
            const bool pop = mode == 0 ? hcnt[bin] > 0 : hmass_fx[bin] > 0ull;
            if (pop) { sel = bin; }
          }
          run = above;
          for (int bin = 255; bin > sel; --bin) run += mode == 0 ? (float)hcnt[bin] : (float)hmass_fx[bin] * (1.0f / 1099511627776.0f);
          if (sel < 0) sel = 0;
        }
        s_prefix = prefix | ((uint32_t)sel << shift);
        s_above = run;
      }
      __syncthreads();
      prefix = s_prefix;
      above = s_above;
      __syncthreads();
    }
    return prefix;
  };

  uint32_t keep_key = 0;                               // survivors: ord_key(logit) >= keep_key
  if (top_k > 0 && top_k < V) keep_key = radix_select(0, (float)top_k, 0u);
  if (top_p < 1.0f) {
    float z1 = 0.f;
    for (int i = tid; i < V; i += 1024) { const float x = row[i]; if (ord_key(x) >= keep_key) z1 += prob(x); }
    z1 = block_sum(z1);
    const uint32_t kp = radix_select(1, top_p * z1, keep_key);
    keep_key = kp > keep_key ? kp : keep_key;
  }
  // inverse CDF in index order: thread t owns the contiguous slice [t * C, (t + 1) * C)
  const int Cn = (V + 1023) / 1024;
  const int i0 = tid * Cn, i1 = (i0 + Cn) < V ? (i0 + Cn) : V;
  float loc = 0.f;
  for (int i = i0; i < i1; ++i) { const float x = row[i]; if (ord_key(x) >= keep_key) loc += prob(x); }
  part[tid] = loc;
  __syncthreads();
  if (tid == 0) {
    float z = 0.f;
    for (int t = 0; t < 1024; ++t) z += part[t];
    const uint32_t step = (uint32_t)gen_count[b];
    const uint32_t rid = row_ids ? (uint32_t)row_ids[b] : (uint32_t)b;    // the sequence's index in the caller's batch (rows move when a batch is compacted)
    const uint32_t h = lowbias32(step ^ lowbias32(rid ^ (uint32_t)seed) ^ (uint32_t)(seed >> 32));
    const float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    const float target = u * z;
    float run = 0.f;
    int t = 0;
    for (; t < 1023; ++t) { if (run + part[t] > target) break; run += part[t]; }
    // walk the slice; the LAST surviving index seen is the fallback when rounding leaves target >= the total
    int pick = -1, last = -1;
    const int j0 = t * Cn, j1 = (j0 + Cn) < V ? (j0 + Cn) : V;
    for (int i = j0; i < j1; ++i) {
      const float x = row[i];
      if (ord_key(x) < keep_key) continue;
      last = i;
      run += prob(x);
      if (run > target) { pick = i; break; }
    }
    if (pick < 0) pick = last;
    if (pick < 0) {   // empty slice (cannot happen with z > 0): fall back to the global argmax bookkeeping index 0
      for (int i = 0; i < V; ++i) if (ord_key(row[i]) >= keep_key) { pick = i; break; }
    }
    choice[b] = pick < 0 ? 0 : pick;
  }
}

int sl_sample_select_impl(const float* logits, int32_t B, int32_t V, float temperature, int32_t top_k, float top_p, uint64_t seed,
                          const int32_t* eos_ids, int32_t n_eos, int32_t pad_id, int32_t use_eos, int32_t advance_ctx, int32_t* unfinished,
                          int32_t* ctx_len, int32_t* gen_count, int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new,
                          int32_t* choice_ws, hipStream_t st, const int32_t* row_limit, const int32_t* row_ids) {
  SL_CHECK_ARG(logits && unfinished && ctx_len && gen_count && finish_len && next_ids && out_ids && choice_ws && B > 0 && V > 0,
               "sl_sample_select: bad arguments");
  SL_CHECK_ARG(temperature > 0.f && top_p > 0.f && top_p <= 1.0f && top_k >= 0, "sl_sample_select: need temperature > 0, 0 < top_p <= 1, top_k >= 0 (got %f, %f, %d)",
               (double)temperature, (double)top_p, top_k);
  SL_CHECK_ARG(n_eos >= 0 && n_eos <= 8, "sl_sample_select: at most 8 eos ids");
  EosList e;
  e.n = n_eos;
  for (int i = 0; i < 8; ++i) e.ids[i] = i < n_eos ? eos_ids[i] : -1;
  hipLaunchKernelGGL(sample_rows_kernel, dim3(B), dim3(1024), 0, st, logits, V, 1.0f / temperature, top_k, top_p, seed, gen_count, choice_ws, row_ids);
  SL_CHECK_LAUNCH("sample_rows");
  hipLaunchKernelGGL(greedy_select_kernel, dim3(B), dim3(64), 0, st, logits, V, e, pad_id, use_eos, advance_ctx, unfinished, ctx_len, gen_count, finish_len,
                     next_ids, out_ids, max_new, (const int32_t*)choice_ws, row_limit);
  SL_CHECK_LAUNCH("sample_commit");
  return 0;
}

extern "C" int sl_sample_select(const float* logits, int32_t B, int32_t V, float temperature, int32_t top_k, float top_p, uint64_t seed,
                                const int32_t* eos_ids, int32_t n_eos, int32_t pad_id, int32_t use_eos, int32_t* unfinished, int32_t* ctx_len,
                                int32_t* gen_count, int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new, int32_t* choice_ws,
                                sl_stream stream) {
  return sl_sample_select_impl(logits, B, V, temperature, top_k, top_p, seed, eos_ids, n_eos, pad_id, use_eos, 1, unfinished, ctx_len, gen_count, finish_len,
                               next_ids, out_ids, max_new, choice_ws, (hipStream_t)stream, nullptr, nullptr);
}

extern "C" int sl_greedy_select(const float* logits, int32_t B, int32_t V, const int32_t* eos_ids, int32_t n_eos, int32_t pad_id,
                                int32_t use_eos, int32_t* unfinished, int32_t* ctx_len, int32_t* gen_count, int32_t* finish_len,
                                int32_t* next_ids, int32_t* out_ids, int32_t max_new, sl_stream stream) {
  return sl_greedy_select_impl(logits, B, V, eos_ids, n_eos, pad_id, use_eos, 1, unfinished, ctx_len, gen_count, finish_len, next_ids,
                               out_ids, max_new, (hipStream_t)stream, nullptr);
}

// ----------------------------------------------------------------------------------------------
// fragment-major weight packing (see sl_pack_weight in speechllm.h)
// ----------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const T* __restrict__ src, int64_t ld, T* __restrict__ dst, int N, int K) {
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int KSTEP = 4 * VEC;
  const int nks = K / KSTEP;
  const int64_t total = (int64_t)((N + 15) / 16) * nks * 64;  // one 16-byte chunk per (fragment, k-step, lane)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int lane = (int)(i & 63);
    const int64_t fs = i >> 6;
    const int s = (int)(fs % nks);
    const int64_t f = fs / nks;
    const int64_t row = f * 16 + (lane & 15);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (row < N) v = *(const uint4*)(src + row * ld + (int64_t)s * KSTEP + (lane >> 4) * VEC);
    *(uint4*)(dst + i * VEC) = v;
  }
}

extern "C" int sl_pack_weight(const void* src, int64_t ld_src, void* dst, int32_t N, int32_t K, int32_t dtype, sl_stream stream) {
  SL_CHECK_ARG(src && dst && N > 0 && K > 0, "sl_pack_weight: bad arguments");
  const int vec = dtype == SL_F32 ? 4 : 8;
  SL_CHECK_ARG(K % (4 * vec) == 0 && ld_src % vec == 0, "sl_pack_weight: K=%d must be a multiple of %d", K, 4 * vec);
  const int64_t total = (int64_t)((N + 15) / 16) * (K / (4 * vec)) * 64;
  const unsigned grid = (unsigned)(ceil_div64(total, 256) < 16384 ? ceil_div64(total, 256) : 16384);
  SL_DISPATCH_DTYPE(dtype, T, {
    hipLaunchKernelGGL((pack_weight_kernel<T>), dim3(grid), dim3(256), 0, (hipStream_t)stream, (const T*)src, ld_src, (T*)dst, N, K);
  });
  SL_CHECK_LAUNCH("pack_weight");
  return 0;
}

// ----------------------------------------------------------------------------------------------
// Flash-decoding: one-token attention split over the context so that small batches still fill the chip.
//   grid (kv head, sequence, split); each block owns 64 keys: scores -> local softmax -> partial P.V,
//   and leaves (O[REP][128], m[REP], l[REP]) in fp32; a second tiny kernel merges the splits.
//   (Tried for batches that fill the chip on their own, B * n_kv >= 1024: one block per (sequence, kv head) walking the
//   context in 64-key chunks with an online softmax and the next chunk prefetched — no partial records, no merge launch —
//   was 20 % slower than split + merge at B = 128 and 256: the serial chunk chain exposes three barriers per 32 KiB.)
// ----------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr3_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;
// K / V cache rows are read exactly once per decode step (one block per sequence and kv head): non-temporal loads keep them from
// displacing what other kernels re-read in L2 / the Infinity Cache and land sooner (MI355X_MICROARCH.md, nt-weights).  Measured
// (tools/time_decode_attn.py, bf16, 8 kv heads, profiles/r03_l_attn_nt.txt): 1 024 sequences x 264 keys 194.1 -> 178.3 us
// (5.77 -> 6.28 TB/s = 0.785 of the 8 TB/s peak, the rate a plain copy reaches on this chip), x 393 keys 286.5 -> 260.8 us, 512
// sequences 102.3 -> 92.1 us, 64 sequences 17.3 -> 16.1 us.
#ifndef SL_KV_NT
#define SL_KV_NT 1
#endif
#if SL_KV_NT
#define SL_KV_LOAD(ptr) __builtin_nontemporal_load((const u32x4_t*)(ptr))
#else
#define SL_KV_LOAD(ptr) (*(const u32x4_t*)(ptr))
#endif
// floats per partial record (REP x 128 outputs, REP maxima, REP sums), rounded up to whole 128-byte lines so that the records of
// different (sequence, kv head, split) blocks never share a line (the in-launch merge hands them between workgroups with sc1 stores)
constexpr int split_record_floats(int rep) { return (rep * 128 + 2 * rep + 31) / 32 * 32; }
constexpr int DSPLIT = 64;  // keys per block (KS = 128 for batches that fill the chip anyway: half the records to merge)

// Partial records handed from the split blocks to the block that merges them INSIDE the launch (cnt != nullptr): every record
// word is stored with sc1 (agent scope: through the XCD's non-coherent L2 to the memory side), drained with s_waitcnt before the
// arrival counter is bumped, and re-read with sc1 loads by the block whose add came last (MI355X_MICROARCH.md, Correctness
// boundaries; the same hand-off as gemm_stream.hip's K-split fix-up).  The merge is attn_decode_combine_kernel's arithmetic in
// its order (splits ascending), so both forms give the same bits.
__device__ __forceinline__ void st_rec(float* p, float v, bool sc1) {
  if (sc1) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else *p = v;
}
// agent-scope relaxed load = global_load_dword ... sc1: re-reads a word another workgroup of this launch stored with sc1; the compiler
// keeps several in flight (an asm load + wait per word serialised ~28 memory-side round trips per merge)
__device__ __forceinline__ float ld_rec_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the last of a (sequence, kv head)'s `nsplit` blocks to arrive merges their records and writes the REP heads' outputs
template <typename T, int REP>
__device__ __forceinline__ void split_arrive_and_merge(float* __restrict__ part, int32_t* __restrict__ cnt, T* __restrict__ out, int b, int kvh, int nkv,
                                                        int nsplit) {
  constexpr int D = 128, PSTRIDE = split_record_floats(REP), U = 8;
  __shared__ int s_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's record stores have reached the memory side
  __syncthreads();
  if (threadIdx.x == 0) {
    const int old = atomicAdd(&cnt[b * nkv + kvh], 1);
    s_last = old == nsplit - 1;
    if (old == nsplit - 1) cnt[b * nkv + kvh] = 0;     // leave the counter as it was found: zero between launches
  }
  __syncthreads();
  if (!s_last) return;
  const float* base = part + ((int64_t)b * nkv + kvh) * nsplit * PSTRIDE;
  for (int o = threadIdx.x; o < REP * D; o += 256) {
    const int h = o / D, d = o % D;
    float M = -INFINITY;
    for (int s0 = 0; s0 < nsplit; s0 += U) {
      float mv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int sp2 = s0 + u < nsplit ? s0 + u : nsplit - 1;
        mv[u] = ld_rec_sc1(base + (int64_t)sp2 * PSTRIDE + REP * D + h);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) M = fmaxf(M, mv[u]);
    }
    float Lsum = 0.f, acc = 0.f;
    for (int s0 = 0; s0 < nsplit; s0 += U) {
      float mv[U], lv[U], ov[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int sp2 = s0 + u < nsplit ? s0 + u : nsplit - 1;
        const float* rec = base + (int64_t)sp2 * PSTRIDE;
        mv[u] = ld_rec_sc1(rec + REP * D + h);
        lv[u] = ld_rec_sc1(rec + REP * D + REP + h);
        ov[u] = ld_rec_sc1(rec + h * D + d);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (s0 + u >= nsplit || mv[u] == -INFINITY) continue;
        const float w = __expf(mv[u] - M);
        Lsum += w * lv[u];
        acc += w * ov[u];
      }
    }
    out[((int64_t)b * nkv * REP + kvh * REP + h) * D + d] = from_f32<T>(Lsum > 0.f ? acc / Lsum : 0.f);
  }
}

template <typename T, int REP, int KS>
__global__ __launch_bounds__(256) void attn_decode_split_kernel(const T* __restrict__ q, int64_t q_stride, const T* __restrict__ kc,
                                                                const T* __restrict__ vc, float* __restrict__ part,
                                                                const int32_t* __restrict__ ctx_len, int ctx_add, int nkv, int max_ctx,
                                                                float scale, int32_t* __restrict__ cnt, T* __restrict__ out) {
  constexpr int D = 128;
  constexpr int VEC = Vec16<T>::VEC;
  constexpr int EPL = D / 16, CPLN = EPL / VEC;
  constexpr int PSTRIDE = split_record_floats(REP);  // floats per partial record
  constexpr int NPS = KS / 16;   // passes of 16 keys
  constexpr bool MFMA_QK = sizeof(T) == 2;   // bf16: scores on the matrix core (the VALU form was ~75 % VALU-busy at B = 256)
  constexpr int KT_BYTES = MFMA_QK ? KS * D * (int)sizeof(T) : 16, RED_BYTES = 16 * REP * D * (int)sizeof(float);
  __shared__ float sc[REP][KS];
  // bf16: K tile of the score MFMAs ([KS rows][256 B], chunk c of row r at c ^ (r & 15)), then — same bytes — the V tile of
  // the P.V MFMAs ([KS rows][256 B], chunk c of row r at voff()); fp32: the cross-thread reduction buffer of the VALU form
  __shared__ __attribute__((aligned(16))) unsigned char un[KT_BYTES > RED_BYTES ? KT_BYTES : RED_BYTES];
  constexpr int PROW = KS * 2 + 16;   // bytes per row of the bf16 probability tile (padded: rows land on different banks)
  __shared__ __attribute__((aligned(16))) unsigned char pt[MFMA_QK ? 16 * PROW : 16];
  float (*red)[REP][D] = (float (*)[REP][D])un;
  const int kvh = blockIdx.x, b = blockIdx.y, sp = blockIdx.z, nsplit = gridDim.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, gl = lane & 15;
  const int n_keys = ctx_len[b] + ctx_add;
  const int k0 = sp * KS;
  float* rec = part + (((int64_t)b * nkv + kvh) * nsplit + sp) * PSTRIDE;
  const bool fuse = cnt != nullptr;      // merge inside this launch (no combine kernel)
  if (k0 >= n_keys) {  // empty split: neutral record
    if (tid < REP) { st_rec(rec + REP * D + tid, -INFINITY, fuse); st_rec(rec + REP * D + REP + tid, 0.f, fuse); }
    if (fuse) split_arrive_and_merge<T, REP>(part, cnt, out, b, kvh, nkv, nsplit);
    return;
  }
  const int nk = (n_keys - k0) < KS ? (n_keys - k0) : KS;
  const T* kbase = kc + (((int64_t)b * nkv + kvh) * max_ctx + k0) * D;
  const T* vbase = vc + (((int64_t)b * nkv + kvh) * max_ctx + k0) * D;

  // phase 1: scores; every K and V row of the split is requested up front (16-byte chunks, 256-byte rows coalesced)
  u32x4_t kraw[NPS][CPLN];   // ext-vector type: HIP's uint4 struct copies to LDS went through scratch
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    int key = ps * 16 + wave * 4 + grp;
    key = key < nk ? key : nk - 1;
#pragma unroll
    for (int c = 0; c < CPLN; ++c) kraw[ps][c] = SL_KV_LOAD(kbase + (int64_t)key * D + gl * EPL + c * VEC);
  }
  // V rows for phase 3 are requested now: their HBM latency hides behind the score / softmax phases
  const int kg = tid >> 4, dc = tid & 15;
  u32x4_t vraw[NPS][CPLN];
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    int key = kg + 16 * ps;
    key = key < nk ? key : nk - 1;
#pragma unroll
    for (int c = 0; c < CPLN; ++c) vraw[ps][c] = SL_KV_LOAD(vbase + (int64_t)key * D + dc * EPL + c * VEC);
  }
  if constexpr (MFMA_QK) {
    // S[head][key] = Q . K^T on the matrix core: A = the REP query heads of this kv head (rows REP..15 zero), B = 16 keys
    // from the LDS tile; wave w scores keys [w KS/4, (w+1) KS/4).  Lanes 0..15 end up holding heads 0..3 of their key.
    const int r = lane & 15, q4 = lane >> 4;
    uint4 qf[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
      qf[s4] = r < REP ? *(const uint4*)(q + (int64_t)b * q_stride + (int64_t)(kvh * REP + r) * D + 32 * s4 + 8 * q4) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int kl = ps * 16 + wave * 4 + grp;
      *(u32x4_t*)(un + kl * 256 + ((gl ^ (kl & 15)) << 4)) = kraw[ps][0];
    }
    __syncthreads();
    constexpr int NF = KS / 64;
#pragma unroll
    for (int n = 0; n < NF; ++n) {
      const int rowbase = wave * (KS / 4) + n * 16;
      f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) {
        const uint4 kf = *(const uint4*)(un + (rowbase + r) * 256 + (((4 * s4 + q4) ^ r) << 4));
        MMA<T>::step(sacc, qf[s4], kf);
      }
      if (q4 == 0) {
        const int key = rowbase + r;
#pragma unroll
        for (int h = 0; h < REP; ++h) sc[h][key] = key < nk ? sacc[h] * scale : -INFINITY;
      }
    }
  } else {
    float qr[REP][EPL];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
      const T* qp = q + (int64_t)b * q_stride + (int64_t)(kvh * REP + h) * D + gl * EPL;
#pragma unroll
      for (int c = 0; c < CPLN; ++c) Vec16<T>::unpack(*(const uint4*)(qp + c * VEC), &qr[h][c * VEC]);
    }
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int key = ps * 16 + wave * 4 + grp;
      float kf[EPL];
#pragma unroll
      for (int c = 0; c < CPLN; ++c) Vec16<T>::unpack(make_uint4(kraw[ps][c].x, kraw[ps][c].y, kraw[ps][c].z, kraw[ps][c].w), &kf[c * VEC]);
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) d = fmaf(qr[h][e], kf[e], d);
        d += __shfl_xor(d, 8, 64); d += __shfl_xor(d, 4, 64); d += __shfl_xor(d, 2, 64); d += __shfl_xor(d, 1, 64);
        if (gl == 0) sc[h][key] = key < nk ? d * scale : -INFINITY;
      }
    }
  }
  __syncthreads();
  // phase 2: local softmax, one wave per head, one key per lane
  for (int h = wave; h < REP; h += 4) {
    float sv[KS / 64], m = -INFINITY;
#pragma unroll
    for (int j = 0; j < KS / 64; ++j) { sv[j] = sc[h][lane + 64 * j]; m = fmaxf(m, sv[j]); }
    m = wave_max(m);
    float l = 0.f;
#pragma unroll
    for (int j = 0; j < KS / 64; ++j) {
      const float p = __expf(sv[j] - m);  // masked keys: exp(-inf) = 0
      if constexpr (MFMA_QK) *(T*)(pt + h * PROW + (lane + 64 * j) * 2) = from_f32<T>(p);   // P rounded to bf16, as HF eager does
      else sc[h][lane + 64 * j] = p;
      l += p;
    }
    l = wave_sum(l);
    if (lane == 0) { st_rec(rec + REP * D + h, m, fuse); st_rec(rec + REP * D + REP + h, l, fuse); }
  }
  if constexpr (MFMA_QK) {
    // phase 3 on the matrix core: O[head][dim] = P . V with V consumed column-wise by ds_read_b64_tr_b16 (a 16-lane group
    // reads a 4-key x 16-dim block and receives it transposed: lane i gets the 4 keys of dim i).  V tile image: chunk c of
    // row r at c ^ (((r & 3) << 2) | ((r >> 2) & 3)) — conflict-free for these reads (guide T10, image (b)).
    auto voff = [](int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); };
    for (int o = tid; o < (16 - REP) * (KS / 8); o += 256)   // zero the padding rows of P (heads REP..15)
      *(uint4*)(pt + (REP + o / (KS / 8)) * PROW + (o % (KS / 8)) * 16) = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) *(u32x4_t*)(un + voff(kg + 16 * ps, dc)) = vraw[ps][0];   // K tile is dead: scores are in sc
    __syncthreads();
    const int r = lane & 15, q4 = lane >> 4, qq = r >> 2, pp = r & 3;
    const uint32_t ub = (uint32_t)(uintptr_t)(lds_ptr3_t)un;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int df = 2 * wave + j;   // 16-dim fragment of the head dimension
      f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS / 32; ++ks) {
        const uint4 pa = *(const uint4*)(pt + r * PROW + (32 * ks + 8 * q4) * 2);
        const int r0 = 32 * ks + 8 * q4 + qq;
        u32x2_t lo, hi;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(ub + (uint32_t)(voff(r0, 2 * df + (pp >> 1)) + 8 * (pp & 1))) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(ub + (uint32_t)(voff(r0 + 4, 2 * df + (pp >> 1)) + 8 * (pp & 1))) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi));
        MMA<T>::step(oacc, pa, make_uint4(lo.x, lo.y, hi.x, hi.y));
      }
      if (q4 == 0) {
#pragma unroll
        for (int h = 0; h < REP; ++h) st_rec(rec + h * D + df * 16 + r, oacc[h], fuse);
      }
    }
    if (fuse) split_arrive_and_merge<T, REP>(part, cnt, out, b, kvh, nkv, nsplit);
    return;
  }
  __syncthreads();
  // phase 3: partial P.V
  {
    float acc[REP][EPL];
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
      for (int e = 0; e < EPL; ++e) acc[h][e] = 0.f;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int key = kg + 16 * ps;
      float vf[EPL];
#pragma unroll
      for (int c = 0; c < CPLN; ++c) Vec16<T>::unpack(make_uint4(vraw[ps][c].x, vraw[ps][c].y, vraw[ps][c].z, vraw[ps][c].w), &vf[c * VEC]);
#pragma unroll
      for (int h = 0; h < REP; ++h) {
        const float p = sc[h][key];  // 0 for keys past nk
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[h][e] = fmaf(p, vf[e], acc[h][e]);
      }
    }
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
      for (int e = 0; e < EPL; ++e) red[kg][h][dc * EPL + e] = acc[h][e];
  }
  __syncthreads();
  for (int o = tid; o < REP * D; o += 256) {
    const int h = o / D, d = o % D;
    float s = 0.f;
#pragma unroll
    for (int kg = 0; kg < 16; ++kg) s += red[kg][h][d];
    st_rec(rec + h * D + d, s, fuse);
  }
  if (fuse) split_arrive_and_merge<T, REP>(part, cnt, out, b, kvh, nkv, nsplit);
}

// Single-pass form (bf16, B * n_kv >= 32): one block per (sequence, kv head)
// walks the context in 128-key chunks with an online softmax, all products on the matrix core as in the split kernel.  No
// partial records, no merge launch.  PREFETCH = false (used): a chunk's K rows are requested at its start and its V rows as
// soon as the K registers are free (138 VGPRs, 3 blocks per CU, other blocks cover the latency): 97 us per layer at B = 512;
// PREFETCH = true holds the next chunk's K and V rows in a second register set (239 VGPRs, 2 blocks per CU): 104 us.
// Four blocks per CU (query fragments and a shared zero row in LDS: 128 VGPRs + 16 B of scratch, 36 KiB) would make 4096 blocks exactly
// four rounds instead of 5.33 on 768 slots, but measured 117 us: the extra LDS reads per MFMA and the fourth block's traffic cost more
// than the idle third of the last round.
// KS_ = keys per chunk.  128 (default): the measurements above.  64: 19.4 KiB of LDS and ~100 VGPRs per block instead of 38 KiB / 138 —
// small enough to sit on a CU BESIDE a 256 x 256 GEMM block of another stream (130 of the 160 KiB of LDS, half the register file):
// the HBM-bound attention of one in-flight batch can then run under the matrix-core-bound encode / prefill of the other
// (SL_ATTN_DECODE_KS=64; DESIGN §8.10 has what it measured).
template <int REP, bool PREFETCH, int KS_ = 128>
__global__ __launch_bounds__(256) void attn_decode_full_kernel(const bf16_t* __restrict__ q, int64_t q_stride, const bf16_t* __restrict__ kc,
                                                               const bf16_t* __restrict__ vc, bf16_t* __restrict__ out,
                                                               const int32_t* __restrict__ ctx_len, int ctx_add, int nh, int nkv, int max_ctx,
                                                               float scale, int shared_prefix) {
  using T = bf16_t;
  constexpr int D = 128, KS = KS_, NPS = KS / 16;
  static_assert(KS == 64 || KS == 128, "chunks of 64 or 128 keys");
  constexpr int PROW = KS * 2 + 16;
  __shared__ float sc[REP][KS];
  __shared__ float alpha[4], linv[4];
  __shared__ __attribute__((aligned(16))) unsigned char un[KS * D * 2];   // K tile, then V tile
  __shared__ __attribute__((aligned(16))) unsigned char pt[16 * PROW];
  const int kvh = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, grp = lane >> 4, gl = lane & 15;
  const int kg = tid >> 4, dc = tid & 15;
  const int r = lane & 15, q4 = lane >> 4, qq = r >> 2, pp = r & 3;
  const int n_keys = ctx_len[b] + ctx_add;
  const T* kbase = kc + ((int64_t)b * nkv + kvh) * max_ctx * D;
  const T* vbase = vc + ((int64_t)b * nkv + kvh) * max_ctx * D;
  // positions below shared_prefix hold the same rows in every slot (the caller's promise: one prompt prefix for the whole batch);
  // every block reads them from slot 0, so they come out of L2 instead of HBM.  Measured (tools/time_decode_step.py 1024 with
  // SHARED_PREFIX=9, three A/B rounds on one box, profiles/r04_w_shared_prefix.txt): decode step 11.08 -> 10.82 ms; the same rows
  // through plain (not non-temporal) loads behind a per-load branch: 11.30 ms, slower than no sharing.
  const T* kbase0 = kc + (int64_t)kvh * max_ctx * D;
  const T* vbase0 = vc + (int64_t)kvh * max_ctx * D;
  auto voff = [](int row, int ch) { return 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))); };

  uint4 qf[4];
#pragma unroll
  for (int s4 = 0; s4 < 4; ++s4)
    qf[s4] = r < REP ? *(const uint4*)(q + (int64_t)b * q_stride + (int64_t)(kvh * REP + r) * D + 32 * s4 + 8 * q4) : make_uint4(0, 0, 0, 0);
  for (int o = tid; o < (16 - REP) * (KS / 8); o += 256)   // padding rows of P (heads REP..15) stay zero
    *(uint4*)(pt + (REP + o / (KS / 8)) * PROW + (o % (KS / 8)) * 16) = make_uint4(0, 0, 0, 0);

  u32x4_t kraw[PREFETCH ? 2 : 1][NPS], vraw[PREFETCH ? 2 : 1][NPS];
  auto fetch_k = [&](int buf, int k0) {   // rows past the context are clamped (their scores are masked)
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      int key = k0 + ps * 16 + wave * 4 + grp; key = key < n_keys ? key : n_keys - 1;
      kraw[buf][ps] = SL_KV_LOAD((key < shared_prefix ? kbase0 : kbase) + (int64_t)key * D + gl * 8);
    }
  };
  auto fetch_v = [&](int buf, int k0) {
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      int key = k0 + kg + 16 * ps; key = key < n_keys ? key : n_keys - 1;
      vraw[buf][ps] = SL_KV_LOAD((key < shared_prefix ? vbase0 : vbase) + (int64_t)key * D + dc * 8);
    }
  };
  auto fetch = [&](int buf, int k0) { fetch_k(buf, k0); fetch_v(buf, k0); };
  f32x4 oacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  float m_run = -INFINITY, l_run = 0.f;   // live in wave h (< REP), replicated over its lanes
  const uint32_t ub = (uint32_t)(uintptr_t)(lds_ptr3_t)un;
  const int nchunk = (n_keys + KS - 1) / KS;
  if constexpr (PREFETCH) fetch(0, 0);
  for (int c0 = 0; c0 < nchunk; c0 += 2) {
#pragma unroll
    for (int u2 = 0; u2 < 2; ++u2) {
      constexpr int dummy = 0; (void)dummy;
      const int u = PREFETCH ? u2 : 0;
      const int ci = c0 + u2;
      if (ci >= nchunk) break;
      const int k0 = ci * KS;
      if constexpr (PREFETCH) { if (ci + 1 < nchunk) fetch(u ^ 1, k0 + KS); }
      else fetch_k(0, k0);
      // K tile -> LDS -> scores on the matrix core
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) {
        const int kl = ps * 16 + wave * 4 + grp;
        *(u32x4_t*)(un + kl * 256 + ((gl ^ (kl & 15)) << 4)) = kraw[u][ps];
      }
      if constexpr (!PREFETCH) fetch_v(0, k0);   // K registers are free again: V rows fly under the score / softmax phases
      __syncthreads();
#pragma unroll
      for (int n = 0; n < KS / 64; ++n) {
        const int rowbase = wave * (KS / 4) + n * 16;
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const uint4 kf = *(const uint4*)(un + (rowbase + r) * 256 + (((4 * s4 + q4) ^ r) << 4));
          MMA<T>::step(sacc, qf[s4], kf);
        }
        if (q4 == 0) {
          const int key = rowbase + r;
#pragma unroll
          for (int h = 0; h < REP; ++h) sc[h][key] = (k0 + key) < n_keys ? sacc[h] * scale : -INFINITY;
        }
      }
      __syncthreads();
      // online softmax: wave h owns head h (two keys per lane); P rounded to bf16 as HF eager does
      if (wave < REP) {
        const float s0 = sc[wave][lane], s1 = KS == 128 ? sc[wave][(lane + 64) & (KS - 1)] : -INFINITY;
        const float m_new = fmaxf(m_run, wave_max(fmaxf(s0, s1)));   // finite: key k0 is inside the context
        const float p0 = __expf(s0 - m_new), p1 = KS == 128 ? __expf(s1 - m_new) : 0.f;
        const float a = __expf(m_run - m_new);                        // first chunk: exp(-inf) = 0
        l_run = l_run * a + wave_sum(p0 + p1);
        m_run = m_new;
        *(T*)(pt + wave * PROW + lane * 2) = from_f32<T>(p0);
        if constexpr (KS == 128) *(T*)(pt + wave * PROW + (lane + 64) * 2) = from_f32<T>(p1);
        if (lane == 0) alpha[wave] = a;
      }
      // V tile over the (dead) K tile
#pragma unroll
      for (int ps = 0; ps < NPS; ++ps) *(u32x4_t*)(un + voff(kg + 16 * ps, dc)) = vraw[u][ps];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int df = 2 * wave + j;
#pragma unroll
        for (int h = 0; h < REP; ++h) oacc[j][h] *= alpha[h];
#pragma unroll
        for (int ks = 0; ks < KS / 32; ++ks) {
          const uint4 pa = *(const uint4*)(pt + r * PROW + (32 * ks + 8 * q4) * 2);
          const int r0 = 32 * ks + 8 * q4 + qq;
          u32x2_t lo, hi;
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo) : "v"(ub + (uint32_t)(voff(r0, 2 * df + (pp >> 1)) + 8 * (pp & 1))) : "memory");
          asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(hi) : "v"(ub + (uint32_t)(voff(r0 + 4, 2 * df + (pp >> 1)) + 8 * (pp & 1))) : "memory");
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi));
          MMA<T>::step(oacc[j], pa, make_uint4(lo.x, lo.y, hi.x, hi.y));
        }
      }
      __syncthreads();   // un / pt / sc / alpha are rewritten by the next chunk
    }
  }
  if (wave < REP && lane == 0) linv[wave] = l_run > 0.f ? 1.0f / l_run : 0.f;
  __syncthreads();
  if (q4 == 0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int h = 0; h < REP; ++h)
        out[(int64_t)b * nh * D + (int64_t)(kvh * REP + h) * D + (2 * wave + j) * 16 + r] = from_f32<T>(oacc[j][h] * linv[h]);
  }
}

template <typename T>
__global__ __launch_bounds__(128) void attn_decode_combine_kernel(const float* __restrict__ part, T* __restrict__ out, int nh, int nkv,
                                                                  int nsplit) {
  constexpr int D = 128;
  const int b = blockIdx.y, head = blockIdx.x, d = threadIdx.x;
  const int rep = nh / nkv, kvh = head / rep, h = head % rep;
  const int pstride = split_record_floats(rep);
  const float* base = part + ((int64_t)b * nkv + kvh) * nsplit * pstride;
  float M = -INFINITY;
  for (int s = 0; s < nsplit; ++s) M = fmaxf(M, base[(int64_t)s * pstride + rep * D + h]);
  float L = 0.f, o = 0.f;
  for (int s = 0; s < nsplit; ++s) {
    const float* rec = base + (int64_t)s * pstride;
    const float m = rec[rep * D + h];
    if (m == -INFINITY) continue;
    const float w = __expf(m - M);
    L += w * rec[rep * D + rep + h];
    o += w * rec[h * D + d];
  }
  out[((int64_t)b * nh + head) * D + d] = from_f32<T>(L > 0.f ? o / L : 0.f);
}

// workspace = the partial records, then (256-byte aligned) one arrival counter per (sequence, kv head) for the in-launch merge
static size_t attn_split_records_bytes(int B, int n_heads, int n_kv, int max_ctx) {
  const int rep = n_heads / n_kv, nsplit = (max_ctx + DSPLIT - 1) / DSPLIT;   // sized for the finer split
  return ((size_t)B * n_kv * nsplit * split_record_floats(rep) * sizeof(float) + 255) & ~(size_t)255;
}
size_t sl_attn_decode_split_ws(int B, int n_heads, int n_kv, int max_ctx) {
  return attn_split_records_bytes(B, n_heads, n_kv, max_ctx) + (((size_t)B * n_kv * sizeof(int32_t) + 255) & ~(size_t)255);
}
// The counters must be zero when a launch starts; every launch leaves them zero.  The owner of the workspace zeroes them once
// (the decode runtime: at the start of a generate / decode-step call, outside the captured graph).
int sl_attn_decode_split_zero_counters(void* workspace, int B, int n_heads, int n_kv, int max_ctx, hipStream_t st) {
  SL_HIP(hipMemsetAsync((unsigned char*)workspace + attn_split_records_bytes(B, n_heads, n_kv, max_ctx), 0, (size_t)B * n_kv * sizeof(int32_t), st));
  return 0;
}

template <typename T, int REP>
static int launch_attn_decode_split(const void* q, int64_t q_stride, const void* kc, const void* vc, void* out, float* part,
                                    const int32_t* ctx_len, int ctx_add, int B, int nh, int nkv, int max_ctx, float scale, hipStream_t st,
                                    int32_t* cnt, int shared_prefix) {
  if constexpr (sizeof(T) == 2) {
    const int64_t full_min = sl_env().attn_full_min;   // tuning switch: (sequence, kv head) pairs from which the single-pass form runs; measured faster than split + merge from B = 4 up (9.1 vs 11.4 us), 97 vs 127 us at B = 512
    // ... except long caches at batches that leave the chip under-filled: a block of the single-pass form walks its whole context in
    // 128-key chunks one after the other (16 sequences x 8 kv heads = 128 blocks, up to 14 chunks each at 1 800 keys), the split
    // form spreads the same keys over (sequence, kv head, 64 keys) blocks: long-form leg of bench.py (16 utterances of 30-120 s,
    // contexts up to 1 782) decode 680 -> 627 ms; at contexts of a few hundred keys the single pass stays ahead (1.88 vs 1.94 ms
    // per step at 16 sequences, 709 vs 743 ms for the Whisper leg's 32 sequences)
    const int64_t Bf = sl_family_rows(B);              // the form follows the pinned family's rows (common.h sl_family_rows)
    const bool long_thin = max_ctx >= 1024 && Bf * nkv < 768;
    if (Bf * nkv >= full_min && !long_thin && !sl_env().attn_force_split) {
      if (sl_env().attn_decode_ks == 65)        // 64-key chunks with the NEXT chunk's K / V rows held in a second register set (twice the bytes in flight per block)
        hipLaunchKernelGGL((attn_decode_full_kernel<REP, true, 64>), dim3(nkv, B), dim3(256), 0, st, (const bf16_t*)q, q_stride, (const bf16_t*)kc,
                           (const bf16_t*)vc, (bf16_t*)out, ctx_len, ctx_add, nh, nkv, max_ctx, scale, shared_prefix);
      else if (sl_env().attn_decode_ks == 64)
        hipLaunchKernelGGL((attn_decode_full_kernel<REP, false, 64>), dim3(nkv, B), dim3(256), 0, st, (const bf16_t*)q, q_stride, (const bf16_t*)kc,
                           (const bf16_t*)vc, (bf16_t*)out, ctx_len, ctx_add, nh, nkv, max_ctx, scale, shared_prefix);
      else
        hipLaunchKernelGGL((attn_decode_full_kernel<REP, false>), dim3(nkv, B), dim3(256), 0, st, (const bf16_t*)q, q_stride, (const bf16_t*)kc,
                           (const bf16_t*)vc, (bf16_t*)out, ctx_len, ctx_add, nh, nkv, max_ctx, scale, shared_prefix);
      SL_CHECK_LAUNCH("attn_decode_full");
      return 0;
    }
  }
  int nsplit;
  if ((int64_t)sl_family_rows(B) * nkv >= 512) {
    nsplit = (max_ctx + 127) / 128;
    hipLaunchKernelGGL((attn_decode_split_kernel<T, REP, 128>), dim3(nkv, B, nsplit), dim3(256), 0, st, (const T*)q, q_stride, (const T*)kc,
                       (const T*)vc, part, ctx_len, ctx_add, nkv, max_ctx, scale, cnt, (T*)out);
  } else {
    nsplit = (max_ctx + DSPLIT - 1) / DSPLIT;
    hipLaunchKernelGGL((attn_decode_split_kernel<T, REP, DSPLIT>), dim3(nkv, B, nsplit), dim3(256), 0, st, (const T*)q, q_stride, (const T*)kc,
                       (const T*)vc, part, ctx_len, ctx_add, nkv, max_ctx, scale, cnt, (T*)out);
  }
  SL_CHECK_LAUNCH("attn_decode_split");
  if (cnt) return 0;   // the last block of every (sequence, kv head) merged its records
  hipLaunchKernelGGL((attn_decode_combine_kernel<T>), dim3(nh, B), dim3(128), 0, st, part, (T*)out, nh, nkv, nsplit);
  SL_CHECK_LAUNCH("attn_decode_combine");
  return 0;
}

// counters: 0 = separate combine launch; 1 = merge inside the split launch, the caller has zeroed the counters
// (sl_attn_decode_split_zero_counters) on this stream; 2 = the same, zeroing them here first (one memset per call)
int sl_attn_decode_split_impl(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache, void* out, void* workspace,
                              const int32_t* ctx_len, int ctx_add, int32_t B, int32_t n_heads, int32_t n_kv, int32_t D, int32_t max_ctx,
                              float scale, int32_t dtype, hipStream_t st, int counters, int shared_prefix) {
  SL_CHECK_ARG(q && k_cache && v_cache && out && workspace && ctx_len && B > 0, "sl_attn_decode_split: bad arguments");
  SL_CHECK_ARG(D == 128, "sl_attn_decode_split: head_dim %d not built (128)", D);
  SL_CHECK_ARG(n_kv > 0 && n_heads % n_kv == 0, "sl_attn_decode_split: n_heads %% n_kv != 0");
  SL_CHECK_ARG(shared_prefix >= 0 && shared_prefix <= max_ctx, "sl_attn_decode_split: shared_prefix %d outside [0, max_ctx=%d]", shared_prefix, max_ctx);
  float* part = (float*)workspace;
  const int rep = n_heads / n_kv;
  // Measured (tools/time_decode_step.py, Llama-3.2-3B, 128 new tokens, 7 splits x 8 kv heads): merged in-launch 1.5725 / 1.5884 / 1.5922 ms
  // per step at batch 1 / 2 / 3 against 1.6136 / 1.6213 / 1.6303 ms with the separate combine launch (the merge costs an arrival-counter
  // round trip plus two batches of sc1 re-reads, ~3 us, the launch boundary + combine kernel ~4.5 us).  A first version that waited
  // for every sc1 load on its own was SLOWER than the combine launch (1.778 vs 1.640 ms per token).  Default: merge where the split
  // path serves small batches (B * n_kv <= 32); large fp32 batches keep the combine launch (one counter per (sequence, kv head)).
  // SL_ATTN_SPLIT_MERGE = 0 / 1 forces it off / on.
  const int mode = sl_env().attn_split_merge;
  if (mode == 0 || (mode < 0 && (int64_t)sl_family_rows(B) * n_kv > 32)) counters = 0;
  int32_t* cnt = counters ? (int32_t*)((unsigned char*)workspace + attn_split_records_bytes(B, n_heads, n_kv, max_ctx)) : nullptr;
  if (counters == 2) SL_TRY(sl_attn_decode_split_zero_counters(workspace, B, n_heads, n_kv, max_ctx, st));
  SL_DISPATCH_DTYPE(dtype, T, {
    switch (rep) {
      case 1: return launch_attn_decode_split<T, 1>(q, q_stride, k_cache, v_cache, out, part, ctx_len, ctx_add, B, n_heads, n_kv, max_ctx, scale, st, cnt, shared_prefix);
      case 2: return launch_attn_decode_split<T, 2>(q, q_stride, k_cache, v_cache, out, part, ctx_len, ctx_add, B, n_heads, n_kv, max_ctx, scale, st, cnt, shared_prefix);
      case 3: return launch_attn_decode_split<T, 3>(q, q_stride, k_cache, v_cache, out, part, ctx_len, ctx_add, B, n_heads, n_kv, max_ctx, scale, st, cnt, shared_prefix);
      case 4: return launch_attn_decode_split<T, 4>(q, q_stride, k_cache, v_cache, out, part, ctx_len, ctx_add, B, n_heads, n_kv, max_ctx, scale, st, cnt, shared_prefix);
      default: sl_set_error("sl_attn_decode_split: n_heads/n_kv=%d not built (1..4)", rep); return SL_ERR_UNSUPPORTED;
    }
  });
}

extern "C" size_t sl_attn_decode_workspace_bytes(int32_t B, int32_t n_heads, int32_t n_kv, int32_t max_ctx) {
  return sl_attn_decode_split_ws(B, n_heads, n_kv, max_ctx);
}

extern "C" int sl_attn_decode_split(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache, void* out, void* workspace,
                                    const int32_t* ctx_len, int32_t B, int32_t n_heads, int32_t n_kv, int32_t D, int32_t max_ctx, float scale,
                                    int32_t dtype, sl_stream stream) {
  return sl_attn_decode_split_impl(q, q_stride, k_cache, v_cache, out, workspace, ctx_len, 0, B, n_heads, n_kv, D, max_ctx, scale, dtype,
                                   (hipStream_t)stream, 2, 0);
}
