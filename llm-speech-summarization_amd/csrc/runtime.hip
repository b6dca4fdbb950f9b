// runtime.hip — C++ host runtime above the kernels: HuBERT encoder forward over a batch of utterances,
// Llama prefill / decode step / greedy generation (decode step captured once into a hipGraph and
// replayed, all per-step state lives on the device).  Workspace is carved from one caller-owned buffer.
#include <vector>

#include "common.h"

int sl_attn_decode_impl(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache, void* out, const int32_t* ctx_len,
                        int ctx_add, int32_t B, int32_t n_heads, int32_t n_kv, int32_t D, int32_t max_ctx, float scale, int32_t dtype,
                        hipStream_t st);
int sl_greedy_select_impl(const float* logits, int32_t B, int32_t V, const int32_t* eos_ids, int32_t n_eos, int32_t pad_id,
                          int32_t use_eos, int32_t advance_ctx, int32_t* unfinished, int32_t* ctx_len, int32_t* gen_count,
                          int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new, hipStream_t st, const int32_t* row_limit);
int sl_greedy_select_partial_impl(const float* amax_val, const int32_t* amax_idx, int32_t n_groups, int32_t B, const int32_t* eos_ids, int32_t n_eos,
                                  int32_t pad_id, int32_t use_eos, int32_t advance_ctx, int32_t* unfinished, int32_t* ctx_len, int32_t* gen_count,
                                  int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new, hipStream_t st, const int32_t* row_limit);
int sl_sample_select_impl(const float* logits, int32_t B, int32_t V, float temperature, int32_t top_k, float top_p, uint64_t seed,
                          const int32_t* eos_ids, int32_t n_eos, int32_t pad_id, int32_t use_eos, int32_t advance_ctx, int32_t* unfinished,
                          int32_t* ctx_len, int32_t* gen_count, int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new,
                          int32_t* choice_ws, hipStream_t st, const int32_t* row_limit, const int32_t* row_ids);

int sl_attn_decode_split_zero_counters(void* workspace, int B, int n_heads, int n_kv, int max_ctx, hipStream_t st);
int sl_attn_decode_split_impl(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache, void* out, void* workspace,
                              const int32_t* ctx_len, int ctx_add, int32_t B, int32_t n_heads, int32_t n_kv, int32_t D, int32_t max_ctx,
                              float scale, int32_t dtype, hipStream_t st, int counters, int shared_prefix);
size_t sl_attn_decode_split_ws(int B, int n_heads, int n_kv, int max_ctx);
int sl_rmsnorm_rstd_impl(const void* x, void* y, const void* w, float* rstd_out, int64_t rows, int32_t cols, float eps, int32_t dtype, hipStream_t st);
int sl_gemm_impl(const sl_gemm_args* a, const sl_gemm_fused* fx, const sl_gemm_ex_args* ex, hipStream_t st);
bool sl_gemm_rows_epilogue_ok(int M, int N, int K, int dtype);

namespace {

struct Carver {
  unsigned char* base;
  size_t off = 0, cap;
  Carver(void* b, size_t c) : base((unsigned char*)b), cap(c) {}
  void* take(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* p = base ? base + off : nullptr;
    off += bytes;
    return p;
  }
  bool ok() const { return off <= cap; }
};

int gemm(int dtype, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* res,
         int64_t ldr, int M, int N, int K, int act, int out_f32, hipStream_t st) {
  sl_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = bias; a.residual = res; a.ldr = ldr;
  a.M = M; a.N = N; a.K = K; a.batch = 1; a.dtype = dtype; a.act = act; a.out_f32 = out_f32;
  return sl_gemm(&a, (sl_stream)st);
}

inline unsigned char* bptr(void* p) { return (unsigned char*)p; }

}  // namespace

// ================================================================================================
// HuBERT
// ================================================================================================
extern "C" int sl_hubert_num_frames(const sl_hubert_model* m, int64_t n_samples) {
  int64_t L = n_samples;
  for (int i = 0; i < m->n_conv; ++i) {
    if (L < m->conv_kernel[i]) return 0;
    L = (L - m->conv_kernel[i]) / m->conv_stride[i] + 1;
  }
  return (int)L;
}

struct HubertPlan {
  int64_t max_conv_elems = 0;  // largest conv activation (elements) over layers and utterances
  int64_t total_T = 0, max_T = 0, total_P = 0;
  int64_t conv_elems[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  std::vector<int64_t> T, P, tok0;
};

static int hubert_plan(const sl_hubert_model* m, const int64_t* offs, int n_utt, HubertPlan& pl) {
  pl.T.resize(n_utt); pl.P.resize(n_utt); pl.tok0.resize(n_utt + 1);
  for (int u = 0; u < n_utt; ++u) {
    int64_t L = offs[u + 1] - offs[u];
    for (int i = 0; i < m->n_conv; ++i) {
      SL_CHECK_ARG(L >= m->conv_kernel[i], "sl_hubert_forward: utterance %d too short (%lld samples)", u, (long long)(offs[u + 1] - offs[u]));
      L = (L - m->conv_kernel[i]) / m->conv_stride[i] + 1;
      pl.conv_elems[i] += L * m->conv_dim[i];   // conv activations of the whole batch are packed per layer
      if (pl.conv_elems[i] > pl.max_conv_elems) pl.max_conv_elems = pl.conv_elems[i];
    }
    pl.T[u] = L;
    SL_CHECK_ARG(L >= m->pool_kernel, "sl_hubert_forward: utterance %d gives %lld frames < pool kernel %d", u, (long long)L, m->pool_kernel);
    pl.P[u] = (L - m->pool_kernel) / m->pool_stride + 1;
    pl.tok0[u] = pl.total_T;
    pl.total_T += L;
    pl.total_P += pl.P[u];
    if (L > pl.max_T) pl.max_T = L;
  }
  pl.tok0[n_utt] = pl.total_T;
  return 0;
}

struct HubertWs {
  void *convA, *convB, *feat, *x, *ln, *qkv, *att, *mid, *xg, *pooled;
  int32_t *cu, *cuk, *klen;
  int64_t* desc;
  int64_t* c0desc;  // n_utt + 1 sample offsets, then n_utt + 1 row offsets of the conv0 output (batched conv0 launch)
  int64_t* pdesc;   // per utterance {pooled rows, element offset into `pooled`, element offset into `out`, 0}: AvgPool + grouped projector
  float* ln_stats;  // LayerNorm fold: [frames][hidden / 64][2] partial {sum, sum of squares} out of the out_proj / FFN2 epilogues
  float* ln_mr;     //                 [frames][2] {mean, rstd}
};

static size_t hubert_carve(const sl_hubert_model* m, const HubertPlan& pl, int n_utt, void* base, size_t cap, HubertWs& w) {
  const size_t sz = sl_dtype_size(m->dtype);
  const int H = m->hidden;
  Carver c(base, cap);
  w.convA = c.take(pl.max_conv_elems * sz);
  w.convB = c.take(pl.max_conv_elems * sz);
  w.feat = c.take(pl.total_T * m->conv_dim[m->n_conv - 1] * sz);
  w.x = c.take(pl.total_T * H * sz);
  w.ln = c.take(pl.total_T * H * sz);
  w.qkv = c.take(pl.total_T * 3 * H * sz);
  w.att = c.take(pl.total_T * H * sz);
  w.mid = c.take(pl.total_T * (int64_t)m->ffn * sz);
  w.xg = c.take((pl.total_T + (int64_t)n_utt * m->pos_k) * H * sz);
  w.desc = (int64_t*)c.take((size_t)((m->n_conv - 1) * n_utt + n_utt * m->pos_groups) * 4 * sizeof(int64_t));
  w.pooled = c.take(pl.total_P * H * sz);
  w.pdesc = (int64_t*)c.take((size_t)n_utt * 4 * sizeof(int64_t));
  w.c0desc = (int64_t*)c.take((size_t)(n_utt + 1) * 2 * sizeof(int64_t));
  w.cu = (int32_t*)c.take((n_utt + 1) * sizeof(int32_t));
  w.cuk = (int32_t*)c.take(n_utt * sizeof(int32_t));
  w.klen = (int32_t*)c.take(n_utt * sizeof(int32_t));
  w.ln_stats = (float*)c.take((size_t)pl.total_T * (size_t)((H + 63) / 64) * 2 * sizeof(float));
  w.ln_mr = (float*)c.take((size_t)pl.total_T * 2 * sizeof(float));
  return c.off + 256;
}

// records of the pooled rows / projector output rows of every utterance (see HubertWs::pdesc)
static void proj_records(const sl_hubert_model* m, const HubertPlan& pl, int n_utt, int64_t out_ld, const int64_t* out_row_offsets_host,
                         std::vector<int64_t>& rec) {
  rec.assign((size_t)n_utt * 4, 0);
  int64_t prow = 0;
  for (int u = 0; u < n_utt; ++u) {
    rec[4 * u] = pl.P[u];
    rec[4 * u + 1] = prow * m->hidden;
    rec[4 * u + 2] = (out_row_offsets_host ? out_row_offsets_host[u] : prow) * out_ld;
    prow += pl.P[u];
  }
}

// transformer layers on the packed frames of the batch (varlen attention), final LayerNorm, AvgPool + projection.
// Shared by the HuBERT and Whisper front ends (both are 16x64 pre-LN encoders: hf:...hubert.py:504-547,
// hf:models/whisper/modeling_whisper.py:360-414).
static int encoder_tail(const sl_hubert_model* m, HubertWs& w, const HubertPlan& pl, int n_utt, void* out, int64_t out_ld,
                        const int64_t* out_row_offsets_host, void* last_hidden, hipStream_t st) {
  sl_stream stream = (sl_stream)st;
  const int dt = m->dtype, H = m->hidden, NT = (int)pl.total_T;
  const size_t sz = sl_dtype_size(dt);
  // LayerNorm fold (bf16 inference, weights prepared by the caller: m->fold): no LayerNorm pass over the frames inside the layers.
  // x -> [q|k|v Linear with ln1 folded in, row statistics applied in its epilogue] -> attention -> out_proj (+ bias + residual;
  // its epilogue leaves the row statistics of the new x) -> [FFN1 with ln2 folded in, GELU] -> FFN2 (+ bias + residual, statistics
  // again).  Two LayerNorm launches per layer (160 us each at 255 k frames, HBM-bound) become two 8 us finalize launches.
  const bool fold = m->fold != nullptr && dt == SL_BF16 && !sl_env().no_ln_fold && H % 64 == 0 && m->ffn % 64 == 0 &&
                    sl_gemm_rows_epilogue_ok(NT, 3 * H, H, dt) && sl_gemm_rows_epilogue_ok(NT, H, H, dt) &&
                    sl_gemm_rows_epilogue_ok(NT, m->ffn, H, dt) && sl_gemm_rows_epilogue_ok(NT, H, m->ffn, dt);
  auto gemm_x = [&](const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* res, int N, int K, int act,
                    const float* ln_u, const float* ln_c, bool stats) -> int {
    sl_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = bias; a.residual = res; a.ldr = res ? ldc : 0;
    a.M = NT; a.N = N; a.K = K; a.batch = 1; a.dtype = dt; a.act = act;
    sl_gemm_ex_args ex;
    memset(&ex, 0, sizeof(ex));
    ex.w_mod = 1;
    if (ln_u) { ex.ln_mr = w.ln_mr; ex.ln_u = ln_u; ex.ln_c = ln_c; }
    if (stats) ex.stats_out = w.ln_stats;
    return sl_gemm_impl(&a, nullptr, &ex, st);
  };
  if (fold) SL_TRY(sl_layernorm_stats(w.x, NT, H, m->ln_eps, w.ln_mr, dt, stream));       // the first layer's input comes out of the positional conv
  for (int l = 0; l < m->n_layers; ++l) {
    const sl_hubert_layer& L = m->layers[l];
    if (fold) {
      const sl_hubert_fold& F = m->fold[l];
      SL_TRY(gemm_x(w.x, H, F.wqkv_f, H, w.qkv, 3 * H, nullptr, nullptr, 3 * H, H, SL_ACT_NONE, F.uqkv, F.cqkv, false));
    } else {
      SL_TRY(sl_layernorm(w.x, w.ln, L.ln1_g, L.ln1_b, NT, H, m->ln_eps, 0, dt, stream));
      SL_TRY(gemm(dt, w.ln, H, L.wqkv, H, w.qkv, 3 * H, L.bqkv, nullptr, 0, NT, 3 * H, H, SL_ACT_NONE, 0, st));
    }
    sl_attn_args a;
    memset(&a, 0, sizeof(a));
    a.q = w.qkv; a.q_row_stride = 3 * H; a.q_head_stride = 64;
    a.k = bptr(w.qkv) + (size_t)H * sz; a.k_row_stride = 3 * H; a.k_head_stride = 64;
    a.v = bptr(w.qkv) + (size_t)2 * H * sz; a.v_row_stride = 3 * H; a.v_head_stride = 64;
    a.out = w.att; a.o_row_stride = H; a.o_head_stride = 64;
    a.cu_q = w.cu; a.cu_k = w.cuk; a.klen = w.klen;
    a.nseq = n_utt; a.max_qlen = (int)pl.max_T; a.n_heads = m->n_heads; a.n_kv_heads = m->n_heads; a.head_dim = 64; a.causal = 0;
    a.dtype = dt; a.scale = 0.125f;
    SL_TRY(sl_attn_fwd(&a, stream));
    if (fold) {
      const sl_hubert_fold& F = m->fold[l];
      SL_TRY(gemm_x(w.att, H, L.wo, H, w.x, H, L.bo, w.x, H, H, SL_ACT_NONE, nullptr, nullptr, true));
      SL_TRY(sl_layernorm_stats_finalize(w.ln_stats, H / 64, NT, H, m->ln_eps, w.ln_mr, stream));
      SL_TRY(gemm_x(w.x, H, F.w1_f, H, w.mid, m->ffn, nullptr, nullptr, m->ffn, H, SL_ACT_GELU, F.u1, F.c1, false));
      const bool more = l + 1 < m->n_layers;       // the final LayerNorm keeps its own kernel (its output is the product)
      SL_TRY(gemm_x(w.mid, m->ffn, L.w2, m->ffn, w.x, H, L.b2, w.x, H, m->ffn, SL_ACT_NONE, nullptr, nullptr, more));
      if (more) SL_TRY(sl_layernorm_stats_finalize(w.ln_stats, H / 64, NT, H, m->ln_eps, w.ln_mr, stream));
      continue;
    }
    SL_TRY(gemm(dt, w.att, H, L.wo, H, w.x, H, L.bo, w.x, H, NT, H, H, SL_ACT_NONE, 0, st));
    SL_TRY(sl_layernorm(w.x, w.ln, L.ln2_g, L.ln2_b, NT, H, m->ln_eps, 0, dt, stream));
    SL_TRY(gemm(dt, w.ln, H, L.w1, H, w.mid, m->ffn, L.b1, nullptr, 0, NT, m->ffn, H, SL_ACT_GELU, 0, st));
    SL_TRY(gemm(dt, w.mid, m->ffn, L.w2, m->ffn, w.x, H, L.b2, w.x, H, NT, H, m->ffn, SL_ACT_NONE, 0, st));
  }
  void* lh = last_hidden ? last_hidden : w.ln;
  SL_TRY(sl_layernorm(w.x, lh, m->final_ln_g, m->final_ln_b, NT, H, m->ln_eps, 0, dt, stream));
  // ---- AvgPool over time + projection into the caller's (prompt) buffer
  if (!m->proj_w) return 0;  // stack / ctc_pool: the host finishes from last_hidden
  // one AvgPool launch and one grouped projector GEMM for the whole ragged batch (records uploaded by the caller: w.pdesc)
  int64_t max_P = 0;
  for (int u = 0; u < n_utt; ++u) max_P = pl.P[u] > max_P ? pl.P[u] : max_P;
  SL_TRY(sl_avgpool_batch(lh, w.pooled, w.cu, w.klen, w.pdesc, n_utt, max_P, H, m->pool_kernel, m->pool_stride, dt, stream));
  sl_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = w.pooled; a.lda = H;
  a.W = m->proj_w; a.ldw = H;
  a.C = out; a.ldc = out_ld;
  a.bias = m->proj_b;
  a.M = (int)max_P; a.N = m->llm_dim; a.K = H; a.batch = n_utt; a.dtype = dt; a.act = SL_ACT_NONE;
  sl_gemm_ex_args ex;
  memset(&ex, 0, sizeof(ex));
  ex.groups = w.pdesc; ex.w_mod = 1;
  SL_TRY(sl_gemm_impl(&a, nullptr, &ex, st));
  return 0;
}

extern "C" size_t sl_hubert_workspace_bytes(const sl_hubert_model* m, const int64_t* sample_offsets_host, int32_t n_utt) {
  HubertPlan pl;
  if (hubert_plan(m, sample_offsets_host, n_utt, pl) != 0) return 0;
  HubertWs w;
  return hubert_carve(m, pl, n_utt, nullptr, 0, w);
}

extern "C" int sl_hubert_forward(const sl_hubert_model* m, const float* waves, const int64_t* sample_offsets_host, int32_t n_utt,
                                 void* out, int64_t out_ld, const int64_t* out_row_offsets_host, void* last_hidden, void* workspace,
                                 size_t workspace_bytes, sl_stream stream) {
  SL_CHECK_ARG(m && waves && sample_offsets_host && workspace && n_utt > 0, "sl_hubert_forward: bad arguments");
  SL_CHECK_ARG((m->proj_w && out) || (!m->proj_w && last_hidden), "sl_hubert_forward: need `out` (pool) or `last_hidden` (host-side downsample)");
  SL_CHECK_ARG(m->n_conv >= 2 && m->n_conv <= 8 && m->hidden % m->n_heads == 0 && m->hidden / m->n_heads == 64,
               "sl_hubert_forward: need 2..8 conv layers and head_dim 64 (hidden=%d heads=%d)", m->hidden, m->n_heads);
  hipStream_t st = (hipStream_t)stream;
  const int dt = m->dtype;
  const int H = m->hidden;
  HubertPlan pl;
  SL_TRY(hubert_plan(m, sample_offsets_host, n_utt, pl));
  HubertWs w;
  const size_t need = hubert_carve(m, pl, n_utt, workspace, workspace_bytes, w);
  SL_CHECK_ARG(need <= workspace_bytes, "sl_hubert_forward: workspace %zu B < required %zu B", workspace_bytes, need);
  const int Cl = m->conv_dim[m->n_conv - 1];

  // ---- descriptors of the ragged batch: row offsets of every utterance in each packed conv activation,
  //      grouped-GEMM records {M, a_off, c_off, r_off} for conv layers 1.. and for the positional conv
  const int nc = m->n_conv, G = m->pos_groups, Hg = H / G, kpos = m->pos_k;
  std::vector<std::vector<int64_t>> row0(nc, std::vector<int64_t>(n_utt + 1, 0)), Lc(nc, std::vector<int64_t>(n_utt, 0));
  for (int u = 0; u < n_utt; ++u) {
    int64_t L = sample_offsets_host[u + 1] - sample_offsets_host[u];
    for (int i = 0; i < nc; ++i) { L = (L - m->conv_kernel[i]) / m->conv_stride[i] + 1; Lc[i][u] = L; }
  }
  for (int i = 0; i < nc; ++i)
    for (int u = 0; u < n_utt; ++u) row0[i][u + 1] = row0[i][u] + Lc[i][u];
  std::vector<int64_t> desc((size_t)((nc - 1) * n_utt + n_utt * G) * 4);
  std::vector<int64_t> max_rows(nc, 0);
  for (int i = 1; i < nc; ++i)
    for (int u = 0; u < n_utt; ++u) {
      int64_t* d = &desc[(size_t)((i - 1) * n_utt + u) * 4];
      d[0] = Lc[i][u]; d[1] = row0[i - 1][u] * m->conv_dim[i - 1]; d[2] = row0[i][u] * m->conv_dim[i]; d[3] = 0;
      if (Lc[i][u] > max_rows[i]) max_rows[i] = Lc[i][u];
    }
  std::vector<int64_t> xg_off(n_utt + 1, 0);
  for (int u = 0; u < n_utt; ++u) xg_off[u + 1] = xg_off[u] + (pl.T[u] + kpos) * H;
  const size_t pos_desc0 = (size_t)(nc - 1) * n_utt * 4;
  for (int u = 0; u < n_utt; ++u)
    for (int g = 0; g < G; ++g) {
      int64_t* d = &desc[pos_desc0 + (size_t)(u * G + g) * 4];
      d[0] = pl.T[u]; d[1] = xg_off[u] + (int64_t)g * (pl.T[u] + kpos) * Hg; d[2] = pl.tok0[u] * H + (int64_t)g * Hg; d[3] = d[2];
    }
  {
    std::vector<int32_t> cu(n_utt + 1), kl(n_utt);
    for (int u = 0; u <= n_utt; ++u) cu[u] = (int32_t)pl.tok0[u];
    for (int u = 0; u < n_utt; ++u) kl[u] = (int32_t)pl.T[u];
    SL_HIP(hipMemcpyAsync(w.cu, cu.data(), (n_utt + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    SL_HIP(hipMemcpyAsync(w.cuk, cu.data(), n_utt * sizeof(int32_t), hipMemcpyHostToDevice, st));
    SL_HIP(hipMemcpyAsync(w.klen, kl.data(), n_utt * sizeof(int32_t), hipMemcpyHostToDevice, st));
    SL_HIP(hipMemcpyAsync(w.desc, desc.data(), desc.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    std::vector<int64_t> prec;
    proj_records(m, pl, n_utt, out_ld, out_row_offsets_host, prec);
    SL_HIP(hipMemcpyAsync(w.pdesc, prec.data(), prec.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    std::vector<int64_t> c0(2 * (size_t)(n_utt + 1));
    for (int u = 0; u <= n_utt; ++u) { c0[u] = sample_offsets_host[u] - sample_offsets_host[0]; c0[n_utt + 1 + u] = row0[0][u]; }
    SL_HIP(hipMemcpyAsync(w.c0desc, c0.data(), c0.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    SL_HIP(hipStreamSynchronize(st));  // host vectors go out of scope; pageable copies are staged but be explicit
  }
  // ---- conv feature extractor: layer 0 per utterance (fused conv+LN+GELU), layers 1.. as ONE grouped implicit GEMM
  //      + ONE LayerNorm+GELU over the packed rows of the whole batch
  void* cur = w.convA;
  {
    int64_t max_L0 = 0;
    for (int u = 0; u < n_utt; ++u) max_L0 = Lc[0][u] > max_L0 ? Lc[0][u] : max_L0;
    SL_TRY(sl_hubert_conv0_batch(waves + sample_offsets_host[0], w.c0desc, w.c0desc + n_utt + 1, n_utt, max_L0, m->conv0_w, m->conv0_b, m->conv0_g,
                                 m->conv0_beta, cur, m->conv_dim[0], m->conv_kernel[0], m->conv_stride[0], 1e-5f, dt, stream));
  }
  for (int i = 1; i < nc; ++i) {
    const int Cin = m->conv_dim[i - 1], Cout = m->conv_dim[i], k = m->conv_kernel[i], s = m->conv_stride[i];
    void* dst = (i == nc - 1) ? w.feat : (cur == w.convA ? w.convB : w.convA);
    sl_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = cur; a.lda = (int64_t)s * Cin;   // implicit GEMM: output row t reads the k*Cin contiguous elements from input row t*s
    a.W = m->conv_w[i]; a.ldw = (int64_t)k * Cin;
    a.C = dst; a.ldc = Cout; a.bias = m->conv_b[i];
    a.M = (int)max_rows[i]; a.N = Cout; a.K = k * Cin; a.batch = n_utt; a.dtype = dt; a.act = SL_ACT_NONE;
    sl_gemm_ex_args ex;
    memset(&ex, 0, sizeof(ex));
    ex.groups = w.desc + (size_t)(i - 1) * n_utt * 4; ex.w_mod = 1;
    SL_TRY(sl_gemm_impl(&a, nullptr, &ex, st));
    SL_TRY(sl_layernorm(dst, dst, m->conv_g[i], m->conv_beta[i], row0[i][n_utt], Cout, 1e-5f, 1, dt, stream));
    cur = dst;
  }
  // ---- feature projection on the packed tokens
  const int NT = (int)pl.total_T;
  SL_TRY(sl_layernorm(w.feat, w.feat, m->fp_ln_g, m->fp_ln_b, NT, Cl, m->ln_eps, 0, dt, stream));
  SL_TRY(gemm(dt, w.feat, Cl, m->fp_w, Cl, w.x, H, m->fp_b, nullptr, 0, NT, H, Cl, SL_ACT_NONE, 0, st));
  // ---- positional conv embedding: x += gelu(grouped_conv(x)); every (utterance, group) is one record of ONE grouped GEMM
  {
    SL_TRY(sl_posconv_stage_batch(w.x, w.xg, w.cu, w.klen, n_utt, pl.max_T, H, G, kpos, dt, stream));
    sl_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = w.xg; a.lda = Hg;
    a.W = m->pos_w; a.ldw = (int64_t)kpos * Hg; a.strideW = (int64_t)Hg * kpos * Hg;
    a.C = w.ln; a.ldc = H;
    a.bias = m->pos_b; a.strideBias = Hg;
    a.residual = w.x; a.ldr = H;
    a.M = (int)pl.max_T; a.N = Hg; a.K = kpos * Hg; a.batch = n_utt * G; a.dtype = dt; a.act = SL_ACT_GELU;
    sl_gemm_ex_args ex;
    memset(&ex, 0, sizeof(ex));
    ex.groups = w.desc + pos_desc0; ex.w_mod = G;
    SL_TRY(sl_gemm_impl(&a, nullptr, &ex, st));
    void* t = w.x; w.x = w.ln; w.ln = t;
  }
  return encoder_tail(m, w, pl, n_utt, out, out_ld, out_row_offsets_host, last_hidden, st);
}

// ================================================================================================
// Whisper encoder (hf:models/whisper/modeling_whisper.py:592-646): conv1(k3,p1)+GELU, conv2(k3,s2,p1)+GELU as implicit
// GEMMs over zero-haloed channel-last rows, + positional table, then the shared transformer tail.
// Model struct reuse: conv_dim[0] = n_mel, conv_w/b[1] = conv1, conv_w/b[2] = conv2 (tap-major), pos_w = embed_positions
// (pos_k rows = max_source_positions), frontend = 1.
// ================================================================================================
static int whisper_plan(const sl_hubert_model* m, int n_utt, HubertPlan& pl) {
  const int64_t T = m->pos_k;
  pl.T.assign(n_utt, T); pl.P.resize(n_utt); pl.tok0.resize(n_utt + 1);
  SL_CHECK_ARG(T >= m->pool_kernel, "sl_whisper_forward: %lld frames < pool kernel %d", (long long)T, m->pool_kernel);
  for (int u = 0; u < n_utt; ++u) { pl.P[u] = (T - m->pool_kernel) / m->pool_stride + 1; pl.tok0[u] = (int64_t)u * T; pl.total_P += pl.P[u]; }
  pl.tok0[n_utt] = (int64_t)n_utt * T;
  pl.total_T = (int64_t)n_utt * T; pl.max_T = T;
  // conv buffers: haloed mel (2T+2, n_mel) and haloed conv1 output (2T+2, H) per utterance
  pl.max_conv_elems = (int64_t)n_utt * (2 * T + 2) * (m->hidden > m->conv_dim[0] ? m->hidden : m->conv_dim[0]);
  return 0;
}

extern "C" size_t sl_whisper_workspace_bytes(const sl_hubert_model* m, int32_t n_utt) {
  HubertPlan pl;
  if (whisper_plan(m, n_utt, pl) != 0) return 0;
  HubertWs w;
  return hubert_carve(m, pl, n_utt, nullptr, 0, w);
}

extern "C" int sl_whisper_forward(const sl_hubert_model* m, const void* mel, int32_t n_utt, void* out, int64_t out_ld,
                                  const int64_t* out_row_offsets_host, void* last_hidden, void* workspace, size_t workspace_bytes,
                                  sl_stream stream) {
  SL_CHECK_ARG(m && mel && workspace && n_utt > 0, "sl_whisper_forward: bad arguments");
  SL_CHECK_ARG(m->reserved == 1, "sl_whisper_forward: model struct is not a Whisper front end");
  SL_CHECK_ARG((m->proj_w && out) || (!m->proj_w && last_hidden), "sl_whisper_forward: need `out` (pool) or `last_hidden`");
  SL_CHECK_ARG(m->hidden % m->n_heads == 0 && m->hidden / m->n_heads == 64, "sl_whisper_forward: head_dim must be 64");
  hipStream_t st = (hipStream_t)stream;
  const int dt = m->dtype, H = m->hidden, n_mel = m->conv_dim[0];
  const size_t sz = sl_dtype_size(dt);
  const int64_t T = m->pos_k, F = 2 * T;   // F mel frames in, T positions out
  HubertPlan pl;
  SL_TRY(whisper_plan(m, n_utt, pl));
  HubertWs w;
  const size_t need = hubert_carve(m, pl, n_utt, workspace, workspace_bytes, w);
  SL_CHECK_ARG(need <= workspace_bytes, "sl_whisper_forward: workspace %zu B < required %zu B", workspace_bytes, need);
  {
    std::vector<int32_t> cu(n_utt + 1), kl(n_utt, (int32_t)T);
    for (int u = 0; u <= n_utt; ++u) cu[u] = (int32_t)(u * T);
    SL_HIP(hipMemcpyAsync(w.cu, cu.data(), (n_utt + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    SL_HIP(hipMemcpyAsync(w.cuk, cu.data(), n_utt * sizeof(int32_t), hipMemcpyHostToDevice, st));
    SL_HIP(hipMemcpyAsync(w.klen, kl.data(), n_utt * sizeof(int32_t), hipMemcpyHostToDevice, st));
    std::vector<int64_t> prec;
    proj_records(m, pl, n_utt, out_ld, out_row_offsets_host, prec);
    SL_HIP(hipMemcpyAsync(w.pdesc, prec.data(), prec.size() * sizeof(int64_t), hipMemcpyHostToDevice, st));
    SL_HIP(hipStreamSynchronize(st));
  }
  // haloed inputs: one zero row before and after each utterance's frames (Conv1d padding=1)
  SL_HIP(hipMemsetAsync(w.convA, 0, (size_t)n_utt * (F + 2) * n_mel * sz, st));
  SL_HIP(hipMemsetAsync(w.convB, 0, (size_t)n_utt * (F + 2) * H * sz, st));
  SL_HIP(hipMemcpy2DAsync(bptr(w.convA) + (size_t)n_mel * sz, (size_t)(F + 2) * n_mel * sz, mel, (size_t)F * n_mel * sz, (size_t)F * n_mel * sz, n_utt,
                          hipMemcpyDeviceToDevice, st));
  sl_gemm_args a;
  memset(&a, 0, sizeof(a));   // conv1: out row t reads the 3*n_mel contiguous elements from haloed row t
  a.A = w.convA; a.lda = n_mel; a.strideA = (F + 2) * n_mel;
  a.W = m->conv_w[1]; a.ldw = 3 * n_mel;
  a.C = bptr(w.convB) + (size_t)H * sz; a.ldc = H; a.strideC = (F + 2) * H;
  a.bias = m->conv_b[1];
  a.M = (int)F; a.N = H; a.K = 3 * n_mel; a.batch = n_utt; a.dtype = dt; a.act = SL_ACT_GELU;
  SL_TRY(sl_gemm_impl(&a, nullptr, nullptr, st));
  memset(&a, 0, sizeof(a));   // conv2 (stride 2): out row t reads haloed rows 2t .. 2t+2; + positional table after GELU
  a.A = w.convB; a.lda = 2 * (int64_t)H; a.strideA = (F + 2) * H;
  a.W = m->conv_w[2]; a.ldw = 3 * (int64_t)H;
  a.C = w.x; a.ldc = H; a.strideC = T * H;
  a.bias = m->conv_b[2];
  a.residual = m->pos_w; a.ldr = H; a.strideR = 0;
  a.M = (int)T; a.N = H; a.K = 3 * H; a.batch = n_utt; a.dtype = dt; a.act = SL_ACT_GELU;
  SL_TRY(sl_gemm_impl(&a, nullptr, nullptr, st));
  return encoder_tail(m, w, pl, n_utt, out, out_ld, out_row_offsets_host, last_hidden, st);
}

// ================================================================================================
// Llama
// ================================================================================================
// SL_MAX_DECODE_BATCH (speechllm.h): rows of one decode step (M of the weight-streaming GEMMs)

struct LlamaWs {
  void *h, *qkv, *att, *mid, *last, *part, *split;
  size_t split_bytes;
  float *rstd_a, *rstd_b;   // RMSNorm scales handed from the o / down projection's reduce pass to the next fused GEMM
  int32_t *tok_seq, *tok_pos, *cu, *cuk, *klen;
};

static size_t llama_carve(const sl_llama_model* m, int64_t n_tok, int nseq, void* base, size_t cap, LlamaWs& w) {
  const size_t sz = sl_dtype_size(m->dtype);
  const int qkv_w = (m->n_heads + 2 * m->n_kv_heads) * m->head_dim;
  Carver c(base, cap);
  w.h = c.take(n_tok * m->hidden * sz);
  w.qkv = c.take(n_tok * qkv_w * sz);
  w.att = c.take(n_tok * (int64_t)m->n_heads * m->head_dim * sz);
  w.mid = c.take(n_tok * (int64_t)m->ffn * sz);
  w.last = c.take((size_t)nseq * m->hidden * sz);
  w.part = c.take(sl_attn_decode_split_ws(nseq, m->n_heads, m->n_kv_heads, m->rope_len));  // rope_len >= max_ctx
  {  // K-split partial sums of the decode GEMMs (batches above the skinny kernel's range)
    size_t sb = 0;
    const int shapes[5][2] = {{qkv_w, m->hidden}, {m->hidden, m->n_heads * m->head_dim}, {2 * m->ffn, m->hidden}, {m->hidden, m->ffn}, {m->vocab, m->hidden}};
    for (auto& sh : shapes) {
      const size_t b = sl_gemm_split_workspace_bytes(nseq, sh[0], sh[1], m->dtype);
      if (b > sb) sb = b;
    }
    w.split_bytes = sb;
    w.split = c.take(sb);
  }
  w.rstd_a = (float*)c.take((size_t)nseq * sizeof(float));
  w.rstd_b = (float*)c.take((size_t)nseq * sizeof(float));
  w.tok_seq = (int32_t*)c.take(n_tok * sizeof(int32_t));
  w.tok_pos = (int32_t*)c.take(n_tok * sizeof(int32_t));
  w.cu = (int32_t*)c.take((nseq + 2) * sizeof(int32_t));     // + 1: the shared prompt prefix is one more attention sequence in prefill
  w.cuk = (int32_t*)c.take((nseq + 1) * sizeof(int32_t));
  w.klen = (int32_t*)c.take((nseq + 1) * sizeof(int32_t));
  return c.off + 256;
}

extern "C" size_t sl_llama_workspace_bytes(const sl_llama_model* m, int64_t n_tok, int32_t nseq) {
  LlamaWs w;
  return llama_carve(m, n_tok, nseq, nullptr, 0, w);
}

static int llama_check(const sl_llama_model* m, const sl_kv_cache* kv) {
  SL_CHECK_ARG(m && kv && m->layers && m->embed && m->lm_head && m->final_norm && m->rope_cos && m->rope_sin, "llama: null model field");
  SL_CHECK_ARG(m->head_dim == 128, "llama: head_dim %d not built (128)", m->head_dim);
  SL_CHECK_ARG(kv->k_cache && kv->v_cache && kv->slots > 0 && kv->max_ctx > 0, "llama: bad kv cache");
  SL_CHECK_ARG(kv->shared_prefix >= 0 && kv->shared_prefix <= kv->max_ctx, "llama: kv cache shared_prefix %d outside [0, max_ctx=%d]", kv->shared_prefix, kv->max_ctx);
  SL_CHECK_ARG(kv->max_ctx <= m->rope_len, "llama: max_ctx %d exceeds the rope table (%d)", kv->max_ctx, m->rope_len);
  return 0;
}

static inline size_t kv_layer_bytes(const sl_llama_model* m, const sl_kv_cache* kv) {
  return (size_t)kv->slots * m->n_kv_heads * kv->max_ctx * m->head_dim * sl_dtype_size(m->dtype);
}

// ---- shared prompt prefix in prefill (sl_kv_cache.shared_prefix = P): the P rows every sequence opens with are computed once ----
// rows of the packed prompt buffer gathered by an index list (16-byte pieces; dst and src do not overlap)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, const int32_t* __restrict__ map,
                                                          int64_t n_rows, int row_vec) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_rows * row_vec) return;
  const int64_t r = i / row_vec;
  const int c = (int)(i - r * row_vec);
  dst[i] = src[(int64_t)map[r] * row_vec + c];
}
// K / V rows [0, P) of slot 0 copied to slots 1 .. nseq-1 of one layer's cache: grid (P, n_kv, nseq - 1); lanes 0..31 move the K row,
// 32..63 the V row, 16 bytes each (rows of 256 B in bf16, 512 B in fp32)
__global__ __launch_bounds__(64) void kv_prefix_broadcast_kernel(uint4* __restrict__ kc, uint4* __restrict__ vc, int nkv, int max_ctx, int row_vec) {
  const int pos = blockIdx.x, h = blockIdx.y, s = blockIdx.z + 1;
  uint4* c = threadIdx.x < 32 ? kc : vc;
  const int lane = threadIdx.x & 31;
  if (lane >= row_vec) return;
  const int64_t src = ((int64_t)h * max_ctx + pos) * row_vec + lane;
  c[((int64_t)s * nkv * max_ctx) * row_vec + src] = c[src];
}

// ---- weight prefetch beside the decode chain (small batches) ----
// At batch 1 a decode step is a chain of ~140 dependent launches that each stream their weights once: HBM idles through every ramp, tail and
// launch gap (r04: lm_head reads at 6.6 TB/s, qkv at 3.0, o at 3.4; 0.53 of the HBM ceiling end to end).  The captured graph therefore carries
// a second branch: while the consumer of weight matrix k runs, a prefetch kernel streams matrix k + 1 through the memory side into the
// 256 MiB Infinity Cache (plain loads allocate there; guide: a table stays resident while table + traffic between uses < 256 MiB, and is
// then read at ~8.6 TB/s against ~6.1 from HBM).  The branch is a hint: it writes nothing, nothing waits for it except the end of the graph,
// and the lead is two matrices (<= 150 MB at Llama-3.2-3B).
// MEASURED (profiles/r06_j_decode_prefetch_ab.txt): the step takes 3.3 ms with the branch against 1.50 ms without, at every batch 1..26, with
// non-temporal and with default-policy weight loads alike — the second branch of the graph does not run beside the chain for free (its long-lived
// blocks and the chain's short launches share one dispatcher, and the bytes are fetched twice).  Kept behind SL_DECODE_PREFETCH=1 (default 0) as the
// record of the experiment.
__global__ __launch_bounds__(256) void weight_prefetch_kernel(const uint4* __restrict__ w, size_t n16, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {       // eight 16-byte loads in flight per lane
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = w[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < n16; i += stride) { const uint4 v = w[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x9e3779b9u && sink) *sink = acc;          // (practically never: keeps the loads alive)
}

struct DecodePrefetch {
  hipStream_t main = nullptr, side = nullptr;
  hipEvent_t ev = nullptr;
  uint32_t* sink = nullptr;
  bool on = false;
  // called right before the consumer of the CURRENT matrix is launched on `main`: the prefetch of `next` starts when everything launched so far has
  // completed, i.e. together with that consumer
  int ahead(const void* next, size_t bytes) {
    if (!on || !next || bytes < (1u << 20)) return 0;
    SL_HIP(hipEventRecord(ev, main));
    SL_HIP(hipStreamWaitEvent(side, ev, 0));
    const size_t n16 = bytes / 16;
    int blocks = (int)((n16 + 256 * 8 - 1) / (256 * 8));
    blocks = blocks > 512 ? 512 : (blocks < 1 ? 1 : blocks);
    hipLaunchKernelGGL(weight_prefetch_kernel, dim3(blocks), dim3(256), 0, side, (const uint4*)next, n16, sink);
    SL_CHECK_LAUNCH("weight_prefetch");
    return 0;
  }
  int join() {
    if (!on) return 0;
    SL_HIP(hipEventRecord(ev, side));
    SL_HIP(hipStreamWaitEvent(main, ev, 0));
    return 0;
  }
};
static thread_local DecodePrefetch* g_prefetch = nullptr;      // set by sl_generate around the capture of a small-batch decode graph

// one decoder layer over `n` token rows; attention chosen by `decode`
// decode GEMM on the fragment-packed weights, optionally absorbing the preceding RMSNorm / RoPE+KV-append
static int dec_gemm(const sl_llama_model* m, const LlamaWs& w, const void* A, int64_t lda, const void* Wp, void* C, int64_t ldc, const void* res,
                    int M, int N, int K, int act, int out_f32, const sl_gemm_fused* fx_in, hipStream_t st, const float* rstd_in = nullptr,
                    float* rstd_out = nullptr, void* norm_out = nullptr, const void* norm_gain = nullptr) {
  sl_gemm_fused fxl;
  if (fx_in) fxl = *fx_in; else memset(&fxl, 0, sizeof(fxl));
  fxl.split_ws = w.split; fxl.split_ws_bytes = w.split_bytes;
  fxl.rstd_in = rstd_in; fxl.rstd_out = rstd_out; fxl.rms_eps = m->rms_eps;
  fxl.norm_out = norm_out; fxl.norm_gain = norm_gain;
  const sl_gemm_fused* fx = &fxl;
  sl_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.W = Wp; a.ldw = K; a.C = C; a.ldc = ldc; a.residual = res; a.ldr = ldc;
  a.M = M; a.N = N; a.K = K; a.batch = 1; a.dtype = m->dtype; a.act = act; a.out_f32 = out_f32; a.w_layout = SL_W_PACKED;
  return sl_gemm_impl(&a, fx, nullptr, st);
}

// one decoder layer over `n` token rows; attention chosen by `decode`
// rstd_chain (decode): the o / down projections run K-split, so their reduce passes emit the RMSNorm scale of the rows
// they store and the next fused GEMM (gate/up, next layer's qkv, lm_head) takes it instead of recomputing it per block;
// rstd_qkv = scale of x on entry (NULL: this layer's qkv takes its own statistics)
// prefix_bcast = P > 0 (prefill with a shared prompt prefix): the P prefix rows were appended to slot 0 only and are copied to the other
// bcast_slots - 1 slots before attention, in which the prefix is one more sequence (nseq counts it)
static int llama_layer(const sl_llama_model* m, const sl_kv_cache* kv, int l, void* x, int64_t n, LlamaWs& w, bool decode, int nseq,
                       int max_qlen, const int32_t* ctx_len_dev, hipStream_t st, bool rstd_chain = false, const float* rstd_qkv = nullptr,
                       int prefix_bcast = 0, int bcast_slots = 0, bool rstd_pass = false) {
  const sl_llama_layer& L = m->layers[l];
  const int dt = m->dtype, H = m->hidden, D = m->head_dim, nh = m->n_heads, nkv = m->n_kv_heads;
  const int qkv_w = (nh + 2 * nkv) * D;
  void* kc = bptr(kv->k_cache) + (size_t)l * kv_layer_bytes(m, kv);
  void* vc = bptr(kv->v_cache) + (size_t)l * kv_layer_bytes(m, kv);
  const float scale = 1.0f / sqrtf((float)D);
  if (decode && L.wqkv_dec && L.wo_dec && L.wgu_dec && L.wdown_dec) {
    // 5-6 launches per layer: [norm+]qkv+rope+append | attention (split + merge) | o+res | [norm+]gate/up+silu.mul | down+res
    sl_gemm_fused fx;
    memset(&fx, 0, sizeof(fx));
    fx.fuse_rms = m->dec_fused_norm; fx.rms_eps = m->rms_eps;
    fx.rope_cos = m->rope_cos; fx.rope_sin = m->rope_sin; fx.tok_pos = ctx_len_dev; fx.tok_seq = w.tok_seq;
    fx.k_cache = kc; fx.v_cache = vc; fx.n_heads = nh; fx.n_kv_heads = nkv; fx.max_ctx = kv->max_ctx;
    // small-batch graphs: the prefetch branch runs two matrices ahead of the chain (DecodePrefetch): released where a consumer is launched
    DecodePrefetch* pf = g_prefetch;
    const size_t esz = sl_dtype_size(dt);
    const size_t b_qkv = (size_t)qkv_w * H * esz, b_o = (size_t)H * nh * D * esz, b_gu = (size_t)2 * m->ffn * H * esz, b_down = (size_t)H * m->ffn * esz;
    const sl_llama_layer* Ln = (l + 1 < m->n_layers) ? &m->layers[l + 1] : nullptr;
    if (pf && l == 0) { SL_TRY(pf->ahead(L.wqkv_dec, b_qkv)); SL_TRY(pf->ahead(L.wo_dec, b_o)); }
    if (pf) SL_TRY(pf->ahead(L.wgu_dec, b_gu));                       // ... while qkv, attention and o run
    const void* a_in = x;
    if (!m->dec_fused_norm) { SL_TRY(sl_rmsnorm(x, w.h, L.norm1, n, H, m->rms_eps, dt, (sl_stream)st)); a_in = w.h; }
    // rstd_pass (rows whose o / down projections run unsplit: no reduce pass forms the RMSNorm scales): one read of x leaves them, so
    // that the gain-folded products stay on the 256 x 128 streaming blocks instead of taking the statistics per block themselves
    if (rstd_pass) { SL_TRY(sl_rmsnorm_rstd_impl(x, nullptr, nullptr, w.rstd_b, n, H, m->rms_eps, dt, st)); rstd_qkv = w.rstd_b; }
    SL_TRY(dec_gemm(m, w, a_in, H, L.wqkv_dec, w.qkv, (int64_t)nh * D, nullptr, (int)n, qkv_w, H, SL_ACT_ROPE_KV, 0, &fx, st, rstd_qkv));
    SL_TRY(sl_attn_decode_split_impl(w.qkv, (int64_t)nh * D, kc, vc, w.att, w.part, ctx_len_dev, 1, (int)n, nh, nkv, D, kv->max_ctx, scale, dt, st, 1, kv->shared_prefix));
    // Above ~900 rows gate/up runs on the row-major 256 x 256 tiles (the prefill kernel): at 1 024 rows it is 4 x 64 = 256 tiles, one per
    // CU, 78-81 us against 99 us on the 256 x 128 streaming block (tools/time_decode_tiled.py; in the graph: profiles/r04_l_*).  Its input
    // must then be normalised: the o projection's reduce pass, which already forms each row's RMSNorm scale, writes the normalised rows
    // beside x (sl_gemm_fused.norm_out) — a separate sl_rmsnorm launch costs 11.5 us per layer in the graph and ate the gain, and o itself
    // stays on the streaming form (27 + 11 us against 47 us on the 128 x 128 tiles its 48 big tiles fall back to).  SL_DECODE_TILED=0: off.
    if (pf) SL_TRY(pf->ahead(L.wdown_dec, b_down));                   // ... while o and gate/up run
    if (rstd_pass) {
      SL_TRY(dec_gemm(m, w, w.att, (int64_t)nh * D, L.wo_dec, x, H, x, (int)n, H, nh * D, SL_ACT_NONE, 0, nullptr, st));
      if (L.wgu && sl_env().decode_tiled) {     // gate/up on the row-major 256 x 256 tiles (2 048 rows: 512 tiles = two whole rounds of the chip)
        SL_TRY(sl_rmsnorm(x, w.h, L.norm2, n, H, m->rms_eps, dt, (sl_stream)st));
        SL_TRY(gemm(dt, w.h, H, L.wgu, H, w.mid, m->ffn, nullptr, nullptr, 0, (int)n, 2 * m->ffn, H, SL_ACT_SILU_MUL, 0, st));
      } else {
        SL_TRY(sl_rmsnorm_rstd_impl(x, nullptr, nullptr, w.rstd_a, n, H, m->rms_eps, dt, st));
        sl_gemm_fused fn;
        memset(&fn, 0, sizeof(fn));
        fn.fuse_rms = 1; fn.rms_eps = m->rms_eps;
        SL_TRY(dec_gemm(m, w, x, H, L.wgu_dec, w.mid, m->ffn, nullptr, (int)n, 2 * m->ffn, H, SL_ACT_SILU_MUL, 0, &fn, st, w.rstd_a));
      }
    } else if (sl_family_rows((int)n) > 896 && dt == SL_BF16 && L.wgu && rstd_chain && sl_env().decode_tiled) {
      SL_TRY(dec_gemm(m, w, w.att, (int64_t)nh * D, L.wo_dec, x, H, x, (int)n, H, nh * D, SL_ACT_NONE, 0, nullptr, st, nullptr, w.rstd_a, w.h, L.norm2));
      SL_TRY(gemm(dt, w.h, H, L.wgu, H, w.mid, m->ffn, nullptr, nullptr, 0, (int)n, 2 * m->ffn, H, SL_ACT_SILU_MUL, 0, st));
    } else {
      SL_TRY(dec_gemm(m, w, w.att, (int64_t)nh * D, L.wo_dec, x, H, x, (int)n, H, nh * D, SL_ACT_NONE, 0, nullptr, st, nullptr,
                      rstd_chain ? w.rstd_a : nullptr));
      sl_gemm_fused fn;
      memset(&fn, 0, sizeof(fn));
      fn.fuse_rms = m->dec_fused_norm; fn.rms_eps = m->rms_eps;
      a_in = x;
      if (pf && Ln) SL_TRY(pf->ahead(Ln->wqkv_dec, b_qkv));           // ... while gate/up and down run
      if (!m->dec_fused_norm) { SL_TRY(sl_rmsnorm(x, w.h, L.norm2, n, H, m->rms_eps, dt, (sl_stream)st)); a_in = w.h; }
      SL_TRY(dec_gemm(m, w, a_in, H, L.wgu_dec, w.mid, m->ffn, nullptr, (int)n, 2 * m->ffn, H, SL_ACT_SILU_MUL, 0, &fn, st,
                      rstd_chain ? w.rstd_a : nullptr));
    }
    if (pf && Ln) SL_TRY(pf->ahead(Ln->wo_dec, b_o));
    SL_TRY(dec_gemm(m, w, w.mid, m->ffn, L.wdown_dec, x, H, x, (int)n, H, m->ffn, SL_ACT_NONE, 0, nullptr, st, nullptr,
                    rstd_chain ? w.rstd_b : nullptr));
    return 0;
  }
  SL_TRY(sl_rmsnorm(x, w.h, L.norm1, n, H, m->rms_eps, dt, (sl_stream)st));
  SL_TRY(gemm(dt, w.h, H, L.wqkv, H, w.qkv, qkv_w, nullptr, nullptr, 0, (int)n, qkv_w, H, SL_ACT_NONE, 0, st));
  SL_TRY(sl_rope_kv_append(w.qkv, kc, vc, w.tok_seq, decode ? ctx_len_dev : w.tok_pos, m->rope_cos, m->rope_sin, n, nh, nkv, D, kv->max_ctx,
                           dt, (sl_stream)st));
  if (decode) {
    SL_TRY(sl_attn_decode_split_impl(w.qkv, qkv_w, kc, vc, w.att, w.part, ctx_len_dev, 1, (int)n, nh, nkv, D, kv->max_ctx, scale, dt, st, 1, kv->shared_prefix));
  } else {
    if (prefix_bcast > 0 && bcast_slots > 1) {
      const int row_vec = (int)(D * sl_dtype_size(dt) / 16);
      SL_CHECK_ARG(row_vec <= 32 && (D * sl_dtype_size(dt)) % 16 == 0, "llama: shared prefix broadcast needs K / V rows of <= 512 bytes");
      hipLaunchKernelGGL(kv_prefix_broadcast_kernel, dim3(prefix_bcast, nkv, bcast_slots - 1), dim3(64), 0, st, (uint4*)kc, (uint4*)vc, nkv, kv->max_ctx, row_vec);
      SL_CHECK_LAUNCH("kv_prefix_broadcast");
    }
    sl_attn_args a;
    memset(&a, 0, sizeof(a));
    a.q = w.qkv; a.q_row_stride = qkv_w; a.q_head_stride = D;
    a.k = kc; a.k_row_stride = D; a.k_head_stride = (int64_t)kv->max_ctx * D;
    a.v = vc; a.v_row_stride = D; a.v_head_stride = (int64_t)kv->max_ctx * D;
    a.out = w.att; a.o_row_stride = (int64_t)nh * D; a.o_head_stride = D;
    a.cu_q = w.cu; a.cu_k = w.cuk; a.klen = w.klen;
    a.nseq = nseq; a.max_qlen = max_qlen; a.n_heads = nh; a.n_kv_heads = nkv; a.head_dim = D; a.causal = 1; a.dtype = dt; a.scale = scale;
    SL_TRY(sl_attn_fwd(&a, (sl_stream)st));
  }
  SL_TRY(gemm(dt, w.att, (int64_t)nh * D, L.wo, (int64_t)nh * D, x, H, nullptr, x, H, (int)n, H, nh * D, SL_ACT_NONE, 0, st));
  SL_TRY(sl_rmsnorm(x, w.h, L.norm2, n, H, m->rms_eps, dt, (sl_stream)st));
  SL_TRY(gemm(dt, w.h, H, L.wgu, H, w.mid, m->ffn, nullptr, nullptr, 0, (int)n, 2 * m->ffn, H, SL_ACT_SILU_MUL, 0, st));
  SL_TRY(gemm(dt, w.mid, m->ffn, L.wdown, m->ffn, x, H, nullptr, x, H, (int)n, H, m->ffn, SL_ACT_NONE, 0, st));
  return 0;
}

extern "C" int sl_llama_prefill(const sl_llama_model* m, const sl_kv_cache* kv, void* x, const int32_t* cu_seqlens_host, int32_t nseq,
                                float* logits, int32_t* ctx_len_dev, void* hidden_taps, void* workspace, size_t workspace_bytes,
                                sl_stream stream) {
  SL_TRY(llama_check(m, kv));
  SL_CHECK_ARG(x && cu_seqlens_host && logits && ctx_len_dev && workspace && nseq > 0 && nseq <= kv->slots, "sl_llama_prefill: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n_tok = cu_seqlens_host[nseq];
  const int dt = m->dtype, H = m->hidden;
  const size_t sz = sl_dtype_size(dt);
  LlamaWs w;
  const size_t need = llama_carve(m, n_tok, nseq, workspace, workspace_bytes, w);
  SL_CHECK_ARG(need <= workspace_bytes, "sl_llama_prefill: workspace %zu B < required %zu B", workspace_bytes, need);
  std::vector<int32_t> tseq(n_tok), tpos(n_tok), cuk(nseq + 1), kl(nseq + 1), ctx(nseq), cuq(nseq + 2), last_row(nseq);
  int max_q = 0;
  // Shared prompt prefix (sl_kv_cache.shared_prefix = P, the caller's promise that rows [0, P) of every sequence are the same rows): they
  // are computed ONCE — the batch becomes [P prefix rows | tail of sequence 0 | tail of sequence 1 | ...], the prefix's K / V rows are
  // appended to slot 0 and copied to the other slots in every layer, and attention sees the prefix as one more sequence while a tail's
  // queries attend [prefix + tail] keys of their own slot (causal with klen > qlen).  Every row-wise product, RoPE at the row's own
  // position and attention over the same keys in the same order give the bits the unshared pass gives (asserted on the cache and the
  // logits).  Not with hidden_taps (they are per row of the caller's layout) or when a sequence is nothing but the prefix.
  int P = (kv->shared_prefix > 0 && nseq > 1 && !hidden_taps && sl_env().prefill_share_prefix) ? kv->shared_prefix : 0;
  for (int s = 0; s < nseq; ++s) {
    const int len = cu_seqlens_host[s + 1] - cu_seqlens_host[s];
    SL_CHECK_ARG(len > 0 && len <= kv->max_ctx, "sl_llama_prefill: sequence %d length %d outside (0, max_ctx=%d]", s, len, kv->max_ctx);
    SL_CHECK_ARG(kv->shared_prefix <= len, "sl_llama_prefill: kv cache shared_prefix %d exceeds sequence %d (%d tokens)", kv->shared_prefix, s, len);
    if (len <= P) P = 0;
  }
  int64_t n_run = n_tok;       // rows the layers run over
  int nseq_attn = nseq;
  if (P > 0) {
    std::vector<int32_t> map;
    map.reserve(n_tok);
    for (int t = 0; t < P; ++t) { map.push_back(cu_seqlens_host[0] + t); }
    cuq[0] = 0; cuq[1] = P; cuk[0] = 0; kl[0] = P;
    for (int t = 0; t < P; ++t) { tseq[t] = 0; tpos[t] = t; }
    max_q = P;
    for (int s = 0; s < nseq; ++s) {
      const int len = cu_seqlens_host[s + 1] - cu_seqlens_host[s], tail = len - P;
      const int r0 = (int)map.size();
      for (int t = 0; t < tail; ++t) { map.push_back(cu_seqlens_host[s] + P + t); tseq[r0 + t] = s; tpos[r0 + t] = P + t; }
      cuq[s + 2] = r0 + tail;
      cuk[s + 1] = s * m->n_kv_heads * kv->max_ctx;
      kl[s + 1] = len; ctx[s] = len;
      last_row[s] = r0 + tail - 1;
      if (tail > max_q) max_q = tail;
    }
    n_run = (int64_t)map.size();
    nseq_attn = nseq + 1;
    // x -> compact rows: gathered into the (still unused) FFN scratch, then copied back to the head of x
    const int row_vec = (int)((size_t)H * sz / 16);
    SL_CHECK_ARG(((size_t)H * sz) % 16 == 0, "sl_llama_prefill: hidden rows must be a multiple of 16 bytes");
    SL_HIP(hipMemcpyAsync(w.tok_pos, map.data(), n_run * sizeof(int32_t), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((n_run * row_vec + 255) / 256)), dim3(256), 0, st, (const uint4*)x, (uint4*)w.mid, w.tok_pos, n_run, row_vec);
    SL_CHECK_LAUNCH("gather_rows");
    SL_HIP(hipMemcpyAsync(x, w.mid, (size_t)n_run * H * sz, hipMemcpyDeviceToDevice, st));
    SL_HIP(hipStreamSynchronize(st));      // `map` leaves scope / tok_pos is rewritten below
  } else {
    for (int s = 0; s < nseq; ++s) {
      const int len = cu_seqlens_host[s + 1] - cu_seqlens_host[s];
      for (int t = 0; t < len; ++t) { tseq[cu_seqlens_host[s] + t] = s; tpos[cu_seqlens_host[s] + t] = t; }
      cuk[s] = s * m->n_kv_heads * kv->max_ctx;  // first cache row of the sequence (rows of D elements, head-major inside)
      kl[s] = len; ctx[s] = len;
      cuq[s] = cu_seqlens_host[s];
      last_row[s] = cu_seqlens_host[s + 1] - 1;
      if (len > max_q) max_q = len;
    }
    cuq[nseq] = cu_seqlens_host[nseq];
  }
  SL_HIP(hipMemcpyAsync(w.tok_seq, tseq.data(), n_run * sizeof(int32_t), hipMemcpyHostToDevice, st));
  SL_HIP(hipMemcpyAsync(w.tok_pos, tpos.data(), n_run * sizeof(int32_t), hipMemcpyHostToDevice, st));
  SL_HIP(hipMemcpyAsync(w.cu, cuq.data(), (nseq_attn + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
  SL_HIP(hipMemcpyAsync(w.cuk, cuk.data(), nseq_attn * sizeof(int32_t), hipMemcpyHostToDevice, st));
  SL_HIP(hipMemcpyAsync(w.klen, kl.data(), nseq_attn * sizeof(int32_t), hipMemcpyHostToDevice, st));
  SL_HIP(hipMemcpyAsync(ctx_len_dev, ctx.data(), nseq * sizeof(int32_t), hipMemcpyHostToDevice, st));
  SL_HIP(hipStreamSynchronize(st));
  for (int l = 0; l < m->n_layers; ++l) {
    if (hidden_taps) SL_HIP(hipMemcpyAsync(bptr(hidden_taps) + (size_t)l * n_tok * H * sz, x, n_tok * H * sz, hipMemcpyDeviceToDevice, st));
    SL_TRY(llama_layer(m, kv, l, x, n_run, w, false, nseq_attn, max_q, nullptr, st, false, nullptr, P, nseq));
  }
  if (hidden_taps) {
    SL_TRY(sl_rmsnorm(x, bptr(hidden_taps) + (size_t)m->n_layers * n_tok * H * sz, m->final_norm, n_tok, H, m->rms_eps, dt, stream));
  }
  // last-token rows -> final norm -> lm_head (fp32 logits)
  for (int s = 0; s < nseq; ++s)
    SL_HIP(hipMemcpyAsync(bptr(w.last) + (size_t)s * H * sz, bptr(x) + (size_t)last_row[s] * H * sz, H * sz, hipMemcpyDeviceToDevice, st));
  SL_TRY(sl_rmsnorm(w.last, w.last, m->final_norm, nseq, H, m->rms_eps, dt, stream));
  SL_TRY(gemm(dt, w.last, H, m->lm_head, H, logits, m->vocab, nullptr, nullptr, 0, nseq, m->vocab, H, SL_ACT_NONE, 1, st));
  return 0;
}

// decode-step state carved from the tail of the workspace by sl_greedy_generate, or supplied by the caller
// rows from which the lm_head of a greedy decode step leaves per-64-column partial maxima instead of fp32 logits (the tiled
// kernels' fused top-1): the logits round trip is B x vocab x 8 bytes per step, 1 GB at 1 024 rows
constexpr int SL_FUSED_ARGMAX_MIN_B = 256;

// = the steps whose lm_head runs on the tiled kernels (below: packed weights stream through dec_gemm up to 255 rows)
static bool decode_fuses_argmax(const sl_llama_model* m, int B) { B = sl_family_rows(B); return B > 64 && !(m->lm_head_dec && B < SL_FUSED_ARGMAX_MIN_B); }

// fused_top1: the caller will run sl_greedy_select_partial_impl over `logits` reinterpreted as [n_groups][B] floats followed by
// [n_groups][B] int32 (n_groups = ceil(vocab / 64)) — only honoured where decode_fuses_argmax() says so
static int decode_step(const sl_llama_model* m, const sl_kv_cache* kv, const int32_t* next_ids, const int32_t* ctx_len, int B, float* logits,
                       void* x, LlamaWs& w, hipStream_t st, bool fused_top1 = false) {
  const int dt = m->dtype, H = m->hidden;
  SL_TRY(sl_embed_gather(m->embed, next_ids, x, B, H, dt, (sl_stream)st));
  // hand RMSNorm scales along the chain when every o / down projection has a reduce pass to compute them in
  bool chain = m->dec_fused_norm && m->lm_head_dec && H / 16 * 4 <= 1024 && H % 16 == 0 &&
               sl_gemm_split_count(B, H, m->n_heads * m->head_dim, dt) > 1 && sl_gemm_split_count(B, H, m->ffn, dt) > 1 && w.split != nullptr;
  for (int l = 0; l < m->n_layers && chain; ++l) {
    const sl_llama_layer& L = m->layers[l];
    chain = L.wqkv_dec && L.wo_dec && L.wgu_dec && L.wdown_dec;
  }
  // rows whose o / down projections run UNSPLIT on the 256 x 128 blocks (from ~1 500 rows at Llama-3.2-3B's widths: 8 x 24 = 192 blocks
  // at 2 048): no reduce pass exists to take the RMSNorm scales in, a one-read pass in front of qkv and gate/up leaves them instead
  bool rstd_pass = !chain && dt == SL_BF16 && m->dec_fused_norm && sl_family_rows(B) > 384 && sl_gemm_split_count(B, H, m->n_heads * m->head_dim, dt) == 1 &&
                   sl_gemm_split_count(B, H, m->ffn, dt) == 1;
  for (int l = 0; l < m->n_layers && rstd_pass; ++l) {
    const sl_llama_layer& L = m->layers[l];
    rstd_pass = L.wqkv_dec && L.wo_dec && L.wgu_dec && L.wdown_dec;
  }
  for (int l = 0; l < m->n_layers; ++l)
    SL_TRY(llama_layer(m, kv, l, x, B, w, true, B, 1, ctx_len, st, chain, (chain && l > 0) ? w.rstd_b : nullptr, 0, 0, rstd_pass));
  // lm_head: above ~256 rows the 128-tile MFMA kernel on the row-major matrix beats the streaming kernel on the packed one
  // (M=512: 439 vs 632 us; the 263 MB of fp32 logits dominate either way)
  if (m->lm_head_dec && sl_family_rows(B) < SL_FUSED_ARGMAX_MIN_B) {
    sl_gemm_fused fx;
    memset(&fx, 0, sizeof(fx));
    fx.fuse_rms = m->dec_fused_norm; fx.rms_eps = m->rms_eps;
    const void* a_in = x;
    if (!m->dec_fused_norm) { SL_TRY(sl_rmsnorm(x, w.last, m->final_norm, B, H, m->rms_eps, dt, (sl_stream)st)); a_in = w.last; }
    return dec_gemm(m, w, a_in, H, m->lm_head_dec, logits, m->vocab, nullptr, B, m->vocab, H, SL_ACT_NONE, 1, &fx, st,
                    (chain && m->n_layers > 0) ? w.rstd_b : nullptr);
  }
  SL_TRY(sl_rmsnorm(x, w.last, m->final_norm, B, H, m->rms_eps, dt, (sl_stream)st));
  if (fused_top1 && decode_fuses_argmax(m, B)) {
    const int64_t ng = (m->vocab + 63) / 64;
    sl_gemm_args a;
    memset(&a, 0, sizeof(a));
    a.A = w.last; a.lda = H; a.W = m->lm_head; a.ldw = H; a.C = nullptr; a.ldc = m->vocab;
    a.M = B; a.N = m->vocab; a.K = H; a.batch = 1; a.dtype = dt; a.act = SL_ACT_NONE; a.out_f32 = 1; a.w_layout = SL_W_ROWMAJOR;
    sl_gemm_ex_args ex;
    memset(&ex, 0, sizeof(ex));
    ex.w_mod = 1;
    ex.amax_val = logits;
    ex.amax_idx = (int32_t*)(logits + ng * B);
    return sl_gemm_impl(&a, nullptr, &ex, st);
  }
  SL_TRY(gemm(dt, w.last, H, m->lm_head, H, logits, m->vocab, nullptr, nullptr, 0, B, m->vocab, H, SL_ACT_NONE, 1, st));
  return 0;
}

__global__ void iota_kernel(int32_t* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = i;
}

extern "C" int sl_llama_decode_step(const sl_llama_model* m, const sl_kv_cache* kv, const int32_t* next_ids_dev, const int32_t* ctx_len_dev,
                                    int32_t B, float* logits, void* workspace, size_t workspace_bytes, sl_stream stream) {
  SL_TRY(llama_check(m, kv));
  SL_CHECK_ARG(next_ids_dev && ctx_len_dev && logits && workspace && B > 0 && B <= kv->slots && B <= SL_MAX_DECODE_BATCH, "sl_llama_decode_step: bad arguments (B<=%d)", SL_MAX_DECODE_BATCH);
  hipStream_t st = (hipStream_t)stream;
  LlamaWs w;
  Carver c(workspace, workspace_bytes);
  void* x = c.take((size_t)B * m->hidden * sl_dtype_size(m->dtype));
  c.take(0);
  const size_t off = c.off;
  SL_CHECK_ARG(off <= workspace_bytes, "sl_llama_decode_step: workspace too small");
  const size_t need = off + llama_carve(m, B, B, bptr(workspace) + off, workspace_bytes - off, w);
  SL_CHECK_ARG(need <= workspace_bytes, "sl_llama_decode_step: workspace %zu B < required %zu B", workspace_bytes, need);
  if (w.split && w.split_bytes >= 8192) SL_HIP(hipMemsetAsync(w.split, 0, 8192, st));   // the K-split fix-up's counters start at zero
  SL_TRY(sl_attn_decode_split_zero_counters(w.part, B, m->n_heads, m->n_kv_heads, kv->max_ctx, st));   // ... and the split attention's arrival counters
  hipLaunchKernelGGL(iota_kernel, dim3((B + 255) / 256), dim3(256), 0, st, w.tok_seq, B);
  SL_CHECK_LAUNCH("iota");
  return decode_step(m, kv, next_ids_dev, ctx_len_dev, B, logits, x, w, st);
}

// ---- instantiated decode graphs, cached per thread (see sl_generate)
struct DecodeGraphKey {
  const void *model, *layers, *w0, *lm, *embed, *kc, *vc, *ws;
  size_t ws_bytes;
  uint64_t content;        // FNV-1a over the model struct and every layer struct: all weight / norm / rope pointers and dimensions
  int device;
  int B, B0, max_new, use_eos, n_eos, pad, max_ctx, slots, shared_prefix, dtype, n_layers, vocab, fused, limits;
  int eos[8];
  int sample, top_k;
  float temperature, top_p;
  uint64_t seed;
};
struct DecodeGraphEntry { DecodeGraphKey key; hipGraph_t graph; hipGraphExec_t exec; uint64_t stamp; };
static thread_local std::vector<DecodeGraphEntry> g_graphs;
static thread_local uint64_t g_graph_clock = 0;
constexpr size_t SL_GRAPH_CACHE = 64;      // a generation that compacts its batch walks down a ladder of row counts: one graph per rung

static uint64_t fnv1a(uint64_t h, const void* p, size_t n) {
  const unsigned char* b = (const unsigned char*)p;
  for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

// Everything a captured decode step bakes in that the caller could change behind the same host addresses: the model struct's
// fields (dimensions, eps, embed / norm / rope / lm_head pointers) and EVERY layer's pointers.  A caller that rebuilds its
// weights and gets the same struct addresses back from malloc therefore misses the cache instead of replaying stale pointers.
static uint64_t model_content_hash(const sl_llama_model* m) {
  uint64_t h = fnv1a(1469598103934665603ull, m, sizeof(*m));
  if (m->layers && m->n_layers > 0) h = fnv1a(h, m->layers, sizeof(m->layers[0]) * (size_t)m->n_layers);
  return h;
}

extern "C" int sl_decode_graph_cache_clear(void) {
  for (auto& e : g_graphs) { (void)hipGraphExecDestroy(e.exec); (void)hipGraphDestroy(e.graph); }
  const int n = (int)g_graphs.size();
  g_graphs.clear();
  return n;
}

static hipGraphExec_t decode_graph_lookup(const DecodeGraphKey& k) {
  for (auto& e : g_graphs)
    if (memcmp(&e.key, &k, sizeof(k)) == 0) { e.stamp = ++g_graph_clock; return e.exec; }
  return nullptr;
}

static void decode_graph_store(const DecodeGraphKey& k, hipGraph_t g, hipGraphExec_t x) {
  if (g_graphs.size() >= SL_GRAPH_CACHE) {   // evict the least recently used
    size_t lru = 0;
    for (size_t i = 1; i < g_graphs.size(); ++i)
      if (g_graphs[i].stamp < g_graphs[lru].stamp) lru = i;
    (void)hipGraphExecDestroy(g_graphs[lru].exec);
    (void)hipGraphDestroy(g_graphs[lru].graph);
    g_graphs.erase(g_graphs.begin() + lru);
  }
  g_graphs.push_back(DecodeGraphEntry{k, g, x, ++g_graph_clock});
}

extern "C" size_t sl_generate_workspace_bytes(const sl_llama_model* m, int64_t n_tok, int32_t nseq, int32_t max_new_tokens) {
  LlamaWs w;
  size_t a = llama_carve(m, n_tok > nseq ? n_tok : nseq, nseq, nullptr, 0, w);
  a += (size_t)nseq * m->hidden * sl_dtype_size(m->dtype) + 256;    // decode x
  a += (size_t)nseq * m->vocab * sizeof(float) + 256;               // logits
  a += ((size_t)nseq * (8 + max_new_tokens)) * sizeof(int32_t) + 11 * 256;
  return a;
}

// ---- compaction of a batch whose rows finish at different steps (sl_generate_opts.compact) ----
// K / V rows [0, len) of slot `src` copied to slot `dst` in every layer: grid (pair, layer * n_kv); src >= every dst (the movers sit
// above the rows that stay), so no block reads what another writes
struct KvMove { int32_t src, dst, len, pad; };
__global__ __launch_bounds__(256) void kv_move_kernel(uint4* __restrict__ kc, uint4* __restrict__ vc, const KvMove* __restrict__ mv, int nkv, int slots,
                                                      int max_ctx, int row_vec) {
  const KvMove m = mv[blockIdx.x];
  const int l = blockIdx.y / nkv, h = blockIdx.y - l * nkv;
  const int64_t lay = (int64_t)l * slots * nkv * max_ctx * row_vec;
  const int64_t s0 = lay + ((int64_t)m.src * nkv + h) * max_ctx * row_vec, d0 = lay + ((int64_t)m.dst * nkv + h) * max_ctx * row_vec;
  const int n = m.len * row_vec;
  for (int i = threadIdx.x; i < n; i += 256) {
    kc[d0 + i] = kc[s0 + i];
    vc[d0 + i] = vc[s0 + i];
  }
}

// rungs of the row-count ladder a compacting generation steps down: each is a row count some kernel family runs at full blocks, and
// each costs one captured graph (kept in the per-thread cache)
static int compact_rung(int n_live) {
  // every 64 rows from 128 up (a captured graph per rung costs ~1 ms once and is cached; padding rows cost a step's per-row time every step:
  // with steps of 128 above 512 rows the synthetic stop mix of bench.py ran 6.2 % more rows x steps than useful tokens), powers of two and
  // their 1.5-multiples below
  static const int small[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128};
  for (int r : small)
    if (n_live <= r) return r;
  return (n_live + 63) / 64 * 64;
}

struct SampleOpts { float temperature; int top_k; float top_p; uint64_t seed; };

static int generate_impl(const sl_llama_model* m, const sl_kv_cache* kv, void* x, const int32_t* cu_seqlens_host, int32_t nseq,
                         const sl_generate_opts* o, int32_t* out_ids_host, sl_generate_stats* stats, void* workspace, size_t workspace_bytes,
                         sl_stream stream) {
  SL_TRY(llama_check(m, kv));
  SL_CHECK_ARG(o != nullptr, "sl_generate: null options");
  // use_eos = 0: the EOS ids are ignored even when the caller left them in the struct — rows then finish on their budgets only
  const int max_new_tokens = o->max_new_tokens, n_eos = o->use_eos ? o->n_eos : 0, pad_id = o->pad_id;
  const int32_t* eos_ids_host = o->eos_ids_host;
  const int use_eos = (o->use_eos || o->row_limits_host) ? 1 : 0;       // per-row budgets finish rows the way EOS does
  SL_CHECK_ARG(x && cu_seqlens_host && out_ids_host && workspace && nseq > 0 && nseq <= SL_MAX_DECODE_BATCH && max_new_tokens > 0,
               "sl_generate: bad arguments (nseq<=%d)", SL_MAX_DECODE_BATCH);
  SL_CHECK_ARG(n_eos >= 0 && n_eos <= 8 && (n_eos == 0 || eos_ids_host != nullptr), "sl_generate: 0..8 eos ids");
  SampleOpts smp_s{o->temperature, o->top_k, o->top_p, o->seed};
  const SampleOpts* smp = o->sample ? &smp_s : nullptr;
  if (smp) SL_CHECK_ARG(smp->temperature > 0.f && smp->top_p > 0.f && smp->top_p <= 1.0f && smp->top_k >= 0, "sl_generate: sampling needs temperature > 0, 0 < top_p <= 1, top_k >= 0");
  hipStream_t st = (hipStream_t)stream;
  const int64_t n_tok = cu_seqlens_host[nseq];
  const int B0 = nseq;
  for (int s = 0; s < nseq; ++s)
    SL_CHECK_ARG(cu_seqlens_host[s + 1] - cu_seqlens_host[s] + max_new_tokens <= kv->max_ctx,
                 "sl_generate: prompt %d (%d tokens) + %d new tokens exceeds max_ctx %d", s,
                 cu_seqlens_host[s + 1] - cu_seqlens_host[s], max_new_tokens, kv->max_ctx);
  for (int s = 0; s < nseq; ++s)
    SL_CHECK_ARG(kv->shared_prefix <= cu_seqlens_host[s + 1] - cu_seqlens_host[s], "sl_generate: kv cache shared_prefix %d exceeds prompt %d (%d tokens)",
                 kv->shared_prefix, s, cu_seqlens_host[s + 1] - cu_seqlens_host[s]);
  if (o->row_limits_host)
    for (int s = 0; s < nseq; ++s)
      SL_CHECK_ARG(o->row_limits_host[s] >= 1 && o->row_limits_host[s] <= max_new_tokens, "sl_generate: row limit %d of sequence %d outside [1, max_new_tokens=%d]",
                   o->row_limits_host[s], s, max_new_tokens);
  SL_CHECK_ARG(sl_generate_workspace_bytes(m, n_tok, nseq, max_new_tokens) <= workspace_bytes, "sl_generate: workspace %zu B < required %zu B",
               workspace_bytes, sl_generate_workspace_bytes(m, n_tok, nseq, max_new_tokens));
  // carve: generation state first, then the prefill/decode scratch
  Carver c(workspace, workspace_bytes);
  float* logits = (float*)c.take((size_t)B0 * m->vocab * sizeof(float));
  void* xdec = c.take((size_t)B0 * m->hidden * sl_dtype_size(m->dtype));
  int32_t* unfinished = (int32_t*)c.take(B0 * sizeof(int32_t));
  int32_t* ctx_len = (int32_t*)c.take(B0 * sizeof(int32_t));
  int32_t* gen_count = (int32_t*)c.take(B0 * sizeof(int32_t));
  int32_t* finish_len = (int32_t*)c.take(B0 * sizeof(int32_t));
  int32_t* next_ids = (int32_t*)c.take(B0 * sizeof(int32_t));
  int32_t* out_ids = (int32_t*)c.take((size_t)B0 * max_new_tokens * sizeof(int32_t));
  int32_t* choice = (int32_t*)c.take(B0 * sizeof(int32_t));      // sampling mode: the drawn token of every row
  int32_t* row_limit = (int32_t*)c.take(B0 * sizeof(int32_t));   // per-row token budgets (sl_generate_opts.row_limits_host)
  int32_t* row_id = (int32_t*)c.take(B0 * sizeof(int32_t));      // the caller's index of the sequence a row holds (rows move when the batch is compacted)
  c.take(0);
  void* scratch = bptr(workspace) + c.off;
  const size_t scratch_bytes = workspace_bytes - c.off;
  const int32_t* row_limit_arg = o->row_limits_host ? row_limit : nullptr;

  std::vector<int32_t> ones(B0, 1), zeros(B0, 0), orig(B0);
  for (int b = 0; b < B0; ++b) orig[b] = b;
  SL_HIP(hipMemcpyAsync(unfinished, ones.data(), B0 * sizeof(int32_t), hipMemcpyHostToDevice, st));
  SL_HIP(hipMemcpyAsync(gen_count, zeros.data(), B0 * sizeof(int32_t), hipMemcpyHostToDevice, st));
  SL_HIP(hipMemcpyAsync(finish_len, zeros.data(), B0 * sizeof(int32_t), hipMemcpyHostToDevice, st));
  SL_HIP(hipMemcpyAsync(row_id, orig.data(), B0 * sizeof(int32_t), hipMemcpyHostToDevice, st));
  if (o->row_limits_host) SL_HIP(hipMemcpyAsync(row_limit, o->row_limits_host, B0 * sizeof(int32_t), hipMemcpyHostToDevice, st));
  SL_HIP(hipMemsetAsync(out_ids, 0, (size_t)B0 * max_new_tokens * sizeof(int32_t), st));

  hipEvent_t ev[3];
  for (auto& e : ev) SL_HIP(hipEventCreate(&e));
  SL_HIP(hipEventRecord(ev[0], st));
  SL_TRY(sl_llama_prefill(m, kv, x, cu_seqlens_host, nseq, logits, ctx_len, nullptr, scratch, scratch_bytes, stream));
  auto select = [&](int B, int advance_ctx, hipStream_t s_) -> int {
    if (smp)
      return sl_sample_select_impl(logits, B, m->vocab, smp->temperature, smp->top_k, smp->top_p, smp->seed, eos_ids_host, n_eos, pad_id, use_eos, advance_ctx,
                                   unfinished, ctx_len, gen_count, finish_len, next_ids, out_ids, max_new_tokens, choice, s_, row_limit_arg, row_id);
    return sl_greedy_select_impl(logits, B, m->vocab, eos_ids_host, n_eos, pad_id, use_eos, advance_ctx, unfinished, ctx_len, gen_count, finish_len, next_ids,
                                 out_ids, max_new_tokens, s_, row_limit_arg);
  };
  SL_TRY(select(B0, 0, st));
  SL_HIP(hipEventRecord(ev[1], st));

  // The decode step for `B` rows: captured ONCE per (model, cache, workspace, row count, limits) and kept in a small per-thread
  // cache — a second call with the same buffers (the usual case: a serving loop reusing its KV cache and workspace) replays it
  // without re-capturing.  Every pointer the captured launches were recorded with is part of the key (the state arrays and the
  // scratch layout follow from the workspace pointer, the ORIGINAL batch B0 and the row count B), so a changed buffer can never
  // replay a stale graph.  Also resets the per-graph scratch (K-split counters, split-attention counters, the slot index list).
  // A compacting generation keeps the kernel family of the batch it started with (common.h sl_family_rows): measured at full depth in bf16,
  // 819 of 1 024 random-init sequences changed ids when each rung picked its own family (near-ties flip between the families' rounding
  // orders) — pinned, a sequence's ids are those of the uncompacted batch, bit for bit.  SL_COMPACT_PIN=0: the old behaviour (A/B only).
  const int pin_rows = (o->compact && use_eos && sl_env().compact_pin) ? B0 : 0;
  auto graph_for = [&](int B, hipGraphExec_t* exec_out) -> int {
    SlFamilyPin pin(pin_rows);
    LlamaWs w;
    llama_carve(m, B, B, scratch, scratch_bytes, w);
    if (w.split && w.split_bytes >= 8192) SL_HIP(hipMemsetAsync(w.split, 0, 8192, st));   // the K-split fix-up's counters start at zero (sl_gemm_fused.split_ws)
    SL_TRY(sl_attn_decode_split_zero_counters(w.part, B, m->n_heads, m->n_kv_heads, kv->max_ctx, st));   // split attention merges its records in-launch: arrival counters start at zero
    hipLaunchKernelGGL(iota_kernel, dim3((B + 255) / 256), dim3(256), 0, st, w.tok_seq, B);
    SL_CHECK_LAUNCH("iota");
    DecodeGraphKey key;
    memset(&key, 0, sizeof(key));
    key.model = m; key.layers = m->layers; key.w0 = m->n_layers > 0 ? m->layers[0].wqkv_dec : nullptr; key.lm = m->lm_head_dec ? m->lm_head_dec : m->lm_head;
    key.embed = m->embed; key.kc = kv->k_cache; key.vc = kv->v_cache; key.ws = workspace; key.ws_bytes = workspace_bytes;
    key.B = B; key.B0 = B0; key.max_new = max_new_tokens; key.use_eos = use_eos; key.n_eos = n_eos; key.pad = pad_id; key.max_ctx = kv->max_ctx;
    key.slots = kv->slots; key.shared_prefix = kv->shared_prefix; key.dtype = m->dtype; key.n_layers = m->n_layers; key.vocab = m->vocab; key.fused = m->dec_fused_norm | (sl_env().decode_tiled << 8) | ((sl_env().attn_decode_ks & 127) << 9) | ((pin_rows ? 1 : 0) << 16) | ((sl_env().decode_prefetch ? 1 : 0) << 17);   // + the switch that shapes the captured launches
    key.limits = row_limit_arg ? 1 : 0;
    key.content = model_content_hash(m);
    SL_HIP(hipGetDevice(&key.device));
    for (int i = 0; i < n_eos && i < 8; ++i) key.eos[i] = eos_ids_host[i];
    if (smp) { key.sample = 1; key.temperature = smp->temperature; key.top_k = smp->top_k; key.top_p = smp->top_p; key.seed = smp->seed; }
    hipGraphExec_t exec = decode_graph_lookup(key);
    if (!exec) {
      hipGraph_t graph = nullptr;
      // Capture on a private stream (the caller's may be the legacy null stream, which cannot capture);
      // capturing records the launches without running them, the graph is then replayed on `st`.
      static thread_local hipStream_t cap_by_dev[SL_MAX_DEVICES] = {};    // a stream belongs to the device that was current when it was made
      SL_CHECK_ARG(key.device >= 0 && key.device < SL_MAX_DEVICES, "sl_generate: device index %d", key.device);
      hipStream_t& cap = cap_by_dev[key.device];
      if (!cap) SL_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
      SL_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal));
      // greedy + a batch the tiled lm_head serves: the lm_head leaves per-group maxima, the select pass reads 1/64 of the bytes
      const bool top1 = !smp && decode_fuses_argmax(m, B);
      // small batches (the skinny kernels: every matrix read once per step by a chain of short launches): a prefetch branch beside the chain
      DecodePrefetch pf;
      static thread_local hipStream_t cap2_by_dev[SL_MAX_DEVICES] = {};
      static thread_local hipEvent_t pev_by_dev[SL_MAX_DEVICES] = {};
      if (sl_env().decode_prefetch && B <= sl_env().stream_min_m && m->dtype == SL_BF16) {
        hipStream_t& cap2 = cap2_by_dev[key.device];
        hipEvent_t& pev = pev_by_dev[key.device];
        if (!cap2) SL_HIP(hipStreamCreateWithFlags(&cap2, hipStreamNonBlocking));
        if (!pev) SL_HIP(hipEventCreateWithFlags(&pev, hipEventDisableTiming));
        pf.main = cap; pf.side = cap2; pf.ev = pev; pf.sink = (uint32_t*)w.tok_pos; pf.on = true;      // (the sink is never written: see the kernel)
        g_prefetch = &pf;
      }
      int rc = decode_step(m, kv, next_ids, ctx_len, B, logits, xdec, w, cap, top1);
      g_prefetch = nullptr;
      if (pf.on) { const int rj = pf.join(); if (rc == 0) rc = rj; }       // the branch joins the chain before the capture ends
      if (rc == 0) {
        if (top1) {
          const int ng = (m->vocab + 63) / 64;
          rc = sl_greedy_select_partial_impl(logits, (const int32_t*)(logits + (int64_t)ng * B), ng, B, eos_ids_host, n_eos, pad_id, use_eos, 1, unfinished,
                                             ctx_len, gen_count, finish_len, next_ids, out_ids, max_new_tokens, cap, row_limit_arg);
        } else {
          rc = select(B, 1, cap);
        }
      }
      hipError_t ce = hipStreamEndCapture(cap, &graph);
      if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
      if (ce != hipSuccess) { sl_set_error("hipStreamEndCapture: %s", hipGetErrorString(ce)); return SL_ERR_LAUNCH; }
      SL_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
      decode_graph_store(key, graph, exec);
    }
    *exec_out = exec;
    return 0;
  };

  // host copies of the per-row state, filled at a check; `done[s]`: sequence s (caller's index) has been written to out_ids_host
  std::vector<int32_t> h_unf(B0, 1), h_fin(B0, 0), h_ctx, h_cnt, h_next, h_lim, h_out;
  std::vector<char> done(B0, 0);
  std::vector<int32_t> fin_of(B0, 0), unf_of(B0, 1);
  int steps_done = 1, B = B0, compactions = 0, launches = 0;
  int64_t row_steps = 0;
  bool all_done = false;
  // rows [0, B) -> the caller's output rows, for the sequences not written yet (a retired sequence's row may live on as padding of
  // the compacted batch: its pads are not written again)
  auto retire = [&](bool only_finished) {
    for (int r = 0; r < B; ++r) {
      const int s = orig[r];
      if (done[s] || (only_finished && h_unf[r] != 0)) continue;
      memcpy(out_ids_host + (size_t)s * max_new_tokens, h_out.data() + (size_t)r * max_new_tokens, (size_t)max_new_tokens * sizeof(int32_t));
      fin_of[s] = h_fin[r]; unf_of[s] = h_unf[r];
      done[s] = 1;
    }
  };
  if (max_new_tokens > 1) {
    hipGraphExec_t exec = nullptr;
    SL_TRY(graph_for(B, &exec));
    const int check_every = o->check_every > 0 ? o->check_every : 16;
    while (steps_done < max_new_tokens) {
      SL_HIP(hipGraphLaunch(exec, st));
      ++steps_done; ++launches; row_steps += B;
      if (!(use_eos && (steps_done % check_every == 0) && steps_done < max_new_tokens)) continue;
      SL_HIP(hipMemcpyAsync(h_unf.data(), unfinished, B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SL_HIP(hipStreamSynchronize(st));
      int n_live = 0;
      for (int b = 0; b < B; ++b) n_live += (h_unf[b] != 0);
      if (n_live == 0) { all_done = true; break; }
      const int rung = compact_rung(n_live);
      if (!o->compact || rung >= B) continue;
      // ---- compact: the live rows that sit at or above `rung` (the movers) take the places of finished rows below it; live rows
      // below it and the remaining finished ones (padding of the rung) stay where they are.  Finished sequences are written to the
      // caller's buffer first.  Everything per-row moves with the row: state arrays (through the host: a few KB), output ids, and
      // the row's K / V slot (on the device).
      h_ctx.resize(B); h_cnt.resize(B); h_next.resize(B); h_lim.resize(B); h_out.resize((size_t)B * max_new_tokens);
      SL_HIP(hipMemcpyAsync(h_fin.data(), finish_len, B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SL_HIP(hipMemcpyAsync(h_ctx.data(), ctx_len, B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SL_HIP(hipMemcpyAsync(h_cnt.data(), gen_count, B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SL_HIP(hipMemcpyAsync(h_next.data(), next_ids, B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      if (row_limit_arg) SL_HIP(hipMemcpyAsync(h_lim.data(), row_limit, B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SL_HIP(hipMemcpyAsync(h_out.data(), out_ids, (size_t)B * max_new_tokens * sizeof(int32_t), hipMemcpyDeviceToHost, st));
      SL_HIP(hipStreamSynchronize(st));
      retire(true);
      std::vector<KvMove> moves;
      int hole = 0;
      for (int r = rung; r < B; ++r) {
        if (h_unf[r] == 0) continue;
        while (h_unf[hole] != 0) ++hole;          // exists: at most `rung` rows are live
        moves.push_back(KvMove{r, hole, h_ctx[r], 0});
        h_unf[hole] = h_unf[r]; h_fin[hole] = h_fin[r]; h_ctx[hole] = h_ctx[r]; h_cnt[hole] = h_cnt[r]; h_next[hole] = h_next[r]; h_lim[hole] = h_lim[r];
        memcpy(h_out.data() + (size_t)hole * max_new_tokens, h_out.data() + (size_t)r * max_new_tokens, (size_t)max_new_tokens * sizeof(int32_t));
        orig[hole] = orig[r];
        ++hole;
      }
      if (!moves.empty()) {
        const int row_vec = (int)(m->head_dim * sl_dtype_size(m->dtype) / 16);
        KvMove* mv_dev = (KvMove*)logits;         // the logits buffer is idle between steps (>= B0 * vocab floats)
        SL_CHECK_ARG(moves.size() * sizeof(KvMove) <= (size_t)B0 * m->vocab * sizeof(float), "sl_generate: move list does not fit");
        SL_HIP(hipMemcpyAsync(mv_dev, moves.data(), moves.size() * sizeof(KvMove), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(kv_move_kernel, dim3((unsigned)moves.size(), m->n_layers * m->n_kv_heads), dim3(256), 0, st, (uint4*)kv->k_cache, (uint4*)kv->v_cache,
                           mv_dev, m->n_kv_heads, kv->slots, kv->max_ctx, row_vec);
        SL_CHECK_LAUNCH("kv_move");
      }
      B = rung;
      SL_HIP(hipMemcpyAsync(unfinished, h_unf.data(), B * sizeof(int32_t), hipMemcpyHostToDevice, st));
      SL_HIP(hipMemcpyAsync(finish_len, h_fin.data(), B * sizeof(int32_t), hipMemcpyHostToDevice, st));
      SL_HIP(hipMemcpyAsync(ctx_len, h_ctx.data(), B * sizeof(int32_t), hipMemcpyHostToDevice, st));
      SL_HIP(hipMemcpyAsync(gen_count, h_cnt.data(), B * sizeof(int32_t), hipMemcpyHostToDevice, st));
      SL_HIP(hipMemcpyAsync(next_ids, h_next.data(), B * sizeof(int32_t), hipMemcpyHostToDevice, st));
      SL_HIP(hipMemcpyAsync(row_id, orig.data(), B * sizeof(int32_t), hipMemcpyHostToDevice, st));
      if (row_limit_arg) SL_HIP(hipMemcpyAsync(row_limit, h_lim.data(), B * sizeof(int32_t), hipMemcpyHostToDevice, st));
      SL_HIP(hipMemcpyAsync(out_ids, h_out.data(), (size_t)B * max_new_tokens * sizeof(int32_t), hipMemcpyHostToDevice, st));
      SL_HIP(hipStreamSynchronize(st));           // the host vectors above are reused / the move list is overwritten by the next step's logits
      SL_TRY(graph_for(B, &exec));
      ++compactions;
    }
  }
  SL_HIP(hipEventRecord(ev[2], st));
  // results: queued behind the last step on the caller's stream, ONE synchronisation for the whole call (plus one per check above)
  h_out.resize((size_t)B * max_new_tokens);
  SL_HIP(hipMemcpyAsync(h_unf.data(), unfinished, B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  SL_HIP(hipMemcpyAsync(h_fin.data(), finish_len, B * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  SL_HIP(hipMemcpyAsync(h_out.data(), out_ids, (size_t)B * max_new_tokens * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  SL_HIP(hipStreamSynchronize(st));
  retire(false);
  int n_cols = steps_done;
  if (use_eos) {
    all_done = true;
    int mx = 0;
    for (int s = 0; s < B0; ++s) { all_done = all_done && unf_of[s] == 0; if (fin_of[s] > mx) mx = fin_of[s]; }
    if (all_done) n_cols = mx;  // HF stops at the step where the last row emitted its EOS
    // a sequence retired at a compaction has pads up to the step of its retirement only: pad the rest of its row, as the uncompacted
    // batch would have (hf:generation/utils.py:2928-2929)
    for (int s = 0; s < B0; ++s)
      if (unf_of[s] == 0)
        for (int t = fin_of[s]; t < max_new_tokens; ++t) out_ids_host[(size_t)s * max_new_tokens + t] = pad_id;
  }
  if (stats) {
    memset(stats, 0, sizeof(*stats));
    stats->n_steps = n_cols; stats->decode_launches = launches; stats->compactions = compactions; stats->final_rows = B; stats->row_steps = row_steps;
    SL_HIP(hipEventElapsedTime(&stats->prefill_ms, ev[0], ev[1]));
    SL_HIP(hipEventElapsedTime(&stats->decode_ms, ev[1], ev[2]));
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  return 0;
}

extern "C" int sl_generate(const sl_llama_model* m, const sl_kv_cache* kv, void* x, const int32_t* cu_seqlens_host, int32_t nseq,
                           const sl_generate_opts* opts, int32_t* out_ids_host, sl_generate_stats* stats, void* workspace, size_t workspace_bytes,
                           sl_stream stream) {
  return generate_impl(m, kv, x, cu_seqlens_host, nseq, opts, out_ids_host, stats, workspace, workspace_bytes, stream);
}

static int generate_compat(const sl_llama_model* m, const sl_kv_cache* kv, void* x, const int32_t* cu_seqlens_host, int32_t nseq, sl_generate_opts& o,
                           int32_t* out_ids_host, int32_t* n_steps_host, float* timings_ms_host, void* workspace, size_t workspace_bytes, sl_stream stream) {
  SL_CHECK_ARG(n_steps_host != nullptr, "sl_greedy_generate: null n_steps");
  sl_generate_stats stt;
  SL_TRY(generate_impl(m, kv, x, cu_seqlens_host, nseq, &o, out_ids_host, &stt, workspace, workspace_bytes, stream));
  *n_steps_host = stt.n_steps;
  if (timings_ms_host) { timings_ms_host[0] = stt.prefill_ms; timings_ms_host[1] = stt.decode_ms; }
  return 0;
}

extern "C" int sl_greedy_generate(const sl_llama_model* m, const sl_kv_cache* kv, void* x, const int32_t* cu_seqlens_host, int32_t nseq,
                                  int32_t max_new_tokens, const int32_t* eos_ids_host, int32_t n_eos, int32_t pad_id, int32_t use_eos,
                                  int32_t check_every, int32_t* out_ids_host, int32_t* n_steps_host, float* timings_ms_host,
                                  void* workspace, size_t workspace_bytes, sl_stream stream) {
  sl_generate_opts o;
  memset(&o, 0, sizeof(o));
  o.max_new_tokens = max_new_tokens; o.eos_ids_host = eos_ids_host; o.n_eos = n_eos; o.pad_id = pad_id; o.use_eos = use_eos; o.check_every = check_every;
  return generate_compat(m, kv, x, cu_seqlens_host, nseq, o, out_ids_host, n_steps_host, timings_ms_host, workspace, workspace_bytes, stream);
}

extern "C" int sl_sample_generate(const sl_llama_model* m, const sl_kv_cache* kv, void* x, const int32_t* cu_seqlens_host, int32_t nseq,
                                  int32_t max_new_tokens, const int32_t* eos_ids_host, int32_t n_eos, int32_t pad_id, int32_t use_eos,
                                  int32_t check_every, float temperature, int32_t top_k, float top_p, uint64_t seed, int32_t* out_ids_host,
                                  int32_t* n_steps_host, float* timings_ms_host, void* workspace, size_t workspace_bytes, sl_stream stream) {
  SL_CHECK_ARG(temperature > 0.f && top_p > 0.f && top_p <= 1.0f && top_k >= 0, "sl_sample_generate: need temperature > 0, 0 < top_p <= 1, top_k >= 0");
  sl_generate_opts o;
  memset(&o, 0, sizeof(o));
  o.max_new_tokens = max_new_tokens; o.eos_ids_host = eos_ids_host; o.n_eos = n_eos; o.pad_id = pad_id; o.use_eos = use_eos; o.check_every = check_every;
  o.sample = 1; o.temperature = temperature; o.top_k = top_k; o.top_p = top_p; o.seed = seed;
  return generate_compat(m, kv, x, cu_seqlens_host, nseq, o, out_ids_host, n_steps_host, timings_ms_host, workspace, workspace_bytes, stream);
}
