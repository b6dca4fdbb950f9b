// train_tape.hip — C++ host runtime of the KD step's layer stacks (ref:trainer.py:270-384): the forward-with-tape and the
// backward of the encoder's pre-LN transformer layers (HuBERT / Whisper, data + parameter gradients) and of the frozen
// Llama decoder (data gradients only) over a PACKED ragged batch, as plain launch sequences over the library's own kernels.
// The Python tape (training.py) used to issue these ~3 500 launches per accumulation window one ctypes call at a time and was
// host-bound (150 ms of kernels in 236 ms of wall time); here one call covers a whole stack.  Saved activations and
// gradient accumulators are caller-owned buffers handed in as plain structs; temporaries come from one workspace.
#include <vector>

#include "common.h"

// train_ops.hip: norm backward with the residual branch's gradient added in the same pass (dx = backward(dy) + add)
// ... and, optionally, dropout(dx) as a second output (dx_drop: the incoming gradient of the sublayer below) / a second joining gradient (add2)
int sl_layernorm_bwd_ws_add_impl(const void* x, const void* gamma, const void* beta, const void* dy, const void* add, void* dx, float* dgamma, float* dbeta,
                                 int64_t rows, int32_t cols, float eps, int32_t dtype, void* workspace, size_t workspace_bytes, sl_stream stream,
                                 void* dx_drop, float drop_p, uint64_t drop_seed, const float* dy_parts, int dy_splits, int32_t* colred_cnt);
int sl_rmsnorm_bwd_add_impl(const void* x, const void* w, const void* dy, const void* add, void* dx, int64_t rows, int32_t cols, float eps, int32_t dtype,
                            sl_stream stream, const void* add2, const float* dy_parts, int dy_splits);

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
};

inline int64_t rup(int64_t n, int64_t m) { return (n + m - 1) / m * m; }

// epilogue fusion of a product (sl_gemm_ex_args.post_op ...): the element-wise launch that used to follow it
struct Post {
  int op = SL_POST_NONE;
  float p = 0.f; uint64_t seed = 0; int64_t ld = 0;     // dropout mask (p = 0: none)
  const void* in = nullptr; int64_t in_ld = 0;          // GELU_BWD: pre-activation; SILU_MUL_BWD: gu
  float* colsum = nullptr;                              // bias gradient of the stored values
};

// C = act(A W^T + bias) + residual   (ops.gemm)
// sk_ws: optional stream-K workspace (sl_gemm_ex_args.sk_ws) for products of a few tiles under a long reduction
int gemm(int dt, const void* A, int64_t lda, const void* W, int64_t ldw, void* C, int64_t ldc, const void* bias, const void* res, int64_t ldr,
         int64_t M, int N, int K, int act, void* aux_out, hipStream_t st, void* sk_ws = nullptr, const Post* post = nullptr, int32_t* defer = nullptr) {
  sl_gemm_args a;
  memset(&a, 0, sizeof(a));
  a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc; a.bias = bias; a.residual = res; a.ldr = ldr;
  a.M = (int)M; a.N = N; a.K = K; a.batch = 1; a.dtype = dt; a.act = act;
  if (defer) *defer = 0;
  if (!aux_out && !sk_ws && !post) return sl_gemm(&a, (sl_stream)st);
  sl_gemm_ex_args ex;
  memset(&ex, 0, sizeof(ex));
  ex.aux_out = aux_out; ex.w_mod = 1;
  if (sk_ws) { ex.sk_ws = sk_ws; ex.sk_ws_bytes = sl_gemm_streamk_workspace_bytes(); }
  if (post) {
    ex.post_op = post->op; ex.drop_p = post->p; ex.drop_seed = post->seed; ex.drop_ld = post->ld;
    ex.post_in = post->in; ex.post_ld = post->in_ld; ex.colsum_out = post->colsum;
  }
  if (sk_ws && sl_env().tape_fuse) ex.deferred_splits = defer;      // the K runs of a few-tile product stay in sk_ws for a norm backward that sums them while loading
  return sl_gemm_ex(&a, &ex, (sl_stream)st);
}

// the fused forms need the tiled kernels' row range and whole K slabs in every product of the layer (sl_gemm_post_ok) and SL_TAPE_FUSE != 0
// (A/B switch; default on)
inline bool fuse_ok(int64_t M, int dt, int H, int F) { return sl_env().tape_fuse != 0 && sl_gemm_post_ok(M, H, F, dt) && sl_gemm_post_ok(M, F, H, dt); }

// scratch for the K-contiguous operand copies of the backward products
struct BwdScratch {
  void* yt;   // dY^T  (N_out_max, Mp)
  void* xt;   // X^T   (K_in_max, Mp)
  void* wt;   // W^T   (K_in_max, ld(N_out_max))
};

// dX (M, K_in) = dY (M, N_out) . W (N_out, K_in), through a transposed copy of W so that the product is K-contiguous
// (ops.dgrad with wt = ops.transpose_pad(W)); wt_cached != NULL: the copy already exists (frozen weights)
int dgrad(int dt, const void* dY, int64_t ldy, const void* W, int n_out, int k_in, const void* wt_cached, void* dX, int64_t ldx, int64_t M,
          const BwdScratch& s, hipStream_t st, void* sk_ws = nullptr, const Post* post = nullptr, int32_t* defer = nullptr) {
  const int vec = dt == SL_F32 ? 4 : 8;
  const int64_t ldw = rup(n_out, vec);
  const void* wt = wt_cached;
  if (!wt) {
    SL_TRY(sl_transpose_pad(W, k_in, s.wt, ldw, n_out, k_in, n_out, dt, (sl_stream)st));
    wt = s.wt;
  }
  return gemm(dt, dY, ldy, wt, wt_cached ? n_out : ldw, dX, ldx, nullptr, nullptr, 0, M, k_in, n_out, SL_ACT_NONE, nullptr, st, sk_ws, post, defer);
}

// dW (N_out, K_in) fp32 += dY^T (N_out, M) . X (M, K_in)       (ops.wgrad_acc, plain Linear)
// sk_ws: split-K workspace OF THE STREAM `st` (few output tiles under thousands of token rows: the 1 024 x 1 024 out_proj gradient)
// db (optional): the bias gradient db += colsum(dY) — inside the token-major product where that kernel runs (sl_gemm_ex_args.colsum_out with
// both operands transposed), else as a sl_colsum launch behind it
int wgrad_acc(int dt, const void* dY, int64_t ldy, int n_out, const void* X, int64_t ldx, int k_in, float* dW, int64_t M, const BwdScratch& s,
              hipStream_t st, void* sk_ws = nullptr, float* db = nullptr) {
  sl_gemm_args a;
  memset(&a, 0, sizeof(a));
  sl_gemm_ex_args ex;
  memset(&ex, 0, sizeof(ex));
  ex.w_mod = 1; ex.residual_f32 = 1;
  a.C = dW; a.ldc = k_in; a.residual = dW; a.ldr = k_in; a.M = n_out; a.N = k_in; a.batch = 1; a.dtype = dt; a.out_f32 = 1;
  if (M >= 256 && sl_env().wgrad_tr && dt == SL_BF16 && n_out % 128 == 0 && k_in % 128 == 0 && ldy % 8 == 0 && ldx % 8 == 0) {
    // both operands as stored (token-major): gemm_tiled_tt_kernel gathers its fragments with transposing LDS reads
    ex.trans_a = 1; ex.trans_w = 1;
    a.A = dY; a.lda = ldy; a.W = X; a.ldw = ldx; a.K = (int)M;
    if (sk_ws) { ex.sk_ws = sk_ws; ex.sk_ws_bytes = sl_gemm_streamk_workspace_bytes(); }
    if (sl_env().tape_fuse) ex.colsum_out = db;
    SL_TRY(sl_gemm_ex(&a, &ex, (sl_stream)st));
    if (db && !ex.colsum_out) SL_TRY(sl_colsum(dY, ldy, db, M, n_out, dt, (sl_stream)st));
    return 0;
  }
  if (db) SL_TRY(sl_colsum(dY, ldy, db, M, n_out, dt, (sl_stream)st));
  if (M >= 256) {
    // contract over token rows on K-contiguous copies (dY^T, X^T zero-padded to whole K slabs): LDS-DMA tiled kernels
    const int64_t Mp = rup(M, sk_ws ? 128 : 64);      // an even number of K slabs, so that the reduction can be cut in two runs (gemm.hip splitk_runs)
    SL_TRY(sl_transpose_pad(dY, ldy, s.yt, Mp, (int)M, n_out, (int)Mp, dt, (sl_stream)st));
    SL_TRY(sl_transpose_pad(X, ldx, s.xt, Mp, (int)M, k_in, (int)Mp, dt, (sl_stream)st));
    a.A = s.yt; a.lda = Mp; a.W = s.xt; a.ldw = Mp; a.K = (int)Mp;
    if (sk_ws) { ex.sk_ws = sk_ws; ex.sk_ws_bytes = sl_gemm_streamk_workspace_bytes(); }
    return sl_gemm_ex(&a, &ex, (sl_stream)st);
  }
  ex.trans_a = 1; ex.trans_w = 1;
  a.A = dY; a.lda = ldy; a.W = X; a.ldw = ldx; a.K = (int)M;
  return sl_gemm_ex(&a, &ex, (sl_stream)st);
}

int attn_fwd(int dt, const void* qkv, int64_t qkv_w, void* out, float* lse, const int32_t* cu, const int32_t* klen, int nseq, int max_len, int nh,
             int nkv, int D, int causal, float scale, float p_drop, uint64_t seed, hipStream_t st) {
  const size_t sz = sl_dtype_size(dt);
  sl_attn_args a;
  memset(&a, 0, sizeof(a));
  a.q = qkv; a.q_row_stride = qkv_w; a.q_head_stride = D;
  a.k = (const unsigned char*)qkv + (size_t)nh * D * sz; a.k_row_stride = qkv_w; a.k_head_stride = D;
  a.v = (const unsigned char*)qkv + (size_t)(nh + nkv) * D * sz; a.v_row_stride = qkv_w; a.v_head_stride = D;
  a.out = out; a.o_row_stride = (int64_t)nh * D; a.o_head_stride = D;
  a.cu_q = cu; a.cu_k = cu; a.klen = klen;
  a.nseq = nseq; a.max_qlen = max_len; a.n_heads = nh; a.n_kv_heads = nkv; a.head_dim = D; a.causal = causal; a.dtype = dt;
  a.scale = scale; a.dropout_p = p_drop; a.dropout_seed = seed; a.lse = lse;
  return sl_attn_fwd(&a, (sl_stream)st);
}

int attn_bwd(int dt, const void* qkv, int64_t qkv_w, const void* out, const void* d_out, const float* lse, float* delta, void* d_qkv,
             const int32_t* cu, const int32_t* klen, int nseq, int max_len, int64_t n_tok, int nh, int nkv, int D, int causal, float scale,
             float p_drop, uint64_t seed, hipStream_t st) {
  const size_t sz = sl_dtype_size(dt);
  sl_attn_bwd_args a;
  memset(&a, 0, sizeof(a));
  const size_t koff = (size_t)nh * D * sz, voff = (size_t)(nh + nkv) * D * sz;
  a.q = qkv; a.q_row_stride = qkv_w; a.q_head_stride = D;
  a.k = (const unsigned char*)qkv + koff; a.k_row_stride = qkv_w; a.k_head_stride = D;
  a.v = (const unsigned char*)qkv + voff; a.v_row_stride = qkv_w; a.v_head_stride = D;
  a.out = out; a.o_row_stride = (int64_t)nh * D; a.o_head_stride = D;
  a.d_out = d_out; a.do_row_stride = (int64_t)nh * D; a.do_head_stride = D;
  a.dq = d_qkv; a.dq_row_stride = qkv_w; a.dq_head_stride = D;
  a.dk = (unsigned char*)d_qkv + koff; a.dk_row_stride = qkv_w; a.dk_head_stride = D;
  a.dv = (unsigned char*)d_qkv + voff; a.dv_row_stride = qkv_w; a.dv_head_stride = D;
  a.lse = lse; a.delta = delta; a.cu_q = cu; a.cu_k = cu; a.klen = klen; a.n_tok_q = n_tok;
  a.nseq = nseq; a.max_qlen = max_len; a.max_klen = max_len; a.n_heads = nh; a.n_kv_heads = nkv; a.head_dim = D; a.causal = causal; a.dtype = dt;
  a.scale = scale; a.dropout_p = p_drop; a.dropout_seed = seed;
  return sl_attn_bwd(&a, (sl_stream)st);
}

// The parameter-gradient products of a layer (dW += dY^T X, db += colsum dY) feed nothing in the backward chain; at KD window sizes
// each of them — and each data-gradient product beside it — fills a quarter to a half of the chip (tools/kd_gemm_shapes.py: 16 to 256
// tiles).  They go to a second stream, forked off the caller's stream where their dY is complete; the caller's stream waits for them
// only where it is about to overwrite a buffer they read, and once at the end.  One side stream and a ring of events per thread
// and device; SL_NO_WGRAD_STREAM=1 keeps everything on the caller's stream.
struct SideStream {
  hipStream_t side = nullptr;
  hipEvent_t fork = nullptr, done[4] = {nullptr, nullptr, nullptr, nullptr};
  bool pending[4] = {false, false, false, false};
  hipStream_t main_ = nullptr;
  bool on = false;
  int init(hipStream_t main, int64_t n_tok) {
    main_ = main;
    // from 512 token rows (SL_WGRAD_STREAM_MIN_TOK): at 998 rows — the per-rank window of an 8-rank step — the fork / join events cost more than the
    // overlap bought in round 5 (-1 %: the bound was 2 048 rows); with the ring kernels (one block per CU, a few dozen to 256 blocks per product) the
    // products of the two streams fill the chip together: 28.50 -> 28.08 ms (profiles/r06_au_kd_windows.txt).  Each stream has its own split-K workspace.
    on = !sl_env().no_wgrad_stream && n_tok >= (sl_env().wgrad_stream_min_tok > 0 ? sl_env().wgrad_stream_min_tok : 512);
    if (!on) return 0;
    int dev = 0;
    SL_HIP(hipGetDevice(&dev));
    SL_CHECK_ARG(dev >= 0 && dev < SL_MAX_DEVICES, "device %d out of range", dev);
    struct PerDev { hipStream_t s; hipEvent_t f; hipEvent_t d[4]; };
    static thread_local PerDev per[SL_MAX_DEVICES] = {};
    PerDev& pd = per[dev];
    if (!pd.s) {
      SL_HIP(hipStreamCreateWithFlags(&pd.s, hipStreamNonBlocking));
      SL_HIP(hipEventCreateWithFlags(&pd.f, hipEventDisableTiming));
      for (auto& e : pd.d) SL_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    side = pd.s; fork = pd.f;
    for (int i = 0; i < 4; ++i) done[i] = pd.d[i];
    return 0;
  }
  // the stream the k-th parameter-gradient group of a layer runs on, ordered after everything issued on the caller's stream so far
  int begin(hipStream_t& st) {
    st = main_;
    if (!on) return 0;
    SL_HIP(hipEventRecord(fork, main_));
    SL_HIP(hipStreamWaitEvent(side, fork, 0));
    st = side;
    return 0;
  }
  int end(int k) {
    if (!on) return 0;
    SL_HIP(hipEventRecord(done[k], side));
    pending[k] = true;
    return 0;
  }
  // the caller's stream may not pass this point before group k (and, the side stream being in order, every earlier one) has finished
  int join(int k) {
    if (!on || !pending[k]) return 0;
    SL_HIP(hipStreamWaitEvent(main_, done[k], 0));
    pending[k] = false;
    return 0;
  }
};

}  // namespace

// ================================================================================================
// encoder stack (hf:models/hubert/modeling_hubert.py:504-547 stable-LN layer; hf:models/whisper/modeling_whisper.py:360-414)
// ================================================================================================
struct EncWs {
  // Buffers the parameter-gradient groups read come in two copies, used by alternate (visited) layers: the side stream may still be working on a
  // layer's groups while the next layer's main chain runs, and the main stream then waits for it once per layer instead of once per group
  // (every event operation on the main stream cost a 5-20 us bubble: 253 of them = 2.6 ms of a 91 ms window, profiles/r06_p_kd_timeline.txt).
  // tmp_h: dropout(dx) (FFN2's incoming gradient), tmp_h2: dropout(d_h2) (the out-projection's)
  void *tmp_h[2], *tmp_h2[2], *d_pre1[2], *d_h2[2], *d_qkv[2];
  void *d_mid, *d_h1, *d_att;
  float* delta;
  void* ln_ws;         // per-block dgamma / dbeta records of the LayerNorm backward (sl_layernorm_bwd_ws)
  size_t ln_ws_bytes;
  void* sk;            // split-K workspace of the few-tile products of a short window (main-stream products only)
  void* sk_w;          // ... of the parameter-gradient group's stream (the side stream of a long window, else the caller's)
  BwdScratch s;
  // every layer's transposed weights for the data-gradient products, [layer][w2^T (F, H) | w1^T (H, F) | wo^T (H, H) | wqkv^T (H, 3 H)]: written by
  // the FORWARD call (one batched launch on a side stream), read by the backward calls on the same workspace (WtCache below)
  unsigned char* wt_all;
  size_t wt_layer_bytes;
  const void* wt(int l, int which, int64_t H, int64_t F, size_t sz) const {
    const size_t off[4] = {0, (size_t)(F * H) * sz, (size_t)(2 * F * H) * sz, (size_t)(2 * F * H + H * H) * sz};
    return wt_all + (size_t)l * wt_layer_bytes + off[which];
  }
};

// Which workspace holds current transposed weights, per thread and device: set by sl_encoder_stack_train_fwd, honoured by
// sl_encoder_stack_train_bwd only for the same (workspace, layer table, shape) — any other backward makes its own transposes as before.
struct WtCache {
  hipStream_t s = nullptr;
  hipEvent_t fork = nullptr, done = nullptr;
  const void* ws = nullptr;
  const void* layers = nullptr;
  int n_layers = 0, H = 0, F = 0, dt = -1;
  bool valid = false, waited = false;
};
static int wt_cache(WtCache*& out) {
  int dev = 0;
  SL_HIP(hipGetDevice(&dev));
  SL_CHECK_ARG(dev >= 0 && dev < SL_MAX_DEVICES, "device %d out of range", dev);
  static thread_local WtCache per[SL_MAX_DEVICES];
  WtCache& c = per[dev];
  if (!c.s) {
    SL_HIP(hipStreamCreateWithFlags(&c.s, hipStreamNonBlocking));
    SL_HIP(hipEventCreateWithFlags(&c.fork, hipEventDisableTiming));
    SL_HIP(hipEventCreateWithFlags(&c.done, hipEventDisableTiming));
  }
  out = &c;
  return 0;
}

static size_t enc_carve(const sl_enc_stack_cfg* c, void* base, size_t cap, EncWs& w) {
  const size_t sz = sl_dtype_size(c->dtype);
  const int64_t n = c->n_tok, H = c->hidden, F = c->ffn;
  const int64_t Mp = rup(n, 128);
  const int64_t big = F > 3 * H ? F : 3 * H;
  Carver cv(base, cap);
  for (int k = 0; k < 2; ++k) {
    w.tmp_h[k] = cv.take(n * H * sz);
    w.tmp_h2[k] = cv.take(n * H * sz);
    w.d_pre1[k] = cv.take(n * F * sz);
    w.d_h2[k] = cv.take(n * H * sz);
    w.d_qkv[k] = cv.take(n * 3 * H * sz);
  }
  w.d_mid = cv.take(n * F * sz);
  w.d_h1 = cv.take(n * H * sz);
  w.d_att = cv.take(n * H * sz);
  w.delta = (float*)cv.take(n * c->n_heads * sizeof(float));
  w.ln_ws_bytes = sl_layernorm_bwd_ws_bytes(n, (int32_t)H);
  w.ln_ws = cv.take(w.ln_ws_bytes);
  w.s.yt = cv.take(big * Mp * sz);
  w.s.xt = cv.take(big * Mp * sz);
  w.s.wt = cv.take(big * (big + 8) * sz);
  // a short window only (the per-rank share of a data-parallel step: ~1 000 frames): there FFN2 and the data gradients under K = 3 072 / 4 096
  // are 64 tiles on 256 CUs, and everything runs on one stream (SideStream::init), so one workspace serves the whole call
  w.sk = n < 2048 ? cv.take(sl_gemm_streamk_workspace_bytes()) : nullptr;
  w.sk_w = cv.take(sl_gemm_streamk_workspace_bytes());      // the weight gradients' own workspace (they may run on their own stream)
  const int vec = c->dtype == SL_F32 ? 4 : 8;
  w.wt_layer_bytes = (H % vec == 0 && F % vec == 0) ? (size_t)(2 * F * H + 4 * H * H) * sz : 0;
  w.wt_all = w.wt_layer_bytes ? (unsigned char*)cv.take(w.wt_layer_bytes * (size_t)c->n_layers) : nullptr;
  return cv.off + 256;
}

extern "C" size_t sl_encoder_stack_train_workspace_bytes(const sl_enc_stack_cfg* c) {
  EncWs w;
  return enc_carve(c, nullptr, 0, w);
}

extern "C" int sl_encoder_stack_train_fwd(const sl_hubert_layer* layers, const sl_enc_stack_cfg* c, const void* x_in, sl_enc_layer_saved* saved,
                                          const void** x_out, void* workspace, size_t workspace_bytes, sl_stream stream) {
  SL_CHECK_ARG(layers && c && x_in && saved && x_out && workspace && c->cu && c->klen && c->seeds && c->skip, "sl_encoder_stack_train_fwd: null pointer");
  SL_CHECK_ARG(c->hidden % c->n_heads == 0 && c->hidden / c->n_heads == 64, "sl_encoder_stack_train_fwd: head_dim must be 64");
  EncWs w;
  SL_CHECK_ARG(enc_carve(c, workspace, workspace_bytes, w) <= workspace_bytes, "sl_encoder_stack_train_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int dt = c->dtype, H = c->hidden, F = c->ffn, nh = c->n_heads;
  const int64_t n = c->n_tok;
  const void* x = x_in;
  if (w.sk) SL_HIP(hipMemsetAsync(w.sk, 0, 1024, st));      // stream-K flags: zero whenever no launch is in flight
  {
    // the backward's W^T copies, off its critical path: one batched launch beside this forward (the weights are final — the optimizer step
    // that wrote them is earlier on `stream`)
    WtCache* wc = nullptr;
    SL_TRY(wt_cache(wc));
    wc->valid = false;
    if (w.wt_all && sl_env().enc_wt_ahead) {
      std::vector<SlTransposeRec> recs;
      const size_t sz = sl_dtype_size(dt);
      for (int l = 0; l < c->n_layers; ++l) {
        if (c->skip[l]) continue;
        const sl_hubert_layer& L = layers[l];
        const void* src[4] = {L.w2, L.w1, L.wo, L.wqkv};
        const int n_out[4] = {H, F, H, 3 * H}, k_in[4] = {F, H, H, H};
        for (int k = 0; k < 4; ++k) {
          SlTransposeRec q;
          q.x = src[k]; q.y = (void*)w.wt(l, k, H, F, sz); q.ldx = k_in[k]; q.ldy = n_out[k]; q.rows = n_out[k]; q.cols = k_in[k]; q.ld_out = n_out[k]; q.tiles_r = 0;
          recs.push_back(q);
        }
      }
      if (!recs.empty()) {
        SL_HIP(hipEventRecord(wc->fork, st));
        SL_HIP(hipStreamWaitEvent(wc->s, wc->fork, 0));
        SL_TRY(sl_transpose_pad_batch_impl(recs.data(), (int)recs.size(), dt, (sl_stream)wc->s));
        SL_HIP(hipEventRecord(wc->done, wc->s));
        wc->ws = workspace; wc->layers = layers; wc->n_layers = c->n_layers; wc->H = H; wc->F = F; wc->dt = dt;
        wc->valid = true; wc->waited = false;
      }
    }
  }
  for (int l = 0; l < c->n_layers; ++l) {
    sl_enc_layer_saved& sv = saved[l];
    sv.x = x;
    if (c->skip[l]) continue;              // LayerDrop: identity in both directions
    const sl_hubert_layer& L = layers[l];
    const uint64_t* sd = c->seeds + 4 * (size_t)l;
    SL_CHECK_ARG(sv.ln1 && sv.qkv && sv.att && sv.lse && sv.x_mid && sv.ln2 && sv.pre1 && sv.mid && sv.x_out, "sl_encoder_stack_train_fwd: layer %d lacks a saved buffer", l);
    SL_TRY(sl_layernorm(x, sv.ln1, L.ln1_g, L.ln1_b, n, H, c->ln_eps, 0, dt, stream));
    SL_TRY(gemm(dt, sv.ln1, H, L.wqkv, H, sv.qkv, 3 * H, L.bqkv, nullptr, 0, n, 3 * H, H, SL_ACT_NONE, nullptr, st));
    SL_TRY(attn_fwd(dt, sv.qkv, 3 * H, sv.att, sv.lse, c->cu, c->klen, c->nseq, c->max_len, nh, nh, 64, 0, 0.125f, c->p_attn, sd[0], st));
    const bool fuse = fuse_ok(n, dt, H, F);
    Post pd;                                // h = residual + dropout(sublayer(h)): the mask in the producing GEMM's epilogue
    pd.op = SL_POST_DROPOUT; pd.p = c->p_hidden; pd.ld = H;
    if (c->p_hidden > 0.f && fuse) {       // h = residual + dropout(attention(layer_norm(h)))
      pd.seed = sd[1];
      SL_TRY(gemm(dt, sv.att, H, L.wo, H, sv.x_mid, H, L.bo, x, H, n, H, H, SL_ACT_NONE, nullptr, st, nullptr, &pd));
    } else if (c->p_hidden > 0.f) {
      SL_TRY(gemm(dt, sv.att, H, L.wo, H, w.tmp_h[0], H, L.bo, nullptr, 0, n, H, H, SL_ACT_NONE, nullptr, st));
      SL_TRY(sl_dropout(w.tmp_h[0], x, sv.x_mid, n * H, c->p_hidden, sd[1], dt, stream));
    } else {
      SL_TRY(gemm(dt, sv.att, H, L.wo, H, sv.x_mid, H, L.bo, x, H, n, H, H, SL_ACT_NONE, nullptr, st));
    }
    SL_TRY(sl_layernorm(sv.x_mid, sv.ln2, L.ln2_g, L.ln2_b, n, H, c->ln_eps, 0, dt, stream));
    if (c->p_act > 0.f && fuse) {          // mid = dropout(gelu(pre1)), pre1 kept
      Post pa;
      pa.op = SL_POST_DROPOUT; pa.p = c->p_act; pa.seed = sd[2]; pa.ld = F;
      SL_TRY(gemm(dt, sv.ln2, H, L.w1, H, sv.mid, F, L.b1, nullptr, 0, n, F, H, SL_ACT_GELU, sv.pre1, st, nullptr, &pa));
    } else {
      SL_TRY(gemm(dt, sv.ln2, H, L.w1, H, sv.mid, F, L.b1, nullptr, 0, n, F, H, SL_ACT_GELU, sv.pre1, st));
      if (c->p_act > 0.f) SL_TRY(sl_dropout(sv.mid, nullptr, sv.mid, n * F, c->p_act, sd[2], dt, stream));
    }
    if (c->p_hidden > 0.f && fuse) {       // h = h + output_dropout(output_dense(...))
      pd.seed = sd[3];
      SL_TRY(gemm(dt, sv.mid, F, L.w2, F, sv.x_out, H, L.b2, sv.x_mid, H, n, H, F, SL_ACT_NONE, nullptr, st, w.sk, &pd));
    } else if (c->p_hidden > 0.f) {
      SL_TRY(gemm(dt, sv.mid, F, L.w2, F, w.tmp_h[0], H, L.b2, nullptr, 0, n, H, F, SL_ACT_NONE, nullptr, st, w.sk));
      SL_TRY(sl_dropout(w.tmp_h[0], sv.x_mid, sv.x_out, n * H, c->p_hidden, sd[3], dt, stream));
    } else {
      SL_TRY(gemm(dt, sv.mid, F, L.w2, F, sv.x_out, H, L.b2, sv.x_mid, H, n, H, F, SL_ACT_NONE, nullptr, st, w.sk));
    }
    x = sv.x_out;
  }
  *x_out = x;
  return 0;
}

extern "C" int sl_encoder_stack_train_bwd(const sl_hubert_layer* layers, const sl_enc_stack_cfg* c, const sl_enc_layer_saved* saved,
                                          const sl_enc_layer_grads* grads, int32_t layer_begin, int32_t layer_end, void* dx, void* workspace,
                                          size_t workspace_bytes, sl_stream stream) {
  SL_CHECK_ARG(layers && c && saved && grads && dx && workspace && c->cu && c->klen && c->seeds && c->skip, "sl_encoder_stack_train_bwd: null pointer");
  SL_CHECK_ARG(0 <= layer_begin && layer_begin <= layer_end && layer_end <= c->n_layers, "sl_encoder_stack_train_bwd: bad layer range");
  EncWs w;
  SL_CHECK_ARG(enc_carve(c, workspace, workspace_bytes, w) <= workspace_bytes, "sl_encoder_stack_train_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int dt = c->dtype, H = c->hidden, F = c->ffn, nh = c->n_heads;
  const int64_t n = c->n_tok;
  const size_t sz = sl_dtype_size(dt);
  SideStream ss;
  SL_TRY(ss.init(st, n));
  hipStream_t sw = st;           // where the current parameter-gradient group runs
  void* sk = w.sk;                                             // split-K workspace of the caller's stream (short windows only: NULL from 2 048 rows)
  if (sk) SL_HIP(hipMemsetAsync(sk, 0, 1024, st));
  // arrival counter of the LayerNorm backward's in-kernel column reduce: the last 256 bytes of its record workspace (sl_layernorm_bwd_ws_bytes), zero
  // whenever no launch is in flight (a call cut short must not poison this one)
  int32_t* const ln_cnt = w.ln_ws_bytes >= 256 ? (int32_t*)((unsigned char*)w.ln_ws + w.ln_ws_bytes - 256) : nullptr;
  if (ln_cnt) SL_HIP(hipMemsetAsync(ln_cnt, 0, 256, st));
  void* sk_w = w.sk_w;                                         // the parameter-gradient group's own
  if (sk_w && sk_w != sk) SL_HIP(hipMemsetAsync(sk_w, 0, 1024, st));      // ordered before the first fork of the side stream
  // tmp_h[par] already holds dropout(dx) under the CURRENT layer's output-dropout mask: the LayerNorm backward that produced dx (the layer
  // above's) wrote it as its second output, so this layer starts without an sl_dropout launch and without re-reading dx
  bool have_drop = false;
  const bool fuse = fuse_ok(n, dt, H, F);
  // transposed weights made beside the forward on this workspace (WtCache): wait for that launch once, then no transposes in this call
  WtCache* wc = nullptr;
  SL_TRY(wt_cache(wc));
  const bool wt_ahead = w.wt_all && wc->valid && wc->ws == workspace && wc->layers == (const void*)layers && wc->n_layers == c->n_layers && wc->H == H &&
                        wc->F == F && wc->dt == dt;
  if (wt_ahead && !wc->waited) {
    SL_HIP(hipStreamWaitEvent(st, wc->done, 0));
    wc->waited = true;
  }
  auto WT = [&](int l, int which) -> const void* { return wt_ahead ? w.wt(l, which, H, F, sz) : nullptr; };
  int par = 0;                   // which copy of the group-read buffers this layer uses (alternates over the layers visited)
  for (int l = layer_end - 1; l >= layer_begin; --l) {
    if (c->skip[l]) continue;
    const sl_hubert_layer& L = layers[l];
    const sl_enc_layer_saved& sv = saved[l];
    const sl_enc_layer_grads& g = grads[l];
    const uint64_t* sd = c->seeds + 4 * (size_t)l;
    void* const tmp_h = w.tmp_h[par];
    void* const tmp_h2 = w.tmp_h2[par];
    void* const d_pre1 = w.d_pre1[par];
    void* const d_h2 = w.d_h2[par];
    void* const d_qkv = w.d_qkv[par];
    // ---- feed-forward half: x_out = x_mid + drop(w2 . drop_act(gelu(w1 . ln2(x_mid) + b1)) + b2)
    const void* d_o2 = dx;
    if (c->p_hidden > 0.f) {
      if (!have_drop) SL_TRY(sl_dropout(dx, nullptr, tmp_h, n * H, c->p_hidden, sd[3], dt, stream));      // (first layer of the call: nothing reads tmp_h[0] yet)
      d_o2 = tmp_h;
    }
    have_drop = false;
    if (fuse) {
      // d_pre1 = gelu'(pre1) * drop_act(d_o2 . w2) and b1 += colsum(d_pre1), all in the data-gradient product's epilogue
      Post pg;
      pg.op = SL_POST_GELU_BWD; pg.p = c->p_act; pg.seed = sd[2]; pg.ld = F; pg.in = sv.pre1; pg.in_ld = F; pg.colsum = g.b1;
      SL_TRY(dgrad(dt, d_o2, H, L.w2, H, F, WT(l, 0), d_pre1, F, n, w.s, st, nullptr, &pg));
    } else {
      SL_TRY(dgrad(dt, d_o2, H, L.w2, H, F, WT(l, 0), w.d_mid, F, n, w.s, st));
      if (c->p_act > 0.f) SL_TRY(sl_dropout(w.d_mid, nullptr, w.d_mid, n * F, c->p_act, sd[2], dt, stream));
      SL_TRY(sl_gelu_bwd(w.d_mid, sv.pre1, d_pre1, n * F, dt, stream));
    }
    // group A (one fork): the two feed-forward parameter gradients
    SL_TRY(ss.begin(sw));
    SL_TRY(wgrad_acc(dt, d_o2, H, H, sv.mid, F, F, g.w2, n, w.s, sw, sk_w, g.b2));
    SL_TRY(wgrad_acc(dt, d_pre1, F, F, sv.ln2, H, H, g.w1, n, w.s, sw, sk_w, fuse ? nullptr : g.b1));      // (fused: b1 came out of the data-gradient epilogue above)
    if (c->p_hidden <= 0.f) SL_TRY(ss.end(2));             // (group A reads dx itself then: it must be through before dx is rewritten below)
    int32_t S1 = 0, S2 = 0;      // K runs left in `sk` for the LayerNorm backward behind the product (0: d_h1 was written as usual)
    SL_TRY(dgrad(dt, d_pre1, F, L.w1, F, H, WT(l, 1), w.d_h1, H, n, w.s, st, sk, nullptr, &S1));
    // d_h2 = d x_mid: the LayerNorm path + the residual path (dx), one pass — and dropout(d_h2), the out-projection's incoming gradient
    const bool drop1 = c->p_hidden > 0.f && fuse;
    SL_TRY(sl_layernorm_bwd_ws_add_impl(sv.x_mid, L.ln2_g, L.ln2_b, w.d_h1, dx, d_h2, g.ln2_g, g.ln2_b, n, H, c->ln_eps, dt, w.ln_ws, w.ln_ws_bytes, stream,
                                        drop1 ? tmp_h2 : nullptr, c->p_hidden, sd[1], S1 ? (const float*)((const unsigned char*)sk + 1024) : nullptr, S1, ln_cnt));
    // ---- attention half: x_mid = x + drop(wo . attn(qkv(ln1(x))) + bo)
    const void* d_o1 = d_h2;
    if (c->p_hidden > 0.f) {
      if (!drop1) SL_TRY(sl_dropout(d_h2, nullptr, tmp_h2, n * H, c->p_hidden, sd[1], dt, stream));
      d_o1 = tmp_h2;
    }
    SL_TRY(dgrad(dt, d_o1, H, L.wo, H, H, WT(l, 2), w.d_att, H, n, w.s, st));
    SL_TRY(attn_bwd(dt, sv.qkv, 3 * H, sv.att, w.d_att, sv.lse, w.delta, d_qkv, c->cu, c->klen, c->nseq, c->max_len, n, nh, nh, 64, 0, 0.125f,
                    c->p_attn, sd[0], st));
    // group B (one fork): the two attention parameter gradients; its completion event covers group A too (the side stream is in order)
    SL_TRY(ss.begin(sw));
    SL_TRY(wgrad_acc(dt, d_o1, H, H, sv.att, H, H, g.wo, n, w.s, sw, sk_w, g.bo));
    SL_TRY(wgrad_acc(dt, d_qkv, 3 * H, 3 * H, sv.ln1, H, H, g.wqkv, n, w.s, sw, sk_w, g.bqkv));
    SL_TRY(ss.end(par));
    SL_TRY(dgrad(dt, d_qkv, 3 * H, L.wqkv, 3 * H, H, WT(l, 3), w.d_h1, H, n, w.s, st, sk, nullptr, &S2));
    // The one wait of the layer: the PREVIOUS visited layer's groups (the other copy of the buffers) must be through before this layer's last
    // kernel writes that copy's tmp_h (the dropped gradient of the next layer down) — and with them everything older, so the next layer may
    // overwrite its own copy freely.  p_hidden = 0: group A read dx, which is rewritten here.
    SL_TRY(ss.join(1 - par));
    SL_TRY(ss.join(2));
    int below = -1;
    for (int k = l - 1; k >= layer_begin && below < 0; --k)
      if (!c->skip[k]) below = k;
    have_drop = below >= 0 && c->p_hidden > 0.f && fuse;
    // dx = d x: the LayerNorm path + the residual path (d_h2), one pass — and dropout(dx) under the mask of the next layer down this call visits
    SL_TRY(sl_layernorm_bwd_ws_add_impl(sv.x, L.ln1_g, L.ln1_b, w.d_h1, d_h2, dx, g.ln1_g, g.ln1_b, n, H, c->ln_eps, dt, w.ln_ws, w.ln_ws_bytes, stream,
                                        have_drop ? w.tmp_h[1 - par] : nullptr, c->p_hidden, have_drop ? c->seeds[4 * (size_t)below + 3] : 0,
                                        S2 ? (const float*)((const unsigned char*)sk + 1024) : nullptr, S2, ln_cnt));
    par ^= 1;
  }
  for (int k = 0; k < 4; ++k) SL_TRY(ss.join(k));          // the caller's stream owns the gradients again
  return 0;
}

// ================================================================================================
// frozen Llama decoder over packed sequences (hf:models/llama/modeling_llama.py:284-324): forward with a tape and the
// data-gradient backward (its weights have requires_grad = False, ref:trainer.py:63-64)
// ================================================================================================
struct LlamaTrainWs {
  void *h, *mid, *qkv, *x2, *gu, *att, *d_mid, *d_gu, *d_h, *dx2, *d_att, *d_qkv;
  float *lse, *delta;
  void* sk;       // stream-K workspace of the data-gradient products (flags zeroed at the start of every backward call)
  BwdScratch s;
};

static size_t llama_train_carve(const sl_llama_stack_cfg* c, void* base, size_t cap, LlamaTrainWs& w) {
  const size_t sz = sl_dtype_size(c->dtype);
  const int64_t n = c->n_tok, H = c->hidden, F = c->ffn;
  const int64_t qkv_w = (int64_t)(c->n_heads + 2 * c->n_kv_heads) * c->head_dim, att_w = (int64_t)c->n_heads * c->head_dim;
  Carver cv(base, cap);
  w.h = cv.take(n * H * sz);
  w.mid = cv.take(n * F * sz);
  w.qkv = cv.take(n * qkv_w * sz);
  w.x2 = cv.take(n * H * sz);
  w.gu = cv.take(n * 2 * F * sz);
  w.att = cv.take(n * att_w * sz);
  w.d_mid = cv.take(n * F * sz);
  w.d_gu = cv.take(n * 2 * F * sz);
  w.d_h = cv.take(n * H * sz);
  w.dx2 = cv.take(n * H * sz);
  w.d_att = cv.take(n * att_w * sz);
  w.d_qkv = cv.take(n * qkv_w * sz);
  w.lse = (float*)cv.take(n * c->n_heads * sizeof(float));
  w.delta = (float*)cv.take(n * c->n_heads * sizeof(float));
  w.sk = cv.take(sl_gemm_streamk_workspace_bytes());
  w.s.yt = w.s.xt = w.s.wt = nullptr;    // data gradients only, on cached transposed weights
  return cv.off + 256;
}

extern "C" size_t sl_llama_stack_train_workspace_bytes(const sl_llama_stack_cfg* c) {
  LlamaTrainWs w;
  return llama_train_carve(c, nullptr, 0, w);
}

extern "C" int sl_llama_stack_train_fwd(const sl_llama_train_layer* layers, const sl_llama_stack_cfg* c, void* const* hidden,
                                        const sl_llama_layer_saved* saved, void* workspace, size_t workspace_bytes, sl_stream stream) {
  SL_CHECK_ARG(layers && c && hidden && workspace && c->cu && c->klen && c->pos && c->rope_cos && c->rope_sin, "sl_llama_stack_train_fwd: null pointer");
  SL_CHECK_ARG(c->head_dim == 128, "sl_llama_stack_train_fwd: head_dim %d not built (128)", c->head_dim);
  LlamaTrainWs w;
  SL_CHECK_ARG(llama_train_carve(c, workspace, workspace_bytes, w) <= workspace_bytes, "sl_llama_stack_train_fwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int dt = c->dtype, H = c->hidden, F = c->ffn, nh = c->n_heads, nkv = c->n_kv_heads, D = c->head_dim;
  const int qkv_w = (nh + 2 * nkv) * D, att_w = nh * D;
  const int64_t n = c->n_tok;
  const float scale = 1.0f / sqrtf((float)D);
  SL_HIP(hipMemsetAsync(w.sk, 0, 1024, st));        // the stream-K flags (gemm.hip): zero whenever no launch is in flight
  for (int l = 0; l < c->n_layers; ++l) {
    const sl_llama_train_layer& L = layers[l];
    const void* x = hidden[l];
    void* x_next = hidden[l + 1];
    SL_CHECK_ARG(x && x_next, "sl_llama_stack_train_fwd: hidden[%d] / hidden[%d] missing", l, l + 1);
    // saved == NULL: teacher pass, nothing is kept but the hidden states themselves
    void* qkv = saved ? saved[l].qkv : w.qkv;
    void* x2 = saved ? saved[l].x2 : w.x2;
    void* gu = saved ? saved[l].gu : w.gu;
    void* att = saved ? saved[l].att : w.att;
    float* lse = saved ? saved[l].lse : nullptr;
    SL_CHECK_ARG(qkv && x2 && gu && att && (!saved || lse), "sl_llama_stack_train_fwd: layer %d lacks a saved buffer", l);
    SL_TRY(sl_rmsnorm(x, w.h, L.norm1, n, H, c->rms_eps, dt, stream));
    // w.sk: products of few tiles (the per-rank window of a data-parallel run: a few hundred rows) are cut along K (gemm.hip splitk_runs)
    SL_TRY(gemm(dt, w.h, H, L.wqkv, H, qkv, qkv_w, nullptr, nullptr, 0, n, qkv_w, H, SL_ACT_NONE, nullptr, st, w.sk));
    SL_TRY(sl_rope_inplace(qkv, c->pos, c->rope_cos, c->rope_sin, n, nh + 2 * nkv, nh + nkv, D, 0, dt, stream));
    SL_TRY(attn_fwd(dt, qkv, qkv_w, att, lse, c->cu, c->klen, c->nseq, c->max_len, nh, nkv, D, 1, scale, 0.f, 0, st));
    SL_TRY(gemm(dt, att, att_w, L.wo, att_w, x2, H, nullptr, x, H, n, H, att_w, SL_ACT_NONE, nullptr, st, w.sk));
    SL_TRY(sl_rmsnorm(x2, w.h, L.norm2, n, H, c->rms_eps, dt, stream));
    SL_TRY(gemm(dt, w.h, H, L.wgu, H, gu, 2 * F, nullptr, nullptr, 0, n, 2 * F, H, SL_ACT_NONE, nullptr, st));   // interleaved gate/up pre-activations
    SL_TRY(sl_silu_mul(gu, w.mid, n, F, dt, stream));
    SL_TRY(gemm(dt, w.mid, F, L.wdown, F, x_next, H, nullptr, x2, H, n, H, F, SL_ACT_NONE, nullptr, st, w.sk));
  }
  return 0;
}

extern "C" int sl_llama_stack_train_bwd(const sl_llama_train_layer* layers, const sl_llama_stack_cfg* c, void* const* hidden,
                                        const sl_llama_layer_saved* saved, void* const* d_tap, void* dx, void* workspace, size_t workspace_bytes,
                                        sl_stream stream) {
  SL_CHECK_ARG(layers && c && hidden && saved && dx && workspace && c->cu && c->klen && c->pos && c->rope_cos && c->rope_sin,
               "sl_llama_stack_train_bwd: null pointer");
  SL_CHECK_ARG(c->head_dim == 128, "sl_llama_stack_train_bwd: head_dim %d not built (128)", c->head_dim);
  LlamaTrainWs w;
  SL_CHECK_ARG(llama_train_carve(c, workspace, workspace_bytes, w) <= workspace_bytes, "sl_llama_stack_train_bwd: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int dt = c->dtype, H = c->hidden, F = c->ffn, nh = c->n_heads, nkv = c->n_kv_heads, D = c->head_dim;
  const int qkv_w = (nh + 2 * nkv) * D, att_w = nh * D;
  const int64_t n = c->n_tok;
  const float scale = 1.0f / sqrtf((float)D);
  SL_HIP(hipMemsetAsync(w.sk, 0, 1024, st));        // the stream-K flags (a run of an earlier call cut short must not poison this one)
  const bool fuse = fuse_ok(n, dt, H, F);
  for (int l = c->n_layers - 1; l >= 0; --l) {
    const sl_llama_train_layer& L = layers[l];
    const sl_llama_layer_saved& sv = saved[l];
    SL_CHECK_ARG(L.wqkv_t && L.wo_t && L.wgu_t && L.wdown_t, "sl_llama_stack_train_bwd: layer %d lacks the transposed weights", l);
    // x3 = x2 + wdown . (silu(gate) * up),   [gate | up] = wgu . rmsnorm(x2)
    if (fuse) {      // d [gate | up] written by the down projection's data-gradient product itself (SL_POST_SILU_MUL_BWD)
      Post ps;
      ps.op = SL_POST_SILU_MUL_BWD; ps.in = sv.gu; ps.in_ld = 2 * F;
      SL_TRY(dgrad(dt, dx, H, L.wdown, H, F, L.wdown_t, w.d_gu, 2 * F, n, w.s, st, w.sk, &ps));
    } else {
      SL_TRY(dgrad(dt, dx, H, L.wdown, H, F, L.wdown_t, w.d_mid, F, n, w.s, st, w.sk));
      SL_TRY(sl_silu_mul_bwd(sv.gu, w.d_mid, w.d_gu, n, F, dt, stream));
    }
    int32_t S1 = 0, S2 = 0;      // K runs of the product in front of each RMSNorm backward, summed there while loading (0: written as usual)
    const float* parts = (const float*)((const unsigned char*)w.sk + 1024);
    SL_TRY(dgrad(dt, w.d_gu, 2 * F, L.wgu, 2 * F, H, L.wgu_t, w.d_h, H, n, w.s, st, w.sk, nullptr, &S1));
    SL_TRY(sl_rmsnorm_bwd_add_impl(sv.x2, L.norm2, w.d_h, dx, w.dx2, n, H, c->rms_eps, dt, stream, nullptr, S1 ? parts : nullptr, S1));      // + the residual path (dx), same pass
    // x2 = x + wo . attn(rope(wqkv . rmsnorm(x)))
    SL_TRY(dgrad(dt, w.dx2, H, L.wo, H, att_w, L.wo_t, w.d_att, att_w, n, w.s, st, w.sk));
    SL_TRY(attn_bwd(dt, sv.qkv, qkv_w, sv.att, w.d_att, sv.lse, w.delta, w.d_qkv, c->cu, c->klen, c->nseq, c->max_len, n, nh, nkv, D, 1, scale, 0.f, 0,
                    st));
    SL_TRY(sl_rope_inplace(w.d_qkv, c->pos, c->rope_cos, c->rope_sin, n, nh + 2 * nkv, nh + nkv, D, 1, dt, stream));
    SL_TRY(dgrad(dt, w.d_qkv, qkv_w, L.wqkv, qkv_w, H, L.wqkv_t, w.d_h, H, n, w.s, st, w.sk, nullptr, &S2));
    // + the residual path (dx2) and the feature-distillation gradient of hidden_states[l], same pass
    SL_TRY(sl_rmsnorm_bwd_add_impl(hidden[l], L.norm1, w.d_h, w.dx2, dx, n, H, c->rms_eps, dt, stream, (d_tap && fuse) ? d_tap[l] : nullptr, S2 ? parts : nullptr, S2));
    if (d_tap && d_tap[l] && !fuse) SL_TRY(sl_axpby(d_tap[l], dx, 1.f, 1.f, n * H, dt, stream));
  }
  return 0;
}
