/*
 * speechllm.h — C ABI of libspeechllm.so: the MI355X (gfx950) hot path of the speech-prompted LLM
 * pipeline (HuBERT encoder -> pooled projector -> Llama prefill + KV-cached greedy decode).
 *
 * The reference (wonjune-kang/llm-speech-summarization) has no native code and no FFI: its hot path
 * is Python calling HuggingFace `transformers` modules.  Each entry point below therefore cites the
 * reference / HF call it replaces (ref: = /root/reference, hf: = site-packages/transformers).  The
 * Python host mirror of the reference classes (the .py files of llm-speech-summarization_amd/) binds these with
 * ctypes; INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes only; no C++ / torch types cross this boundary.
 *   - Every device buffer (weights, activations, KV cache, workspace) is allocated and owned by the
 *     caller.  The library never allocates or frees device memory.  Per-op entry points (sl_gemm*, sl_attn_*,
 *     norms, losses, the KD tape stacks ...) only enqueue launches on `stream` and never synchronise.
 *     The whole-model entry points that take HOST descriptor arrays synchronise as follows:
 *       sl_hubert_forward, sl_whisper_forward, sl_llama_prefill — ONE hipStreamSynchronize(stream) after uploading the
 *         batch descriptors (row offsets / group records built from the host arrays), before the first compute launch;
 *       sl_greedy_generate — that of its prefill, one every `check_every` steps when EOS is enabled, and ONE at the end
 *         behind the asynchronous copy of the ids to the host (the call returns host results).
 *     Nothing synchronises the device or touches the null stream.
 *   - Return value: 0 on success, a negative sl_status otherwise; sl_last_error() gives the text.
 *   - dtype: SL_F32 (exact-fp32 MFMA, the parity mode) or SL_BF16 (bf16 storage, fp32 accumulate).
 *   - Layouts: activations row-major (tokens, channels); Linear weights in nn.Linear layout
 *     (out_features, in_features); conv weights re-laid out by the host as documented per call.
 */
#ifndef SPEECHLLM_H
#define SPEECHLLM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sl_stream; /* hipStream_t */

enum sl_dtype { SL_F32 = 0, SL_BF16 = 1 };
enum sl_status {
  SL_OK = 0,
  SL_ERR_ARG = -1,      /* bad shape / alignment / enum */
  SL_ERR_LAUNCH = -2,   /* HIP launch or runtime error */
  SL_ERR_UNSUPPORTED = -3
};
enum sl_act { SL_ACT_NONE = 0, SL_ACT_GELU = 1, SL_ACT_SILU_MUL = 2, SL_ACT_ROPE_KV = 3 };
enum sl_w_layout { SL_W_ROWMAJOR = 0, SL_W_PACKED = 1 };

const char* sl_last_error(void);       /* thread-local, never NULL */
#define SL_ABI_VERSION 7
int sl_version(void);                  /* == SL_ABI_VERSION of the header the library was built from; bumps on any signature change */
int sl_device_arch(char* buf, int n);  /* gcnArchName of the current device, e.g. "gfx950:sramecc+:xnack-" */
/* Tuning switches (SL_* environment variables, documented in csrc/common.h) are read once, at first use; tools that change
 * them inside one process call this to re-read them.  Not needed in normal operation. */
int sl_tuning_reload(void);

/* ---------------------------------------------------------------------------------------------
 * GEMM  C[b] = act(A[b] . W[b]^T + bias[b]) + residual[b]        (TN: both operands K-contiguous)
 * Replaces every nn.Linear / F.linear on the path (hf:models/hubert/modeling_hubert.py:229,
 * 293-300,340,360-365; hf:models/llama/modeling_llama.py:174-176,254-256,280; lm_head
 * ref:model/audio_llama.py:67) and, through lda < K (overlapping rows), the strided Conv1d layers
 * 1..6 of the HuBERT feature extractor (hf:...hubert.py:141) and the grouped positional conv
 * (hf:...hubert.py:47-53) as implicit GEMMs without an im2col buffer.
 *   A: (M, K) row stride lda;  W: (N, K) row stride ldw;  C: (M, Nout) row stride ldc
 *   act == SL_ACT_SILU_MUL: W rows come in blocks of 16 gate rows followed by 16 up rows;
 *     Nout = N/2, C[m][16p+c] = silu(gate) * up  (hf:...llama.py:175).  Otherwise Nout = N.
 *   bias (N) and residual (M, Nout; row stride ldr) may be NULL.  Order: +bias, act, +residual.
 *   out_f32 != 0 writes C as float regardless of dtype (logits).
 *   Requirements: K % 8 == 0 (bf16) / K % 4 == 0 (f32); A, W 16-byte aligned rows.
 *   M <= 64 with row-major weights dispatches to the weight-streaming (HBM-bound) skinny kernel; packed weights
 *   (sl_pack_weight) run the skinny kernel up to 32 rows and the LDS-staged streaming kernel (gemm_stream.hip) above.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* A; int64_t lda; int64_t strideA;
  const void* W; int64_t ldw; int64_t strideW;
  void* C; int64_t ldc; int64_t strideC;
  const void* bias; int64_t strideBias;
  const void* residual; int64_t ldr; int64_t strideR;
  int32_t M, N, K, batch;
  int32_t dtype, act, out_f32, w_layout;
} sl_gemm_args;
int sl_gemm(const sl_gemm_args* a, sl_stream stream);

/* Fragment-major weight packing for the decode weight-streaming kernels:
 *   dst[f][s][lane][e] = src[16 f + (lane & 15)][KSTEP s + VEC (lane >> 4) + e],  VEC = 8 (bf16) / 4 (f32),
 *   KSTEP = 4 VEC, f < ceil(N/16) (rows past N are zero), so one wave-level weight load is 1 KiB contiguous
 *   (+25..45 % HBM throughput over 16 x 64-byte row segments, tools/tune_skinny.hip).  K %% KSTEP == 0.
 *   dst holds ceil(N/16)*16*K elements.  Pass it to sl_gemm with w_layout = SL_W_PACKED. */
int sl_pack_weight(const void* src, int64_t ld_src, void* dst, int32_t N, int32_t K, int32_t dtype, sl_stream stream);

/* Decode-only fusions on top of sl_gemm (packed weights; M <= 16 rows run the skinny kernel, more rows the
 * LDS-staged streaming kernel of gemm_stream.hip):
 *   fuse_rms  — the consuming Linear absorbs the preceding LlamaRMSNorm (hf:...llama.py:62-67): the
 *               gain is pre-multiplied into W on the host and out[m] *= rsqrt(mean(x[m]^2) + rms_eps),
 *               with the mean taken from the activation fragments the kernel streams anyway;
 *   act == SL_ACT_ROPE_KV — N = (n_heads + 2 n_kv) * 128 fused q|k|v rows whose q/k heads are stored
 *               in 16-row blocks alternating the two rotate_half halves ([0:16],[64:80],[16:32],...):
 *               the epilogue applies RoPE (hf:...llama.py:130-160), writes q to C (row stride ldc,
 *               natural head layout) and k, v straight into the cache at [tok_seq][head][tok_pos]
 *               (replaces sl_rope_kv_append + the qkv round trip). */
typedef struct {
  int32_t fuse_rms; float rms_eps;
  const float* rope_cos; const float* rope_sin;
  const int32_t* tok_pos; const int32_t* tok_seq;
  void* k_cache; void* v_cache;
  int32_t n_heads, n_kv_heads, max_ctx, reserved;
  /* optional scratch for K-split partial sums (M > 16 rows against few weight rows cannot fill 256 CUs
   * otherwise): sl_gemm_split_workspace_bytes(M, N, K, dtype) bytes, or NULL to disable the split.
   * Its first 8192 bytes are arrival counters for the optional in-kernel reduce (SL_STREAM_FIXUP=1: above 384 rows the
   * block that finishes a tile last sums the partial records itself): zero them once when the buffer is created; every
   * call leaves them zero again, in stream order.  One buffer per call chain — two streams must not share it. */
  void* split_ws; size_t split_ws_bytes;
  /* RMSNorm statistics handed from one GEMM to the next (streaming path only):
   *   rstd_out — with the plain epilogue and a K-split, the reduce kernel also writes, per output row,
   *              rsqrt(mean(row^2) + rms_eps) of the values it stores (N <= 4096);
   *   rstd_in  — with fuse_rms, multiply by rstd_in[m] instead of recomputing the statistics of A's rows. */
  const float* rstd_in; float* rstd_out;
  /* with rstd_out: the reduce pass also writes the RMS-normalised rows themselves (hf:models/llama/modeling_llama.py:60-71:
   * weight * (x * rsqrt(mean(x^2) + eps)).to(dtype)) to norm_out (row stride N, the output's type), gain = norm_gain (N values) —
   * for a consumer that runs on the row-major tiled kernels and therefore wants its input normalised (decode steps above ~900 rows:
   * gate/up on the 256 x 256 tiles).  Both NULL: off. */
  void* norm_out; const void* norm_gain;
} sl_gemm_fused;
size_t sl_gemm_split_workspace_bytes(int32_t M, int32_t N, int32_t K, int32_t dtype);
/* K splits sl_gemm_fused_decode will use for this shape when split_ws is supplied (1: no reduce pass, so no rstd_out) */
int32_t sl_gemm_split_count(int32_t M, int32_t N, int32_t K, int32_t dtype);
int sl_gemm_fused_decode(const sl_gemm_args* a, const sl_gemm_fused* fx, sl_stream stream);

/* Backward-pass forms of sl_gemm (training; always the tiled MFMA kernel):
 *   trans_a: A is stored as (K, M) with row stride lda  — C = A^T-stored . W^T   (e.g. dW = dY^T X)
 *   trans_w: W is stored as (K, N) with row stride ldw  — C = A . W-stored       (e.g. dX = dY W)
 *            transposed operands are read in 16-byte chunks along the OUTPUT index: lda/ldw must be
 *            multiples of 8 (bf16) / 4 (f32) and each stored row readable up to that multiple.
 *   aux_out: also store the pre-activation (after bias, before act) — what GELU's backward needs.
 *   residual_f32: with out_f32, the residual is float: C = residual + A.W^T accumulates fp32 gradients.
 *   groups / w_mod: see below. */
typedef struct {
  int32_t trans_a, trans_w, residual_f32, w_mod;
  void* aux_out;
  /* grouped (ragged) batch: device array of `batch` records {M, a_off, c_off, r_off} (int64, element offsets).
   * Batch z computes its own M rows from A + a_off into C + c_off (residual + r_off); W and bias are taken
   * at index z % w_mod (strideW / strideBias).  args->M is the LARGEST group (sizes the grid).  One launch
   * then covers e.g. a conv layer of every utterance of a ragged batch (hf:...hubert.py:141). */
  const int64_t* groups;
  /* groups_ext != 0: records are 8 int64 wide, {M, a_off, c_off, r_off, w_off, N, K, 0}: every group also has its own
   * weight offset, column count and reduction length (args->N / args->K are then the LARGEST ones).  One launch covers
   * e.g. one product of the attention backward over every (sequence, head) of a ragged batch.
   * groups_ext == 2: as 1, and the caller vouches that every group's K is a multiple of 64 bf16 / 32 f32 elements
   * (whole 128-byte slabs), which admits the LDS-DMA tiled kernel. */
  int32_t groups_ext, reserved;
  /* Fused row-wise top-1 in place of the store (greedy decode: lm_head + argmax, hf:generation/utils.py:2911-2925 over the logits
   * of ref:model/audio_llama.py:67): when both are set, C is NOT written; instead, for every row m and every group g of 64
   * output columns, amax_val[g * M + m] = max over columns [64 g, 64 g + 64) of (A.W^T + bias)[m][.] and amax_idx[g * M + m] =
   * its column (the lowest one on a tie; -inf / 0x7fffffff when every value is NaN).  ceil(N / 64) * M entries each.
   * Plain epilogue only (act NONE, no residual), one un-grouped row-major product, M > 64.  sl_greedy_select_partial finishes
   * the argmax over the groups: together they give bit for bit the token sl_gemm(out_f32) + sl_greedy_select give. */
  float* amax_val; int32_t* amax_idx;
  /* LayerNorm folded into the Linears around it (hf:models/hubert/modeling_hubert.py:515-517,612 stable-LN layer: x -> LN -> Linear;
   * bf16, plain row-major un-grouped products with N % 64 == 0, any row count (a product carrying these fields stays on the tiled
   * kernels even below 65 rows, so a short utterance alone gets the bits it gets inside a batch) — sl_gemm_ln_fold_ok()):
   *   producer side, stats_out: besides storing C, write for every row m and 64-column segment s the pair {sum, sum of squares} of
   *     the values AS STORED (after bias / act / residual, rounded to the output type) at stats_out[(m * (N / 64) + s) * 2];
   *     sl_layernorm_stats_finalize turns the N / 64 pairs of a row into {mean, rstd};
   *   consumer side, ln_mr / ln_u / ln_c: A holds the un-normalised rows x; W is the Linear's weight with the LayerNorm gain
   *     multiplied into its columns, ln_u[n] = sum_k W[n][k] (of the stored, rounded W), ln_c[n] = sum_k W0[n][k] beta[k] + bias[n];
   *     with {mean, rstd} = ln_mr[2 m], ln_mr[2 m + 1]:  out[m][n] = rstd * ((x W^T)[m][n] - mean * ln_u[n]) + ln_c[n], then act /
   *     residual as usual (bias must be NULL: it is inside ln_c).  Equals Linear(LayerNorm(x)) without the LayerNorm pass over x. */
  const float* ln_mr; const float* ln_u; const float* ln_c; float* stats_out;
  /* Stream-K workspace (optional).  Products whose 256 x 256 output tiles do not fill the chip evenly — the KD step's windows of a few
   * thousand rows (ref:trainer.py:270-384 at grad_accum_interval 16), its per-rank share under data parallelism, weight gradients of a
   * few dozen tiles under a long reduction — are cut along K as well: every CU takes an equal run of (tile, K slab) units, tiles
   * shared by two or three CUs are summed through this workspace inside the launch (fp32, fixed order: reproducible, not bit-equal
   * to the unsplit product).  The same workspace carries the PLAIN K runs of products of few tiles (S batched runs of whole slabs + a
   * fixed-order reduce launch; round 6: at most one 128 x 128 tile per CU runs on the ring form of the tile kernel, and a 256-tile
   * product that fills 50-66 % of one round is cut into three runs of uneven length).  Caller-owned, sl_gemm_streamk_workspace_bytes()
   * bytes (1 KiB of flags + 128 MiB of fp32 partial tiles), 16-byte aligned, ZEROED ONCE before its first
   * use and not shared by launches that may run concurrently (one per stream).  NULL, a shape the rule does not take, transposed /
   * grouped operands or the ln_* / stats_out / amax_* / aux_out forms: the product runs one block per tile as without it. */
  void* sk_ws; size_t sk_ws_bytes;
  /* Training-tape epilogue fusions (ABI 7; ref:trainer.py:270-384 runs these as separate torch ops around every Linear).  Plain
   * un-grouped row-major products on the tiled kernels only (no trans_*, groups, ln_*, stats_out, amax_*); all in the epilogue, with
   * the roundings of the unfused launch sequence reproduced (bf16: the value is rounded to the storage type where the unfused
   * sequence stored it), so fused and unfused tapes give the same bits.
   *   post_op = SL_POST_DROPOUT:  C = residual + dropout(act(A.W^T + bias))      (hf HubertEncoderLayerStableLayerNorm: h = h + dropout(sublayer))
   *             keep(i) as sl_dropout with seed drop_seed at element index i = row * drop_ld + col, kept values scaled by 1 / (1 - drop_p);
   *             aux_out (the pre-activation) is stored before act as usual.
   *   post_op = SL_POST_GELU_BWD: C = gelu'(post_in) * dropout(A.W^T)            (d pre = gelu'(pre) . d mid: sl_dropout + sl_gelu_bwd behind the
   *             data-gradient product; post_in = the saved pre-activation, (M, N) with row stride post_ld; drop_p = 0: no mask)
   *   post_op = SL_POST_SILU_MUL_BWD: C (M, 2 N) in the interleaved [16 gate | 16 up] layout = sl_silu_mul_bwd(post_in = gu (M, 2 N) with row
   *             stride post_ld, d mid = A.W^T); ldc is the row stride of the 2 N-wide output.
   *   colsum_out (any post_op incl. NONE): fp32 (N) += column sums of the values AS STORED in C (bias gradients: db = colsum(dY)),
   *             float atomics, a few dozen adders per column. */
  int32_t post_op, post_reserved;
  float drop_p; float post_reserved_f;
  uint64_t drop_seed;
  int64_t drop_ld;
  const void* post_in; int64_t post_ld;
  float* colsum_out;
  /* deferred_splits (host pointer, optional; plain un-grouped product with sk_ws, no bias / residual / post_op, output in the storage type): if the
   * product is cut into K runs (the few-tile rule of sk_ws), the reduce pass is NOT launched: *deferred_splits = S >= 2 and the S fp32 partial
   * products stay in sk_ws at byte offset 1024, S slabs of M x N floats (row stride N) — for a consumer that sums them itself while loading
   * (the norm-backward kernels of the training tapes: C = round(sum over runs, in run order) is what the reduce pass would have stored, and C is
   * NOT written).  Otherwise *deferred_splits = 0 and C is written as usual.  The partials live until the next product that uses sk_ws. */
  int32_t* deferred_splits;
} sl_gemm_ex_args;
enum { SL_POST_NONE = 0, SL_POST_DROPOUT = 1, SL_POST_GELU_BWD = 2, SL_POST_SILU_MUL_BWD = 3 };
int sl_gemm_ex(const sl_gemm_args* a, const sl_gemm_ex_args* ex, sl_stream stream);
size_t sl_gemm_streamk_workspace_bytes(void);
/* 1 when sl_gemm_ex takes ln_* / stats_out for a plain (M, N, K) product of this dtype, else 0 */
int32_t sl_gemm_ln_fold_ok(int32_t M, int32_t N, int32_t K, int32_t dtype);
/* The weight side of that fold for one Linear W (N, K) behind LayerNorm(gain, beta) (hf:models/hubert/modeling_hubert.py:515-517,612):
 * Wf = W with the gain multiplied into its columns, rounded to the storage type once; u[n] = sum_k Wf[n][k] (of the rounded values);
 * c[n] = sum_k W[n][k] beta[k] + bias[n] (bias may be NULL) — the ln_u / ln_c vectors and the weight a consumer GEMM takes. */
int sl_layernorm_fold_build(const void* W, const void* gain, const void* beta, const void* bias, void* Wf, float* u, float* c, int32_t N,
                            int32_t K, int32_t dtype, sl_stream stream);
/* {mean, rstd} per row from the per-segment {sum, sum of squares} a stats_out GEMM left ([rows][segs][2] floats, cols = 64 * segs
 * elements per row): mr[2 m] = mean, mr[2 m + 1] = rsqrt(max(E[x^2] - mean^2, 0) + eps); segments summed in order. */
int sl_layernorm_stats_finalize(const float* stats, int32_t segs, int64_t rows, int32_t cols, float eps, float* mr, sl_stream stream);
/* The same pair straight from the rows of x (rows, cols) — two-pass mean / variance, one wave per row — for a row set no GEMM
 * has just produced (the first layer's input). */
int sl_layernorm_stats(const void* x, int64_t rows, int32_t cols, float eps, float* mr, int32_t dtype, sl_stream stream);

/* LayerNorm over the last dim, optional fused GELU (conv layers: hf:...hubert.py:144-150;
 * encoder LNs hf:...hubert.py:515,517,612; feature projection :226).  In-place allowed. */
int sl_layernorm(const void* x, void* y, const void* gamma, const void* beta, int64_t rows, int32_t cols,
                 float eps, int32_t gelu, int32_t dtype, sl_stream stream);

/* LlamaRMSNorm (hf:models/llama/modeling_llama.py:62-67): y = w * (x * rsqrt(mean(x^2)+eps)). */
int sl_rmsnorm(const void* x, void* y, const void* w, int64_t rows, int32_t cols, float eps, int32_t dtype,
               sl_stream stream);

/* HuBERT conv layer 0: Conv1d(1 -> C, k, stride) + LayerNorm(C) + GELU fused, channel-last output
 * (hf:...hubert.py:141-150 for layer_id 0).  wave: (n_samples) float32 (always fp32 on input:
 * ref:inference.py:166);  w: (C, k) float32;  out: (L, C) dtype, L = (n_samples-k)/stride+1.
 * C must be 64*{1,2,4,8}. */
int sl_hubert_conv0(const float* wave, int64_t n_samples, const float* w, const float* bias,
                    const float* gamma, const float* beta, void* out, int32_t C, int32_t k, int32_t stride,
                    float eps, int32_t dtype, sl_stream stream);

/* Positional-conv input staging: x (T, H) -> xg (groups, T + k, H/groups), zero halo of k/2 rows
 * on both sides, so each group's conv window is one contiguous GEMM row (hf:...hubert.py:47-53). */
int sl_posconv_stage(const void* x, void* xg, int64_t T, int32_t H, int32_t groups, int32_t k,
                     int32_t dtype, sl_stream stream);

/* AvgPool1d(kernel, stride) over time on channel-last rows (ref:model/audio_encoder.py:59-63):
 * y[p] = mean(x[p*stride .. p*stride+kernel)).  Also serves ctc_pool via explicit ranges
 * (ref:model/audio_encoder.py:78-82): ranges = int32 (P,2) device array or NULL. */
int sl_avgpool_rows(const void* x, void* y, int64_t T, int32_t H, int32_t kernel, int32_t stride,
                    const int32_t* ranges, int64_t P, int32_t dtype, sl_stream stream);

/* Whole-ragged-batch forms of the three front-end / head stages (one launch for every utterance of a packed batch; the
 * encoder runtime and the KD tape use these: a single 10 s clip is only ~1 300 waves of conv0 work).
 *   sl_hubert_conv0_batch: utterance u has samples [sample_offsets[u], sample_offsets[u+1]) of `waves` and writes rows
 *     [row_offsets[u], row_offsets[u+1]) of out (both int64 device arrays of n_utt + 1 entries); max_L = longest output.
 *   sl_posconv_stage_batch: frames cu[u] .. cu[u] + klen[u] of x; utterance u's staged (groups, T_u + k, H/groups) block
 *     starts (cu[u] + u k) * H elements into xg.
 *   sl_avgpool_batch: rec = int64 (n_utt, 4) device records {pooled rows P_u, element offset of the utterance's rows in y, -, -};
 *     frames cu[u] .. of x (ref:model/audio_encoder.py:59-63). */
int sl_hubert_conv0_batch(const float* waves, const int64_t* sample_offsets_dev, const int64_t* row_offsets_dev, int32_t n_utt,
                          int64_t max_L, const float* w, const float* bias, const float* gamma, const float* beta, void* out, int32_t C,
                          int32_t k, int32_t stride, float eps, int32_t dtype, sl_stream stream);
int sl_posconv_stage_batch(const void* x, void* xg, const int32_t* cu, const int32_t* klen, int32_t n_utt, int64_t max_T, int32_t H,
                           int32_t groups, int32_t k, int32_t dtype, sl_stream stream);
int sl_avgpool_batch(const void* x, void* y, const int32_t* cu, const int32_t* klen, const int64_t* rec, int32_t n_utt, int64_t max_P,
                     int32_t H, int32_t kernel, int32_t stride, int32_t dtype, sl_stream stream);

/* Embedding row gather (hf:...llama.py:380-381; ref:utils.py:63-64). ids int32 on device. */
int sl_embed_gather(const void* table, const int32_t* ids, void* out, int64_t n, int32_t cols,
                    int32_t dtype, sl_stream stream);

/* Flash-style attention forward, variable-length packed sequences.
 *   head_dim 64, non-causal: HuBERT (hf:...hubert.py:234-259);  head_dim 128, causal GQA: Llama
 *   prefill (hf:...llama.py:191-213).  softmax in fp32, scores scaled by `scale`.
 *   q/out: token t of sequence s is row cu_q[s] + t;   k/v: row cu_k[s] + t.
 *   element address = base + row*row_stride + head*head_stride (strides in elements).
 *   Causal: key j visible to query i iff j <= i + (klen - qlen). */
typedef struct {
  const void* q; int64_t q_row_stride, q_head_stride;
  const void* k; int64_t k_row_stride, k_head_stride;
  const void* v; int64_t v_row_stride, v_head_stride;
  void* out; int64_t o_row_stride, o_head_stride;
  const int32_t* cu_q;   /* (nseq+1) device */
  const int32_t* cu_k;   /* (nseq)   device: first k/v row of each sequence */
  const int32_t* klen;   /* (nseq)   device: number of keys of each sequence */
  int32_t nseq, max_qlen, n_heads, n_kv_heads, head_dim, causal, dtype, reserved;
  float scale;
  /* training mode (hf HubertAttention: dropout on the attention probabilities): probability (0 = off) and the seed of
   * the counter-based mask of sl_dropout; the element index of probability (query row t of the packed batch, head h,
   * key j of its sequence) is ((t * n_heads + h) << 16) | j, so sequences are limited to 65 536 keys in this mode.
   * The softmax normaliser is the undropped sum. */
  float dropout_p;
  uint64_t dropout_seed;
  /* training: when non-NULL receives the log-sum-exp of every (query row, head) of the packed batch — float
   * [n_tok_q][n_heads], natural log of sum_j exp(scale * q.k_j) over the visible keys (undropped) — which sl_attn_bwd
   * recomputes the probabilities from. */
  float* lse;
} sl_attn_args;
int sl_attn_fwd(const sl_attn_args* a, sl_stream stream);

/* Flash-style attention BACKWARD for the same packed layouts (the KD step's data gradients through the frozen Llama,
 * data + parameter gradients through the HuBERT / Whisper encoder; autograd of hf:...hubert.py:234-259 /
 * hf:...llama.py:191-213 in the reference, ref:trainer.py:373-374).  Given q, k, v, the forward output `out`, its
 * gradient `d_out` and the forward's `lse`, writes dq (q's layout: rows cu_q[s]+t), dk and dv (k / v's layout: rows
 * cu_k[s]+t, n_kv_heads heads — the query heads of a GQA group are summed inside the kernel).  The S x S probabilities are
 * recomputed per tile and never stored; no atomics (results are bitwise reproducible).  `delta` is a float
 * [n_tok_q][n_heads] workspace (rowsum(d_out * out), filled by the call).  dropout_p / dropout_seed must equal the forward's.
 * Strides in elements; dq rows / heads must start on 4-element boundaries. */
typedef struct {
  const void* q; int64_t q_row_stride, q_head_stride;
  const void* k; int64_t k_row_stride, k_head_stride;
  const void* v; int64_t v_row_stride, v_head_stride;
  const void* out; int64_t o_row_stride, o_head_stride;
  const void* d_out; int64_t do_row_stride, do_head_stride;
  void* dq; int64_t dq_row_stride, dq_head_stride;
  void* dk; int64_t dk_row_stride, dk_head_stride;
  void* dv; int64_t dv_row_stride, dv_head_stride;
  const float* lse;
  float* delta;
  const int32_t* cu_q;   /* (nseq+1) device */
  const int32_t* cu_k;   /* (nseq)   device */
  const int32_t* klen;   /* (nseq)   device */
  int64_t n_tok_q;       /* rows of q / out / d_out / dq (= cu_q[nseq]) */
  int32_t nseq, max_qlen, max_klen, n_heads, n_kv_heads, head_dim, causal, dtype;
  float scale;
  float dropout_p;
  uint64_t dropout_seed;
} sl_attn_bwd_args;
int sl_attn_bwd(const sl_attn_bwd_args* a, sl_stream stream);

/* Backward-side companion of sl_attn_args.dropout_p for the explicit-probability backward: for every score matrix
 * z = seq * n_kv + kv_head of a packed batch ((n_mat, smax, ld) buffers, dims[z] valid rows / columns, query head
 * kv_head * rep + r) writes p_dropped = keep * p / (1-p) and masks the fp32 d(probabilities) in place. */
int sl_attn_dropout_bwd(const void* p, void* p_dropped, float* d_p, int64_t n_mat, int32_t smax, const int32_t* dims, int32_t ld,
                        const int32_t* cu_q, int32_t n_heads, int32_t n_kv_heads, int32_t r, float dropout_p, uint64_t seed, int32_t dtype,
                        sl_stream stream);

/* RoPE (rotate_half form, hf:...llama.py:130-160) on the q and k slices of a fused QKV activation,
 * then KV-cache append (hf:cache_utils.py:127-145 restated as write-at-position).
 *   qkv: (n_tok, (n_heads + 2 n_kv) * D);  q is rotated in place;
 *   k,v -> cache (n_seq_slots, n_kv, max_ctx, D) at [tok_seq[t]][:, tok_pos[t]].
 *   cos/sin: (rope_len, D/2) float32 tables computed on the host exactly as HF does. */
int sl_rope_kv_append(void* qkv, void* k_cache, void* v_cache, const int32_t* tok_seq, const int32_t* tok_pos,
                      const float* cos, const float* sin, int64_t n_tok, int32_t n_heads, int32_t n_kv,
                      int32_t D, int32_t max_ctx, int32_t dtype, sl_stream stream);

/* One-token GQA attention against the KV cache (decode; hf:...llama.py:191-213 with q_len 1).
 *   q: row b of (B, n_heads*D) with row stride q_stride;  ctx_len[b] keys are attended (the
 *   current token must already be appended).  out: (B, n_heads*D).  K/V rows are read from
 *   cache (slots, n_kv, max_ctx, D). */
int sl_attn_decode(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache, void* out,
                   const int32_t* ctx_len, int32_t B, int32_t n_heads, int32_t n_kv, int32_t D,
                   int32_t max_ctx, float scale, int32_t dtype, sl_stream stream);

/* Flash-decoding form of sl_attn_decode: the context is split into 64-key blocks (grid = n_kv x B x
 * ceil(max_ctx/64)) whose fp32 partial (O, max, sum) records are merged by a second kernel, so batch 1
 * fills the chip too.  workspace: sl_attn_decode_workspace_bytes(B, n_heads, n_kv, max_ctx). */
size_t sl_attn_decode_workspace_bytes(int32_t B, int32_t n_heads, int32_t n_kv, int32_t max_ctx);
int sl_attn_decode_split(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache, void* out,
                         void* workspace, const int32_t* ctx_len, int32_t B, int32_t n_heads, int32_t n_kv, int32_t D,
                         int32_t max_ctx, float scale, int32_t dtype, sl_stream stream);

/* Greedy token selection (hf:generation/utils.py:2894,2925-2936): argmax over fp32 logits (lowest index
 * on ties), pad finished rows, EOS check, append to out_ids[b][gen_count[b]], advance gen_count and
 * ctx_len.  logits (B, V) float; eos_ids is a HOST array of at most 8 ids (passed by value to the
 * kernel).  Device state, all int32 (B): unfinished (1 until the row emits EOS), ctx_len, gen_count,
 * finish_len (gen_count at the row's EOS), next_ids (fed to the next embedding gather);
 * out_ids (B, max_new) int32. */
int sl_greedy_select(const float* logits, int32_t B, int32_t V, const int32_t* eos_ids_host, int32_t n_eos,
                     int32_t pad_id, int32_t use_eos, int32_t* unfinished, int32_t* ctx_len, int32_t* gen_count,
                     int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new, sl_stream stream);
/* sl_greedy_select over the per-group partial maxima of a fused lm_head (sl_gemm_ex_args.amax_val / amax_idx, n_groups =
 * ceil(V / 64), layout [group][B]): same token, same EOS / pad / finished-row bookkeeping, 1/64 of the bytes. */
int sl_greedy_select_partial(const float* amax_val, const int32_t* amax_idx, int32_t n_groups, int32_t B, const int32_t* eos_ids_host,
                             int32_t n_eos, int32_t pad_id, int32_t use_eos, int32_t advance_ctx, int32_t* unfinished, int32_t* ctx_len,
                             int32_t* gen_count, int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new, sl_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Knowledge-distillation step (ref:trainer.py:270-374): backward and loss kernels.  GEMM-shaped
 * backward work (dgrad, wgrad, attention products) runs on sl_gemm_ex.
 * ------------------------------------------------------------------------------------------- */
/* dx = dy * gelu'(pre)  (pre = aux_out of the forward GEMM). */
int sl_gelu_bwd(const void* dy, const void* pre, void* dx, int64_t n, int32_t dtype, sl_stream stream);
/* y = a*x + b*y (gradient accumulation at residual joins). */
int sl_axpby(const void* x, void* y, float a, float b, int64_t n, int32_t dtype, sl_stream stream);

/* torch.optim.AdamW.step() for every trainable tensor of the encoder in ONE launch (ref:trainer.py:97-105 builds the optimizer,
 * :380-383 steps it; arithmetic of torch/optim/adamw.py _multi_tensor_adamw, amsgrad off, maximize off: decay, lerp of exp_avg,
 * exp_avg_sq, bias corrections formed in double on the host, addcdiv), and — where `dst` is set — the kernel-layout copy of the
 * updated weight in the compute dtype written in the same pass (what AudioEncoder.refresh_weights would otherwise re-derive).
 * tensors_dev: device array of n_tensors records; first_block_dev: device array of n_tensors int64, first_block[t] = sum over
 * u < t of sl_adamw_blocks(n_u); total_blocks = that sum over all tensors.  p / g / m / v are fp32, caller-owned; `step` is the
 * 1-based count of this update (state['step'] after the increment).  The hyper-parameters are doubles because torch forms
 * 1 - beta, the bias corrections and lr / bias_correction1 from python floats before rounding them to fp32 once (1 - 0.999 in
 * fp32 is off by 1.3e-5).  Launches on `stream`, never synchronises. */
typedef struct sl_adamw_tensor {
  float* p;            /* fp32 master weight, updated in place */
  const float* g;      /* fp32 gradient */
  float* m;            /* exp_avg */
  float* v;            /* exp_avg_sq */
  void* dst;           /* NULL, or where the compute-dtype copy of p goes (same element order) */
  int64_t n;           /* elements */
  int32_t dst_dtype;   /* SL_F32 / SL_BF16 */
  int32_t reserved;
} sl_adamw_tensor;
size_t sl_adamw_blocks(int64_t n);
int sl_adamw_step(const sl_adamw_tensor* tensors_dev, const int64_t* first_block_dev, int32_t n_tensors, int64_t total_blocks, double lr,
                  double beta1, double beta2, double eps, double weight_decay, int64_t step, sl_stream stream);

/* y (cols, ld_out) = x (rows, cols)^T with columns rows..ld_out-1 zero-filled (row strides ldx / ldy in elements): the
 * K-contiguous operand copies of the backward products (torch's autograd transposes implicitly inside its GEMM calls,
 * hf training under ref:trainer.py:374-378 loss.backward()). */
int sl_transpose_pad(const void* x, int64_t ldx, void* y, int64_t ldy, int32_t rows, int32_t cols, int32_t ld_out, int32_t dtype,
                     sl_stream stream);

/* Training-mode dropout (hf:models/hubert/modeling_hubert.py feature-projection / hidden / activation dropouts of the
 * encoder the reference puts in train() mode, ref:trainer.py:258):  y = (residual ? residual : 0) + keep(i) * x / (1 - p).
 * keep(i) is a counter-based hash of (element index, seed) — no mask tensor; the backward pass calls the same entry
 * point on the gradient with the same seed.  n must be a multiple of 16 bytes' worth of elements; in place allowed. */
int sl_dropout(const void* x, const void* residual, void* y, int64_t n, float p, uint64_t seed, int32_t dtype, sl_stream stream);
/* SwiGLU on the 16-row interleaved gate/up activation gu (M, 2F): out (M, F) = silu(g)*u; backward
 * writes d gu in the same interleaved layout (hf:models/llama/modeling_llama.py:175). */
int sl_silu_mul(const void* gu, void* out, int64_t M, int32_t F, int32_t dtype, sl_stream stream);
int sl_silu_mul_bwd(const void* gu, const void* dy, void* dgu, int64_t M, int32_t F, int32_t dtype, sl_stream stream);
/* RoPE in place on the first n_rot heads of (n_tok, heads*D) rows; inverse != 0 applies the transposed
 * rotation (the backward of apply_rotary_pos_emb, hf:...llama.py:130-160). */
int sl_rope_inplace(void* x, const int32_t* tok_pos, const float* cos, const float* sin, int64_t n_tok, int32_t heads,
                    int32_t n_rot, int32_t D, int32_t inverse, int32_t dtype, sl_stream stream);
/* LayerNorm backward, optionally through a fused GELU (HuBERT conv layers): dx (may alias dy) and
 * fp32 dgamma/dbeta ACCUMULATED into the given buffers (may be NULL for input-only gradients). */
int sl_layernorm_bwd(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, float* dgamma,
                     float* dbeta, int64_t rows, int32_t cols, float eps, int32_t gelu, int32_t dtype, sl_stream stream);
/* The same with a scratch buffer of sl_layernorm_bwd_ws_bytes(rows, cols) bytes: the blocks' column partials of dgamma / dbeta go to
 * per-block records summed by a second launch instead of colliding in atomics on 2 x cols addresses (7 984 x 1 024: 79 -> ~20 us).
 * A NULL / too small workspace, or rows wider than 1 024 elements, fall back to the form above. */
size_t sl_layernorm_bwd_ws_bytes(int64_t rows, int32_t cols);
int sl_layernorm_bwd_ws(const void* x, const void* gamma, const void* beta, const void* dy, void* dx, float* dgamma, float* dbeta,
                        int64_t rows, int32_t cols, float eps, int32_t gelu, int32_t dtype, void* workspace, size_t workspace_bytes,
                        sl_stream stream);
/* LlamaRMSNorm backward, data gradient only (the LLM is frozen, ref:trainer.py:63-64). */
int sl_rmsnorm_bwd(const void* x, const void* w, const void* dy, void* dx, int64_t rows, int32_t cols, float eps,
                   int32_t dtype, sl_stream stream);
/* Weight-norm backward of the positional conv (hf:models/hubert/modeling_hubert.py:50-68, weight_norm(dim = 2): W[h][j][t] =
 * g[t] v[h][j][t] / ||v[:, :, t]||; ref:trainer.py:373-384 steps on g and v).  dW_khg: the tape's gradient of the folded weight in
 * the kernel layout (H, k, Hg) fp32; v (H, Hg, k), g (k) fp32 masters; writes dg (k) and dv (H, Hg, k).  workspace:
 * sl_weight_norm_bwd_workspace_bytes(k) bytes.  Fixed summation order (reproducible). */
size_t sl_weight_norm_bwd_workspace_bytes(int32_t k);
int sl_weight_norm_bwd(const float* dW_khg, const float* v, const float* g, float* dg, float* dv, float* workspace, int32_t H, int32_t Hg,
                       int32_t k, sl_stream stream);
/* out[c] += sum_r x[r][c]  (bias gradients), fp32 accumulate. */
int sl_colsum(const void* x, int64_t ld, float* out, int64_t rows, int32_t cols, int32_t dtype, sl_stream stream);
/* Explicit softmax over rows of fp32 scores (n_mats matrices of rows x cols, row stride ld):
 * P = softmax(scale*S [+ causal mask: column j visible to row i iff j <= i + cols - rows]); columns
 * [cols, ld) of P are zeroed.  Backward: dS = scale * P * (dP - rowsum(dP*P)). */
int sl_softmax_rows(const float* S, void* P, int64_t n_mats, int32_t rows, int32_t cols, int64_t ld, float scale,
                    int32_t causal, int32_t dtype, sl_stream stream);
int sl_softmax_bwd(const void* P, const float* dP, void* dS, int64_t nrows, int32_t cols, int64_t ld, float scale,
                   int32_t dtype, sl_stream stream);
/* The same over n_mats square matrices of different sizes stored at a common pitch (rows_per_mat rows of ld elements
 * each): matrix i is mat_dim[i] x mat_dim[i]; rows / columns past it are left alone / written as zeros. */
int sl_softmax_rows_var(const float* S, void* P, int64_t n_mats, int32_t rows_per_mat, const int32_t* mat_dim, int64_t ld, float scale,
                        int32_t causal, int32_t dtype, sl_stream stream);
int sl_softmax_bwd_var(const void* P, const float* dP, void* dS, int64_t n_mats, int32_t rows_per_mat, const int32_t* mat_dim, int64_t ld,
                       float scale, int32_t dtype, sl_stream stream);
/* Losses over rows of fp32 logits; *loss += coef * sum_rows(...), d logits (+)= coef * grad (dtype T):
 *   ce:      lse(s) - s[label]                    (ref:model/audio_llama.py:72-101 with coef = w/(n-1))
 *   soft-ce: -sum softmax(t) * log_softmax(s)     (ref:utils.py:167-178 with coef = w/n)
 *   mse:     mean((a-b)^2)                        (ref:trainer.py:358-370) */
int sl_ce_loss(const float* logits, const int32_t* labels, int64_t rows, int32_t V, float coef, float* loss, void* dlogits,
               int32_t accumulate, int32_t dtype, sl_stream stream);
int sl_soft_ce_loss(const float* student, const float* teacher, int64_t rows, int32_t V, float coef, float* loss,
                    void* dstudent, int32_t accumulate, int32_t dtype, sl_stream stream);
int sl_mse_loss(const void* a, const void* b, int64_t n, float coef, float* loss, void* da, int32_t accumulate,
                int32_t dtype, sl_stream stream);
/* The KD step's losses for a whole accumulation window in single launches (ref:trainer.py:325-370; what the per-utterance
 * forms above do one utterance and one term at a time):
 *   sl_kd_logit_losses: row r of the packed tail logits (student / teacher: float (rows, V)) carries
 *     labels[r] (int32, < 0: no next-token term), row_coef[r] = {ce loss, ce grad, soft-ce loss, soft-ce grad} weights and
 *     row_slot[r] = its utterance; losses (n_slots, loss_ld) float: [slot][0] += ce, [slot][1] += soft-ce;
 *     dstudent (rows, V) of `dtype` is OVERWRITTEN with the summed gradient.  teacher may be NULL (no soft-ce).
 *   sl_kd_mse_rows: feature-distillation MSE of one hidden-state tap: row_coef[r] = {loss weight on sum_h d^2, gradient
 *     weight on d}; losses[slot][loss_col] += ...; da (rows, H) overwritten with the gradient (may be NULL). */
int sl_kd_logit_losses(const float* student, const float* teacher, const int32_t* labels, const float* row_coef, const int32_t* row_slot,
                       int64_t rows, int32_t V, float* losses, int32_t loss_ld, void* dstudent, int32_t dtype, sl_stream stream);
int sl_kd_mse_rows(const void* a, const void* b, const float* row_coef, const int32_t* row_slot, int64_t rows, int32_t H, float* losses,
                   int32_t loss_ld, int32_t loss_col, void* da, int32_t dtype, sl_stream stream);
/* HuBERT front-end backward: AvgPool1d, strided-conv data gradient (col2im of the dgrad GEMM output),
 * fused conv0 (recomputes conv+LN, accumulates fp32 grads of w (C,k), bias, gamma, beta). */
int sl_avgpool_bwd(const void* dy, void* dx, int64_t T, int32_t H, int32_t kernel, int32_t stride, int64_t P, int32_t dtype,
                   sl_stream stream);
int sl_col2im(const void* dcol, void* dx, int64_t Lin, int64_t Lout, int32_t C, int32_t k, int32_t s, int32_t dtype,
              sl_stream stream);
/* The same two for every utterance of a packed (ragged) batch in ONE launch (ABI 7; ref:trainer.py:270-384 loops over utterances):
 * desc_dev = n_utt device records of four int64 {source rows (Lout / P), destination rows (Lin / T), first source row, first destination row}
 * into the packed source / destination buffers; max_Lin / max_T = the largest destination row count (sizes the grid). */
int sl_col2im_batch(const void* dcol, void* dx, const int64_t* desc_dev, int32_t n_utt, int64_t max_Lin, int32_t C, int32_t k, int32_t s,
                    int32_t dtype, sl_stream stream);
int sl_avgpool_bwd_batch(const void* dy, void* dx, const int64_t* desc_dev, int32_t n_utt, int64_t max_T, int32_t H, int32_t kernel,
                         int32_t stride, int32_t dtype, sl_stream stream);
int sl_hubert_conv0_bwd(const float* wave, int64_t n_samples, const float* w, const float* bias, const float* gamma,
                        const float* beta, const void* dy, int32_t C, int32_t k, int32_t stride, float eps, float* dw,
                        float* dbias, float* dgamma, float* dbeta, int32_t dtype, sl_stream stream);

/* sl_hubert_conv0_bwd for a whole ragged batch in one launch: utterance u has samples [sample_offsets[u], sample_offsets[u+1])
 * of `waves`, rows [row_offsets[u], row_offsets[u+1]) of dy, and strips strip_prefix[u] .. strip_prefix[u+1]-1 of 24 time
 * steps (all three arrays on the device, n_utt + 1 entries).  A fixed grid walks the strips and flushes its sums once. */
int sl_hubert_conv0_bwd_batch(const float* waves, const int64_t* sample_offsets_dev, const int64_t* row_offsets_dev,
                              const int64_t* strip_prefix_dev, int32_t n_utt, int64_t total_strips, const float* w, const float* bias,
                              const float* gamma, const float* beta, const void* dy, int32_t C, int32_t k, int32_t stride, float eps,
                              float* dw, float* dbias, float* dgamma, float* dbeta, int32_t dtype, sl_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Whole-model entry points (C++ host runtime inside the library: layer loops, workspace carving,
 * hipGraph capture of the decode step).  Weight tables are plain structs of device pointers.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void *ln1_g, *ln1_b, *wqkv, *bqkv, *wo, *bo, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2;
} sl_hubert_layer;

/* Optional per-layer copies for the inference path's LayerNorm fold (bf16; sl_gemm_ex_args.ln_*): the two LayerNorms of a stable-LN
 * layer (hf:models/hubert/modeling_hubert.py:515-517) disappear into the q|k|v and FFN1 Linears —
 *   wqkv_f = wqkv with ln1's gain multiplied into its columns, uqkv[n] = sum_k wqkv_f[n][k], cqkv[n] = (wqkv . ln1_b)[n] + bqkv[n];
 *   w1_f / u1 / c1 likewise from ln2, w1, b1.  The row statistics come out of the out_proj / FFN2 epilogues (stats_out). */
typedef struct {
  const void* wqkv_f; const float* uqkv; const float* cqkv;
  const void* w1_f; const float* u1; const float* c1;
} sl_hubert_fold;

typedef struct {
  int32_t dtype, n_conv, hidden, n_layers, n_heads, ffn, pos_k, pos_groups;
  int32_t conv_dim[8], conv_kernel[8], conv_stride[8];
  float ln_eps;   /* encoder / feature-projection LayerNorm eps; conv LayerNorms use 1e-5 */
  int32_t pool_kernel, pool_stride, llm_dim, reserved;   /* reserved: 0 = HuBERT front end, 1 = Whisper (see sl_whisper_forward) */
  const float *conv0_w, *conv0_b, *conv0_g, *conv0_beta;       /* fp32: (C,k), (C), (C), (C) */
  const void *conv_w[8], *conv_b[8], *conv_g[8], *conv_beta[8]; /* i>=1: (C_out, k*C_in) tap-major */
  const void *fp_ln_g, *fp_ln_b, *fp_w, *fp_b;                  /* feature projection */
  const void *pos_w, *pos_b;          /* (groups, H/groups, k*H/groups) weight-norm folded; (H) */
  const sl_hubert_layer* layers;      /* host array, n_layers entries */
  const void *final_ln_g, *final_ln_b;
  const void *proj_w, *proj_b;        /* embed_projection (llm_dim, hidden) */
  const sl_hubert_fold* fold;         /* NULL, or host array of n_layers entries: LayerNorms folded into the Linears (bf16 inference) */
} sl_hubert_model;

/* AudioEncoder.forward with the `pool` downsample for a batch of utterances
 * (ref:model/audio_encoder.py:56-63,87).  Every utterance is encoded at its own length (no padding,
 * so results equal the reference's batch-size-1 behaviour): the conv feature extractor
 * and positional conv run per utterance, the 24 transformer layers run on all frames packed together
 * with variable-length attention.
 *   waves: device fp32, utterances concatenated; sample_offsets_host: (n_utt+1) host int64.
 *   out: rows of out_ld elements; utterance u's P_u rows go to row out_row_offsets_host[u] (so the
 *   encoder writes straight into the LLM prompt buffer, ref:utils.py:66-72), or packed if NULL.
 *   last_hidden: optional (sum T_u, hidden) packed encoder output before pooling. */
size_t sl_hubert_workspace_bytes(const sl_hubert_model* m, const int64_t* sample_offsets_host, int32_t n_utt);
int sl_hubert_num_frames(const sl_hubert_model* m, int64_t n_samples);
int sl_hubert_forward(const sl_hubert_model* m, const float* waves, const int64_t* sample_offsets_host, int32_t n_utt,
                      void* out, int64_t out_ld, const int64_t* out_row_offsets_host, void* last_hidden,
                      void* workspace, size_t workspace_bytes, sl_stream stream);

/* Whisper path (BASELINE configs[3]; ref:model/audio_encoder.py:10-13, ref:trainer.py:168-199, 280-291).
 * sl_whisper_logmel: WhisperFeatureExtractor's torch path (hf:models/whisper/feature_extraction_whisper.py:135-168)
 *   for one utterance: zero-pad/trim to n_frames*hop samples, reflect pad, Hann STFT as an fp32 GEMM against
 *   dft_basis ((n_fft/2+1)*2 rows [cos | sin] x n_fft, window folded in), power, mel_w (n_mel x roundup4(n_fft/2+1),
 *   slaney), log10, max(x, max-8), (x+4)/4.  mel_out: (n_frames, n_mel) channel-last, dtype T.
 * sl_whisper_forward: WhisperEncoder (hf:models/whisper/modeling_whisper.py:592-646) + AvgPool + projection for
 *   n_utt mel inputs of exactly 2*max_source_positions frames each.  The model struct is sl_hubert_model with
 *   reserved = 1, conv_dim[0] = n_mel, conv_w/b[1] = conv1 and conv_w/b[2] = conv2 (tap-major (C_out, 3*C_in)),
 *   pos_w = embed_positions (pos_k = max_source_positions rows), layers' bqkv carrying zeros for the bias-less k_proj. */
size_t sl_whisper_logmel_workspace_bytes(int32_t n_fft, int32_t hop, int32_t n_frames, int32_t n_mel);
int sl_whisper_logmel(const float* audio, int64_t n_samples, const float* dft_basis, const float* mel_w, void* mel_out,
                      int32_t n_fft, int32_t hop, int32_t n_frames, int32_t n_mel, void* workspace, size_t workspace_bytes,
                      int32_t dtype, sl_stream stream);
size_t sl_whisper_workspace_bytes(const sl_hubert_model* m, int32_t n_utt);
int sl_whisper_forward(const sl_hubert_model* m, const void* mel, int32_t n_utt, void* out, int64_t out_ld,
                       const int64_t* out_row_offsets_host, void* last_hidden, void* workspace, size_t workspace_bytes,
                       sl_stream stream);

typedef struct {
  const void *norm1, *wqkv, *wo, *norm2, *wgu, *wdown;   /* row-major (prefill); wgu: 16-row gate/up interleave */
  /* decode copies, fragment-packed (sl_pack_weight); NULL -> decode falls back to the row-major set.
   * wqkv_dec rows: q/k heads in rotate_half pair order (see sl_gemm_fused); if dec_fused_norm the
   * RMSNorm gains are pre-multiplied into wqkv_dec / wgu_dec (and final_norm into lm_head_dec). */
  const void *wqkv_dec, *wo_dec, *wgu_dec, *wdown_dec;
} sl_llama_layer;

typedef struct {
  int32_t dtype, hidden, n_layers, n_heads, n_kv_heads, head_dim, ffn, vocab;
  float rms_eps; int32_t rope_len;
  const void *embed, *lm_head, *final_norm;
  const float *rope_cos, *rope_sin;      /* (rope_len, head_dim/2) */
  const sl_llama_layer* layers;          /* host array */
  const void* lm_head_dec;               /* packed (gain-folded if dec_fused_norm) or NULL */
  int32_t dec_fused_norm, reserved;
} sl_llama_model;

typedef struct {
  void* k_cache; void* v_cache;  /* (n_layers, slots, n_kv, max_ctx, D) each */
  int32_t slots, max_ctx;
  /* shared_prefix = P > 0 is the caller's promise that the first P prompt positions of EVERY sequence of a generate call carry the
   * same input rows (one prompt template in front of the audio, ref:inference.py:95-113 builds it per call): prefill then leaves
   * bit-identical K/V rows at positions [0, P) of every slot, and the batched decode attention reads those positions from slot 0
   * (out of L2 instead of once per sequence from HBM).  0 = no promise.  P <= every prompt length (checked). */
  int32_t shared_prefix, reserved;
} sl_kv_cache;

/* LlamaModel.forward over packed prompt embeddings + last-token logits
 * (hf:...llama.py:367-417 + ref:model/audio_llama.py:67 with logits for the last position only).
 *   x: (n_tok, hidden) packed prompts (modified in place: residual stream);  cu_seqlens (nseq+1)
 *   host array.  Writes K/V for positions [0, len) of slot s, logits (nseq, vocab) fp32, and
 *   ctx_len[s] = len (device int32).  hidden_taps, if non-NULL, receives the (n_layers+1) hidden
 *   states (each (n_tok, hidden), last one post-norm) like output_hidden_states=True. */
size_t sl_llama_workspace_bytes(const sl_llama_model* m, int64_t n_tok, int32_t nseq);
int sl_llama_prefill(const sl_llama_model* m, const sl_kv_cache* kv, void* x, const int32_t* cu_seqlens_host,
                     int32_t nseq, float* logits, int32_t* ctx_len_dev, void* hidden_taps, void* workspace,
                     size_t workspace_bytes, sl_stream stream);

/* One KV-cached decode step for B sequences: embeds next_ids, runs all layers with M = B, writes
 * logits (B, vocab) fp32 (B <= SL_MAX_DECODE_BATCH).  The new token's K/V are appended at position ctx_len[b] and
 * ctx_len[b]+1 keys are attended; ctx_len itself is advanced by sl_greedy_select.  Reads/writes only
 * device state, so the call is hipGraph-capturable.  Workspace: sl_llama_workspace_bytes(m, B, B) + B*hidden. */
int sl_llama_decode_step(const sl_llama_model* m, const sl_kv_cache* kv, const int32_t* next_ids_dev,
                         const int32_t* ctx_len_dev, int32_t B, float* logits, void* workspace,
                         size_t workspace_bytes, sl_stream stream);

/* GenerationMixin._sample in greedy mode from prompt embeddings (ref:inference.py:60-66):
 * prefill + up to max_new_tokens decode steps replayed from one captured hipGraph.  out_ids_host
 * (nseq, max_new_tokens) int32 receives new tokens only; *n_steps_host the number of columns
 * produced (all rows finished => early stop, checked every `check_every` steps).  This call
 * synchronises `stream` (it returns host data).
 * LIMIT: nseq <= SL_MAX_DECODE_BATCH (2048) sequences per call — the row count the decode GEMMs, the single-pass attention grid and
 * the 288 GB of HBM are sized for (a KV cache of 2 048 slots x 393 positions of Llama-3.2-3B is 92 GB); a larger batch is rejected with
 * SL_ERR_ARG (split it: sequences are independent — the Python surface does, inference.generate_audio_responses).
 * The captured decode graph is cached per calling thread, keyed by every buffer, limit, the device and a hash of the model's
 * and every layer's fields; sl_decode_graph_cache_clear() destroys the calling thread's cached graphs (returns how many) —
 * call it after re-laying-out weights in place behind unchanged struct contents, or before unloading the library. */
#define SL_MAX_DECODE_BATCH 2048
int sl_decode_graph_cache_clear(void);
size_t sl_generate_workspace_bytes(const sl_llama_model* m, int64_t n_tok, int32_t nseq, int32_t max_new_tokens);
int sl_greedy_generate(const sl_llama_model* m, const sl_kv_cache* kv, void* x, const int32_t* cu_seqlens_host,
                       int32_t nseq, int32_t max_new_tokens, const int32_t* eos_ids_host, int32_t n_eos,
                       int32_t pad_id, int32_t use_eos, int32_t check_every, int32_t* out_ids_host,
                       int32_t* n_steps_host, float* timings_ms_host /* [prefill, decode] or NULL */,
                       void* workspace, size_t workspace_bytes, sl_stream stream);

/* sl_generate (ABI version 6): the general form of the two calls around it — same prefill, same captured decode step, same HF
 * semantics (hf:generation/utils.py:2928-2942: a finished row emits pad_id, generation ends at the step where the last row
 * finished) — plus what a batch whose answers have different lengths needs:
 *   row_limits_host : NULL, or one token budget per sequence in [1, max_new_tokens] (a per-request max_new_tokens): a row that
 *                     has produced its budget is finished exactly like a row that emitted EOS.  Implies use_eos semantics.
 *   compact         : 1 = at a `check_every` synchronisation, once the live rows fit the next lower rung of a fixed ladder of
 *                     row counts (multiples of 64 from 128 up; 96, 64, 48, 32, 24, 16, 12, 8, 6, 4, 3, 2, 1 below), the batch is COMPACTED: finished
 *                     sequences are written out, the live rows above the rung take the places of finished rows below it — their
 *                     state, output ids and K / V cache slot move with them — and decoding continues with that many rows (one
 *                     cached graph per rung).  Per sequence the result is what compact = 0 gives: identical ids in SL_F32 (every
 *                     kernel family is bit-exact there); in SL_BF16 the row count selects the GEMM family, so a logit near-tie
 *                     may resolve differently after a compaction, as it may between two batch sizes.  The K / V cache is
 *                     re-ordered in place: after the call slot j no longer belongs to sequence j.
 *   sample          : 0 = greedy argmax; 1 = temperature / top_k / top_p / seed as in sl_sample_generate (the draw of a
 *                     sequence depends on (seed, the sequence's index in THIS call, step) only, compacted or not).
 * out_ids_host (nseq, max_new_tokens) is indexed by the caller's sequence order whatever moved.  stats may be NULL. */
typedef struct {
  const int32_t* eos_ids_host;     /* n_eos ids (<= 8) */
  const int32_t* row_limits_host;  /* NULL or nseq budgets */
  uint64_t seed;
  int32_t max_new_tokens;
  int32_t n_eos;
  int32_t pad_id;
  int32_t use_eos;
  int32_t check_every;             /* host check of the finished flags every this many steps (<= 0: 16) */
  int32_t sample;
  float temperature;
  int32_t top_k;
  float top_p;
  int32_t compact;
} sl_generate_opts;
typedef struct {
  int64_t row_steps;        /* sum over decode launches of the rows the launch ran (nseq x launches without compaction) */
  int32_t n_steps;          /* columns of out_ids_host produced */
  int32_t decode_launches;
  int32_t compactions;
  int32_t final_rows;       /* rows of the last decode launch */
  float prefill_ms;
  float decode_ms;
} sl_generate_stats;
int sl_generate(const sl_llama_model* m, const sl_kv_cache* kv, void* x, const int32_t* cu_seqlens_host, int32_t nseq,
                const sl_generate_opts* opts, int32_t* out_ids_host, sl_generate_stats* stats, void* workspace,
                size_t workspace_bytes, sl_stream stream);

/* Sampled generation (hf:generation/utils.py:2911-2923 with do_sample = True — what the hub's generation_config.json of
 * Llama-3.2-3B-Instruct asks for when a caller does not force greedy): HF's logits warpers in HF's order —
 * temperature (scores / T), top-k (keep scores >= the k-th largest; 0 = off; HF's default is 50), top-p over the
 * top-k-filtered distribution (drop a token iff the mass ranked above it is >= top_p; 1.0 = off) — then one draw per row from
 * the renormalised survivors by inverse CDF in index order at u = hash(seed, row, step): reproducible for a given seed (HF's
 * own stream is torch.multinomial on the global CUDA generator, which no other implementation can reproduce).
 * sl_sample_select is sl_greedy_select with the draw in place of the argmax (choice_ws: B int32 scratch);
 * sl_sample_generate is sl_greedy_generate with it (same workspace size, same synchronisation points). */
int sl_sample_select(const float* logits, int32_t B, int32_t V, float temperature, int32_t top_k, float top_p, uint64_t seed,
                     const int32_t* eos_ids_host, int32_t n_eos, int32_t pad_id, int32_t use_eos, int32_t* unfinished, int32_t* ctx_len,
                     int32_t* gen_count, int32_t* finish_len, int32_t* next_ids, int32_t* out_ids, int32_t max_new, int32_t* choice_ws,
                     sl_stream stream);
int sl_sample_generate(const sl_llama_model* m, const sl_kv_cache* kv, void* x, const int32_t* cu_seqlens_host, int32_t nseq,
                       int32_t max_new_tokens, const int32_t* eos_ids_host, int32_t n_eos, int32_t pad_id, int32_t use_eos,
                       int32_t check_every, float temperature, int32_t top_k, float top_p, uint64_t seed, int32_t* out_ids_host,
                       int32_t* n_steps_host, float* timings_ms_host, void* workspace, size_t workspace_bytes, sl_stream stream);

/* ---------------------------------------------------------------------------------------------
 * KD step, layer stacks (C++ host runtime of the training tape: one call issues the launches of a whole stack of layers
 * over a packed ragged batch; ref:trainer.py:270-384 runs the same arithmetic through autograd, one utterance at a time).
 * Saved activations, hidden states and fp32 gradient accumulators are caller-owned; temporaries come from `workspace`.
 *
 * Encoder (HuBERT stable-LN layer hf:models/hubert/modeling_hubert.py:504-547; Whisper hf:models/whisper/modeling_whisper.py:
 * 360-414; head_dim 64, bidirectional): train()-mode regularisers as sl_dropout / sl_attn_args.dropout_p, LayerDrop via `skip`.
 *   seeds: HOST array (n_layers, 4) = {attention-probability, attention-output, activation, ffn-output} dropout seeds.
 *   fwd: layer l reads its input (x_in or the previous layer's x_out), fills saved[l] (and records the input pointer in
 *        saved[l].x); *x_out = the stack's output.  bwd: layers layer_end-1 .. layer_begin; dx holds d(output of layer
 *        layer_end-1) on entry and d(input of layer layer_begin) on return; parameter gradients ACCUMULATE into grads[l]
 *        (kernel layouts: wqkv (3H, H), w1 (F, H), w2 (H, F), ...).
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  int32_t dtype, hidden, n_heads, ffn, n_layers, nseq, max_len, reserved;
  int64_t n_tok;
  float ln_eps, p_hidden, p_act, p_attn;
  const int32_t* cu;       /* (nseq+1) device: first packed row of each sequence */
  const int32_t* klen;     /* (nseq) device */
  const uint8_t* skip;     /* (n_layers) HOST: 1 = LayerDrop skips the layer */
  const uint64_t* seeds;   /* (n_layers, 4) HOST */
} sl_enc_stack_cfg;
typedef struct {
  const void* x;                                            /* layer input, recorded by the forward */
  void *ln1, *qkv, *att, *x_mid, *ln2, *pre1, *mid, *x_out; /* (n_tok, H | 3H | H | H | H | F | F | H) */
  float* lse;                                               /* (n_tok, n_heads) */
} sl_enc_layer_saved;
typedef struct {
  float *ln1_g, *ln1_b, *wqkv, *bqkv, *wo, *bo, *ln2_g, *ln2_b, *w1, *b1, *w2, *b2;
} sl_enc_layer_grads;
size_t sl_encoder_stack_train_workspace_bytes(const sl_enc_stack_cfg* c);
int sl_encoder_stack_train_fwd(const sl_hubert_layer* layers, const sl_enc_stack_cfg* c, const void* x_in, sl_enc_layer_saved* saved,
                               const void** x_out, void* workspace, size_t workspace_bytes, sl_stream stream);
int sl_encoder_stack_train_bwd(const sl_hubert_layer* layers, const sl_enc_stack_cfg* c, const sl_enc_layer_saved* saved,
                               const sl_enc_layer_grads* grads, int32_t layer_begin, int32_t layer_end, void* dx, void* workspace,
                               size_t workspace_bytes, sl_stream stream);

/* Frozen Llama decoder (hf:models/llama/modeling_llama.py:284-324; head_dim 128, causal GQA) over packed sequences:
 *   hidden: HOST array of n_layers + 1 device buffers (n_tok, H): hidden[0] = input embeddings (caller-filled),
 *           hidden[l + 1] = output of layer l (written by fwd) — these are HF's `hidden_states` taps before the final norm.
 *   fwd:    saved == NULL keeps nothing else (teacher pass, torch.no_grad in ref:trainer.py:337-344); otherwise saved[l]
 *           receives what the data-gradient backward needs.
 *   bwd:    dx = d(hidden[n_layers]) on entry, d(hidden[0]) on return; d_tap (HOST array of n_layers pointers or NULL):
 *           d_tap[l] != NULL is added to the gradient of hidden[l] (feature-distillation terms, ref:trainer.py:358-370).
 *   pos: (n_tok) device position ids; layers[l].*_t: transposed copies (in, out) of the frozen weights for the dgrad GEMMs. */
typedef struct {
  int32_t dtype, hidden, n_heads, n_kv_heads, head_dim, ffn, n_layers, nseq, max_len, reserved;
  int64_t n_tok;
  float rms_eps;
  const int32_t* cu;
  const int32_t* klen;
  const int32_t* pos;
  const float* rope_cos;
  const float* rope_sin;
} sl_llama_stack_cfg;
typedef struct {
  const void *norm1, *wqkv, *wo, *norm2, *wgu, *wdown;   /* nn.Linear layouts; gate/up interleaved in 16-row blocks */
  const void *wqkv_t, *wo_t, *wgu_t, *wdown_t;           /* (in, out) copies; only the backward reads them */
} sl_llama_train_layer;
typedef struct {
  void *qkv, *x2, *gu, *att;   /* (n_tok, (nh+2nkv)D | H | 2F | nh D) */
  float* lse;                  /* (n_tok, n_heads) */
} sl_llama_layer_saved;
size_t sl_llama_stack_train_workspace_bytes(const sl_llama_stack_cfg* c);
int sl_llama_stack_train_fwd(const sl_llama_train_layer* layers, const sl_llama_stack_cfg* c, void* const* hidden,
                             const sl_llama_layer_saved* saved, void* workspace, size_t workspace_bytes, sl_stream stream);
int sl_llama_stack_train_bwd(const sl_llama_train_layer* layers, const sl_llama_stack_cfg* c, void* const* hidden,
                             const sl_llama_layer_saved* saved, void* const* d_tap, void* dx, void* workspace, size_t workspace_bytes,
                             sl_stream stream);

/* ---------------------------------------------------------------------------------------------
 * Collective (group 11): the ONE exchange of the hot path — the in-place sum of the encoder's fp32 gradient arena over the ranks
 * of a data-parallel KD step.  The reference has none (ref:README.md:86: "only supports training on a single GPU with a batch size
 * of 1"); its accumulation boundary ref:trainer.py:373-384 is where the sum goes.  RCCL over xGMI, one process per GPU; librccl is
 * bound at run time (a host without it gets SL_ERR_UNSUPPORTED from these calls and everything else keeps working).
 *   sl_comm_unique_id : rank 0 draws the 128-byte RCCL id; the caller carries it to the other ranks over whatever channel launched
 *                       them (torchrun's store, MPI, a file) — the library opens no sockets of its own.
 *   sl_comm_init      : collective over `world` processes, each on its own current device; the communicator belongs to that device.
 *   sl_allreduce_sum  : in place on `count` elements (SL_F32 / SL_BF16) of caller-owned device memory, asynchronous on `stream`
 *                       (the KD step passes a side stream behind an event on the kernels' stream); same call order on all ranks.
 *   sl_comm_destroy   : collective; the caller has synchronised the streams it used.
 *   sl_comm_abort     : LOCAL tear-down (ncclCommAbort): for a communicator whose peers may never have made theirs — a start-up the
 *                       ranks voted to abandon, or one obtained after this rank had given up waiting (ABI version 6).
 * --------------------------------------------------------------------------------------------- */
#define SL_COMM_ID_BYTES 128
typedef struct sl_comm_s* sl_comm;
int sl_comm_unique_id(void* id_out /* SL_COMM_ID_BYTES */);
int sl_comm_init(sl_comm* comm_out, const void* unique_id /* SL_COMM_ID_BYTES */, int32_t rank, int32_t world);
int sl_allreduce_sum(sl_comm comm, void* buf, int64_t count, int32_t dtype, sl_stream stream);
int sl_comm_destroy(sl_comm comm);
int sl_comm_abort(sl_comm comm);
int32_t sl_comm_rank(sl_comm comm);    /* -1: not a communicator */
int32_t sl_comm_world(sl_comm comm);

#ifdef __cplusplus
}
#endif
#endif /* SPEECHLLM_H */
