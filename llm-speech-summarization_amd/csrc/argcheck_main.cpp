// argcheck_main.cpp — sanitizer driver for the HOST side of libspeechllm (the reference has no sanitizer runs).
// Built by `make asan` against a host-ASan + UBSan build of the library (device code is not instrumented: GPU ASan needs
// xnack+, which this pool does not offer) and run on a machine WITHOUT a GPU: every call below must be rejected or answered by
// host code alone — argument validation, shape / workspace arithmetic, the tuning-switch parser — before any HIP call, with
// the right status and a non-empty error string, and without the sanitizers reporting anything.
#include <initializer_list>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/speechllm.h"

static int g_fail = 0;
#define EXPECT(cond, what)                                                         \
  do {                                                                             \
    if (!(cond)) { fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, what); ++g_fail; } \
  } while (0)
#define EXPECT_ARG_ERROR(call)                                                      \
  do {                                                                              \
    const int rc__ = (call);                                                        \
    EXPECT(rc__ == SL_ERR_ARG || rc__ == SL_ERR_UNSUPPORTED, #call " must be rejected"); \
    EXPECT(sl_last_error()[0] != 0, #call " must leave an error message");         \
  } while (0)

static void fill_hubert(sl_hubert_model* m, sl_hubert_layer* layers, int n_layers) {
  memset(m, 0, sizeof(*m));
  m->dtype = SL_BF16; m->n_conv = 7; m->hidden = 1024; m->n_layers = n_layers; m->n_heads = 16; m->ffn = 4096; m->pos_k = 128; m->pos_groups = 16;
  const int k[7] = {10, 3, 3, 3, 3, 2, 2}, s[7] = {5, 2, 2, 2, 2, 2, 2};
  for (int i = 0; i < 7; ++i) { m->conv_dim[i] = 512; m->conv_kernel[i] = k[i]; m->conv_stride[i] = s[i]; }
  m->ln_eps = 1e-5f; m->pool_kernel = 8; m->pool_stride = 4; m->llm_dim = 3072;
  m->layers = layers;
}

int main() {
  EXPECT(sl_version() == SL_ABI_VERSION && SL_ABI_VERSION == 7, "ABI version");
  EXPECT(sl_last_error() != nullptr, "error string never NULL");

  // ---- tuning switches: parser under the sanitizers, garbage included
  setenv("SL_STREAM_CFG", "4,2,16", 1);
  setenv("SL_STREAM_MIN_M", "-7", 1);
  setenv("SL_T256_MIN_TILES", "99999999999999999999", 1);
  setenv("SL_DISABLE_GLDS", "2", 1);
  EXPECT(sl_tuning_reload() == 0, "tuning reload");
  setenv("SL_STREAM_CFG", ",,,;;", 1);
  EXPECT(sl_tuning_reload() == 0, "tuning reload (garbage)");
  unsetenv("SL_STREAM_CFG"); unsetenv("SL_STREAM_MIN_M"); unsetenv("SL_T256_MIN_TILES"); unsetenv("SL_DISABLE_GLDS");
  EXPECT(sl_tuning_reload() == 0, "tuning reload (defaults)");

  // ---- GEMM family
  EXPECT_ARG_ERROR(sl_gemm(nullptr, nullptr));
  sl_gemm_args g;
  memset(&g, 0, sizeof(g));
  EXPECT_ARG_ERROR(sl_gemm(&g, nullptr));                       // M = N = K = 0
  g.M = 4; g.N = 8; g.K = 8; g.batch = 1; g.dtype = 7;
  EXPECT_ARG_ERROR(sl_gemm(&g, nullptr));                       // unknown dtype
  g.dtype = SL_BF16; g.K = 12;
  EXPECT_ARG_ERROR(sl_gemm(&g, nullptr));                       // K % 8
  EXPECT_ARG_ERROR(sl_gemm_ex(nullptr, nullptr, nullptr));
  EXPECT_ARG_ERROR(sl_gemm_fused_decode(nullptr, nullptr, nullptr));
  EXPECT_ARG_ERROR(sl_pack_weight(nullptr, 0, nullptr, 16, 32, SL_BF16, nullptr));
  for (int M : {1, 16, 33, 64, 128, 129, 512, 1024})
    for (int N : {48, 3072, 5120, 16384, 128256})
      for (int K : {64, 3072, 8192}) {
        const size_t b = sl_gemm_split_workspace_bytes(M, N, K, SL_BF16);
        const int sp = sl_gemm_split_count(M, N, K, SL_BF16);
        EXPECT(sp >= 1 && sp <= 64, "split count in range");
        EXPECT(b < ((size_t)1 << 40), "split workspace sane");
      }

  // ---- attention
  EXPECT_ARG_ERROR(sl_attn_fwd(nullptr, nullptr));
  sl_attn_args a;
  memset(&a, 0, sizeof(a));
  EXPECT_ARG_ERROR(sl_attn_fwd(&a, nullptr));
  int dummy[4] = {0, 1, 2, 3};
  a.q = a.k = a.v = dummy; a.out = dummy; a.cu_q = a.cu_k = a.klen = dummy;
  a.nseq = 1; a.max_qlen = 4; a.n_heads = 3; a.n_kv_heads = 2; a.head_dim = 64; a.dtype = SL_BF16;
  EXPECT_ARG_ERROR(sl_attn_fwd(&a, nullptr));                   // heads not a multiple of kv heads
  a.n_heads = 4; a.dropout_p = 1.5f;
  EXPECT_ARG_ERROR(sl_attn_fwd(&a, nullptr));                   // dropout_p outside [0, 1)
  a.dropout_p = 0.f; a.q_row_stride = 3;
  EXPECT_ARG_ERROR(sl_attn_fwd(&a, nullptr));                   // misaligned strides
  EXPECT_ARG_ERROR(sl_attn_bwd(nullptr, nullptr));
  sl_attn_bwd_args ab;
  memset(&ab, 0, sizeof(ab));
  EXPECT_ARG_ERROR(sl_attn_bwd(&ab, nullptr));
  EXPECT_ARG_ERROR(sl_attn_decode(nullptr, 0, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0, 0, 1.f, SL_BF16, nullptr));
  EXPECT(sl_attn_decode_workspace_bytes(512, 24, 8, 448) > 0, "decode attention workspace");

  // ---- norms / element-wise / losses
  EXPECT_ARG_ERROR(sl_layernorm(nullptr, nullptr, nullptr, nullptr, 4, 1024, 1e-5f, 0, SL_BF16, nullptr));
  EXPECT_ARG_ERROR(sl_rmsnorm(nullptr, nullptr, nullptr, 4, 3072, 1e-5f, SL_BF16, nullptr));
  EXPECT_ARG_ERROR(sl_embed_gather(nullptr, nullptr, nullptr, 4, 3072, SL_BF16, nullptr));
  EXPECT_ARG_ERROR(sl_kd_logit_losses(nullptr, nullptr, nullptr, nullptr, nullptr, 4, 1000, nullptr, 3, nullptr, SL_BF16, nullptr));
  EXPECT_ARG_ERROR(sl_kd_mse_rows(nullptr, nullptr, nullptr, nullptr, 4, 3072, nullptr, 3, 2, nullptr, SL_BF16, nullptr));
  EXPECT_ARG_ERROR(sl_ce_loss(nullptr, nullptr, 4, 1000, 1.f, nullptr, nullptr, 0, SL_BF16, nullptr));
  EXPECT_ARG_ERROR(sl_hubert_conv0(nullptr, 16000, nullptr, nullptr, nullptr, nullptr, nullptr, 512, 10, 5, 1e-5f, SL_BF16, nullptr));
  float w1[1] = {0.f};
  EXPECT_ARG_ERROR(sl_hubert_conv0(w1, 16000, w1, w1, w1, w1, w1, 512, 3, 2, 1e-5f, SL_BF16, nullptr));   // geometry not built
  EXPECT_ARG_ERROR(sl_hubert_conv0(w1, 5, w1, w1, w1, w1, w1, 512, 10, 5, 1e-5f, SL_BF16, nullptr));      // shorter than the kernel

  // ---- HuBERT runtime: frame arithmetic and plan validation are host code
  sl_hubert_layer hl[24];
  memset(hl, 0, sizeof(hl));
  sl_hubert_model hm;
  fill_hubert(&hm, hl, 24);
  EXPECT(sl_hubert_num_frames(&hm, 160000) == 499, "10 s -> 499 frames");
  EXPECT(sl_hubert_num_frames(&hm, 16000) == 49, "1 s -> 49 frames");
  EXPECT(sl_hubert_num_frames(&hm, 300) == 0, "too short -> 0");
  int64_t offs[4] = {0, 160000, 160000 + 32000, 160000 + 32000 + 1920000};
  EXPECT(sl_hubert_workspace_bytes(&hm, offs, 3) > 0, "encoder workspace");
  int64_t short_offs[2] = {0, 200};
  EXPECT(sl_hubert_workspace_bytes(&hm, short_offs, 1) == 0, "too-short utterance -> 0 workspace + error");
  float wave[16] = {0};
  char ws[256];
  hm.proj_w = w1;
  EXPECT_ARG_ERROR(sl_hubert_forward(&hm, wave, short_offs, 1, ws, 3072, nullptr, nullptr, ws, sizeof(ws), nullptr));
  EXPECT_ARG_ERROR(sl_hubert_forward(&hm, wave, offs, 3, ws, 3072, nullptr, nullptr, ws, sizeof(ws), nullptr));     // workspace too small
  EXPECT_ARG_ERROR(sl_hubert_forward(nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, nullptr));
  sl_hubert_model wm = hm;
  EXPECT_ARG_ERROR(sl_whisper_forward(&wm, wave, 1, ws, 3072, nullptr, nullptr, ws, sizeof(ws), nullptr));          // not a Whisper struct

  // ---- Llama runtime
  sl_llama_layer ll[28];
  memset(ll, 0, sizeof(ll));
  sl_llama_model lm;
  memset(&lm, 0, sizeof(lm));
  lm.dtype = SL_BF16; lm.hidden = 3072; lm.n_layers = 28; lm.n_heads = 24; lm.n_kv_heads = 8; lm.head_dim = 128; lm.ffn = 8192; lm.vocab = 128256;
  lm.rms_eps = 1e-5f; lm.rope_len = 448; lm.layers = ll;
  EXPECT(sl_llama_workspace_bytes(&lm, 512 * 137, 512) > 0, "prefill workspace");
  EXPECT(sl_generate_workspace_bytes(&lm, 512 * 137, 512, 256) > sl_llama_workspace_bytes(&lm, 512 * 137, 512), "generate workspace");
  sl_kv_cache kv;
  memset(&kv, 0, sizeof(kv));
  int32_t cu[2] = {0, 137};
  float logits[1];
  int32_t ctx[1];
  EXPECT_ARG_ERROR(sl_llama_prefill(&lm, &kv, ws, cu, 1, logits, ctx, nullptr, ws, sizeof(ws), nullptr));            // null model fields
  lm.embed = lm.lm_head = lm.final_norm = w1; lm.rope_cos = lm.rope_sin = w1;
  EXPECT_ARG_ERROR(sl_llama_prefill(&lm, &kv, ws, cu, 1, logits, ctx, nullptr, ws, sizeof(ws), nullptr));            // bad kv cache
  kv.k_cache = kv.v_cache = ws; kv.slots = 1; kv.max_ctx = 4096;
  EXPECT_ARG_ERROR(sl_llama_prefill(&lm, &kv, ws, cu, 1, logits, ctx, nullptr, ws, sizeof(ws), nullptr));            // max_ctx beyond the rope table
  kv.max_ctx = 256;
  int32_t out_ids[8], n_steps = 0;
  int32_t eos[3] = {128001, 128008, 128009};
  EXPECT_ARG_ERROR(sl_greedy_generate(&lm, &kv, ws, cu, 1, 256, eos, 3, 128001, 1, 16, out_ids, &n_steps, nullptr, ws, sizeof(ws), nullptr));   // prompt + new > max_ctx
  EXPECT_ARG_ERROR(sl_greedy_generate(&lm, &kv, ws, cu, 4096, 8, eos, 3, 128001, 1, 16, out_ids, &n_steps, nullptr, ws, sizeof(ws), nullptr));  // batch above the limit
  kv.shared_prefix = 200;
  EXPECT_ARG_ERROR(sl_greedy_generate(&lm, &kv, ws, cu, 1, 8, eos, 3, 128001, 1, 16, out_ids, &n_steps, nullptr, ws, sizeof(ws), nullptr));     // shared prefix longer than the prompt
  kv.shared_prefix = 0;
  {   // sl_generate (ABI 6): options checked before anything is launched
    sl_generate_opts go;
    memset(&go, 0, sizeof(go));
    sl_generate_stats gs;
    go.max_new_tokens = 8; go.eos_ids_host = eos; go.n_eos = 3; go.pad_id = 128001; go.use_eos = 1; go.compact = 1;
    EXPECT_ARG_ERROR(sl_generate(&lm, &kv, ws, cu, 1, nullptr, out_ids, &gs, ws, sizeof(ws), nullptr));                 // no options
    EXPECT_ARG_ERROR(sl_generate(&lm, &kv, ws, cu, SL_MAX_DECODE_BATCH + 1, &go, out_ids, &gs, ws, sizeof(ws), nullptr));   // batch above the limit
    go.n_eos = 9;
    EXPECT_ARG_ERROR(sl_generate(&lm, &kv, ws, cu, 1, &go, out_ids, &gs, ws, sizeof(ws), nullptr));                      // more than 8 eos ids
    go.n_eos = 3;
    int32_t lim[1] = {9};
    go.row_limits_host = lim;
    EXPECT_ARG_ERROR(sl_generate(&lm, &kv, ws, cu, 1, &go, out_ids, &gs, ws, sizeof(ws), nullptr));                      // budget above max_new_tokens
    lim[0] = 0;
    EXPECT_ARG_ERROR(sl_generate(&lm, &kv, ws, cu, 1, &go, out_ids, &gs, ws, sizeof(ws), nullptr));                      // budget below 1
    go.row_limits_host = nullptr;
    go.sample = 1; go.temperature = 0.f; go.top_p = 1.f;
    EXPECT_ARG_ERROR(sl_generate(&lm, &kv, ws, cu, 1, &go, out_ids, &gs, ws, sizeof(ws), nullptr));                      // temperature 0
    go.sample = 0;
    EXPECT_ARG_ERROR(sl_generate(&lm, &kv, ws, cu, 1, &go, out_ids, &gs, ws, 64, nullptr));                              // workspace too small
  }
  EXPECT(sl_comm_abort(nullptr) == 0, "aborting no communicator is a no-op");
  kv.shared_prefix = -1;
  EXPECT_ARG_ERROR(sl_llama_prefill(&lm, &kv, ws, cu, 1, logits, ctx, nullptr, ws, sizeof(ws), nullptr));            // negative shared prefix
  kv.shared_prefix = 0;
  lm.head_dim = 96;
  EXPECT_ARG_ERROR(sl_llama_prefill(&lm, &kv, ws, cu, 1, logits, ctx, nullptr, ws, sizeof(ws), nullptr));            // head_dim not built
  EXPECT_ARG_ERROR(sl_llama_decode_step(&lm, &kv, nullptr, nullptr, 1, logits, ws, sizeof(ws), nullptr));

  // ---- KD tape runtime
  sl_enc_stack_cfg ec;
  memset(&ec, 0, sizeof(ec));
  ec.dtype = SL_BF16; ec.hidden = 1024; ec.n_heads = 16; ec.ffn = 4096; ec.n_layers = 24; ec.nseq = 16; ec.max_len = 499; ec.n_tok = 16 * 499;
  EXPECT(sl_encoder_stack_train_workspace_bytes(&ec) > 0, "encoder tape workspace");
  EXPECT_ARG_ERROR(sl_encoder_stack_train_fwd(nullptr, &ec, nullptr, nullptr, nullptr, nullptr, 0, nullptr));
  EXPECT_ARG_ERROR(sl_encoder_stack_train_bwd(nullptr, &ec, nullptr, nullptr, 0, 24, nullptr, nullptr, 0, nullptr));
  sl_llama_stack_cfg lc;
  memset(&lc, 0, sizeof(lc));
  lc.dtype = SL_BF16; lc.hidden = 3072; lc.n_heads = 24; lc.n_kv_heads = 8; lc.head_dim = 128; lc.ffn = 8192; lc.n_layers = 28; lc.nseq = 16; lc.max_len = 200;
  lc.n_tok = 3200;
  EXPECT(sl_llama_stack_train_workspace_bytes(&lc) > 0, "llama tape workspace");
  EXPECT_ARG_ERROR(sl_llama_stack_train_fwd(nullptr, &lc, nullptr, nullptr, nullptr, 0, nullptr));
  EXPECT_ARG_ERROR(sl_llama_stack_train_bwd(nullptr, &lc, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr));

  // ---- round-3 entry points
  EXPECT_ARG_ERROR(sl_adamw_step(nullptr, nullptr, 0, 0, 5e-5, 0.9, 0.999, 1e-8, 0.01, 1, nullptr));
  EXPECT_ARG_ERROR(sl_adamw_step((const sl_adamw_tensor*)ws, (const int64_t*)ws, 1, 1, 5e-5, 1.5, 0.999, 1e-8, 0.01, 1, nullptr));    // beta1 >= 1
  EXPECT_ARG_ERROR(sl_adamw_step((const sl_adamw_tensor*)ws, (const int64_t*)ws, 1, 1, 5e-5, 0.9, 0.999, 1e-8, 0.01, 0, nullptr));    // step counts from 1
  EXPECT(sl_adamw_blocks(0) == 0 && sl_adamw_blocks(1) == 1 && sl_adamw_blocks(4096) == 1 && sl_adamw_blocks(4097) == 2, "adamw block count");
  EXPECT(sl_layernorm_bwd_ws_bytes(7984, 1024) > 0 && sl_layernorm_bwd_ws_bytes(7984, 2048) == 0 && sl_layernorm_bwd_ws_bytes(0, 1024) == 0, "layernorm backward scratch size");
  EXPECT_ARG_ERROR(sl_layernorm_bwd_ws(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 16, 1024, 1e-5f, 0, SL_BF16, nullptr, 0, nullptr));
  EXPECT_ARG_ERROR(sl_greedy_select_partial(nullptr, nullptr, 2004, 1024, eos, 3, 128001, 1, 1, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 8, nullptr));
  EXPECT_ARG_ERROR(sl_greedy_select_partial((const float*)ws, (const int32_t*)ws, 2004, 8, eos, 9, 128001, 1, 1, ctx, ctx, ctx, ctx, ctx, out_ids, 8, nullptr));   // > 8 eos ids
  {
    sl_gemm_args ga;
    memset(&ga, 0, sizeof(ga));
    ga.A = ws; ga.W = ws; ga.lda = 64; ga.ldw = 64; ga.M = 32; ga.N = 128; ga.K = 64; ga.batch = 1; ga.dtype = SL_BF16; ga.out_f32 = 1;
    sl_gemm_ex_args gx;
    memset(&gx, 0, sizeof(gx));
    gx.w_mod = 1; gx.amax_val = (float*)ws;                         // amax_idx missing
    EXPECT_ARG_ERROR(sl_gemm_ex(&ga, &gx, nullptr));
    gx.amax_idx = (int32_t*)ws;                                     // M <= 64: the fused top-1 lives in the tiled kernels
    EXPECT_ARG_ERROR(sl_gemm_ex(&ga, &gx, nullptr));
    // training-tape epilogue fusions (ABI 7): every refused combination comes back as an argument error, before any launch
    memset(&gx, 0, sizeof(gx));
    gx.w_mod = 1; ga.M = 256; ga.out_f32 = 0; ga.C = ws; ga.ldc = 128;
    gx.post_op = 9;                                                 // unknown post-op
    EXPECT_ARG_ERROR(sl_gemm_ex(&ga, &gx, nullptr));
    gx.post_op = SL_POST_DROPOUT; gx.drop_p = 0.1f; gx.drop_ld = 0; // a mask needs its index stride
    EXPECT_ARG_ERROR(sl_gemm_ex(&ga, &gx, nullptr));
    gx.drop_ld = 128; gx.drop_p = 1.5f;                             // p outside [0, 1)
    EXPECT_ARG_ERROR(sl_gemm_ex(&ga, &gx, nullptr));
    gx.drop_p = 0.1f; gx.trans_w = 1;                               // transposed operands keep the plain epilogue
    EXPECT_ARG_ERROR(sl_gemm_ex(&ga, &gx, nullptr));
    gx.trans_w = 0; gx.post_op = SL_POST_GELU_BWD; gx.post_in = nullptr;   // GELU' without the saved pre-activation
    EXPECT_ARG_ERROR(sl_gemm_ex(&ga, &gx, nullptr));
    gx.post_op = SL_POST_SILU_MUL_BWD; gx.post_in = ws; gx.post_ld = 256; gx.drop_p = 0.f;     // ldc must span the 2 N-wide output
    EXPECT_ARG_ERROR(sl_gemm_ex(&ga, &gx, nullptr));
    gx.post_op = SL_POST_NONE; gx.colsum_out = (float*)ws; ga.M = 32;      // M <= 64: post-ops / colsum_out live in the tiled kernels
    EXPECT_ARG_ERROR(sl_gemm_ex(&ga, &gx, nullptr));
    gx.trans_a = gx.trans_w = 1; ga.M = 192; ga.N = 128; ga.K = 512;        // the bias-gradient rider needs the token-major kernel's shapes (M % 128)
    EXPECT_ARG_ERROR(sl_gemm_ex(&ga, &gx, nullptr));
  }
  EXPECT_ARG_ERROR(sl_col2im_batch(nullptr, nullptr, nullptr, 4, 100, 64, 3, 2, SL_BF16, nullptr));
  EXPECT_ARG_ERROR(sl_col2im_batch(ws, ws, (const int64_t*)ws, 4, 100, 60, 3, 2, SL_BF16, nullptr));            // C not a multiple of 8
  EXPECT_ARG_ERROR(sl_avgpool_bwd_batch(ws, ws, (const int64_t*)ws, 0, 100, 128, 8, 4, SL_BF16, nullptr));      // no utterances
  EXPECT_ARG_ERROR(sl_avgpool_bwd_batch(ws, ws, (const int64_t*)ws, 2, 100, 130, 8, 4, SL_BF16, nullptr));      // H not a multiple of 8
  EXPECT(sl_decode_graph_cache_clear() == 0, "nothing cached on this thread");

  {   // collective group: argument checks only (no device here)
    sl_comm c = nullptr;
    unsigned char id[SL_COMM_ID_BYTES] = {0};
    EXPECT(sl_comm_unique_id(nullptr) == SL_ERR_ARG, "sl_comm_unique_id(null)");
    EXPECT(sl_comm_init(nullptr, id, 0, 1) == SL_ERR_ARG, "sl_comm_init(null out)");
    EXPECT(sl_comm_init(&c, id, 2, 2) == SL_ERR_ARG && c == nullptr, "sl_comm_init(rank >= world)");
    EXPECT(sl_allreduce_sum(nullptr, id, 4, SL_F32, nullptr) == SL_ERR_ARG, "sl_allreduce_sum(no communicator)");
    EXPECT(sl_comm_destroy(nullptr) == 0 && sl_comm_rank(nullptr) == -1 && sl_comm_world(nullptr) == -1, "sl_comm_destroy(null) is a no-op");
  }
  if (g_fail == 0) printf("argcheck ok\n");
  return g_fail == 0 ? 0 : 1;
}
