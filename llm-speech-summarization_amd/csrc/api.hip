// api.hip — error reporting and library identity for libspeechllm.
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "common.h"

static thread_local char g_err[512] = "";

void sl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* sl_last_error(void) { return g_err; }

extern "C" int sl_version(void) { return SL_ABI_VERSION; }   // 7: sl_gemm_ex_args.post_op / drop_* / post_in / colsum_out (training-tape epilogue fusions), sl_comm_world asks RCCL; 6: sl_comm_abort, SL_MAX_DECODE_BATCH 2048, sl_gemm_fused.norm_out / norm_gain (grew that struct in round 4), sl_generate_opts (compaction); 5: sl_kv_cache.shared_prefix; 4: sl_gemm_ex_args.sk_ws / sk_ws_bytes, sl_gemm_streamk_workspace_bytes; 3: sl_gemm_ex_args.amax_*, sl_greedy_select_partial, sl_adamw_step, sl_layernorm_bwd_ws, sl_decode_graph_cache_clear

extern "C" int sl_device_arch(char* buf, int n) {
  SL_CHECK_ARG(buf != nullptr && n > 0, "sl_device_arch: bad buffer");
  int dev = 0;
  SL_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  SL_HIP(hipGetDeviceProperties(&prop, dev));
  snprintf(buf, (size_t)n, "%s", prop.gcnArchName);
  return 0;
}

// ----------------------------------------------------------------------------------------------
// tuning switches (common.h SlEnv)
// ----------------------------------------------------------------------------------------------
// Readers take an immutable snapshot through one atomic pointer; a reload publishes a NEW snapshot and never touches a
// published one (the old snapshots stay allocated: a handful of 80-byte tables per process), so a dispatching thread can never
// see a half-written table.
static std::atomic<const SlEnv*> g_env{nullptr};
static std::mutex g_env_mu;

static int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}

static const SlEnv* env_load() {
  SlEnv& e = *new SlEnv();
  // 26: the decode step at 17..32 sequences on either family (tools/time_decode_step.py, profiles/r04_zb_stream_min_m.txt): the skinny kernels win
  // up to 26 rows (2.34 vs 2.36 ms), the 32-row streaming blocks from 27 (2.42 vs 2.38) to 32 (2.57 vs 2.40)
  e.compact_pin = env_int("SL_COMPACT_PIN", 1);
  e.tape_fuse = env_int("SL_TAPE_FUSE", 1);
  e.attn_bwd_kf = env_int("SL_ATTN_BWD_KF", 0);
  e.decode_prefetch = env_int("SL_DECODE_PREFETCH", 0);
  e.attn_fwd_st = env_int("SL_ATTN_FWD_ST", 1);
  e.conv0_fold = env_int("SL_CONV0_FOLD", 1);
  e.enc_wt_ahead = env_int("SL_ENC_WT_AHEAD", 1);
  e.rms_bwd_lean = env_int("SL_RMSBWD_LEAN", 2);
  e.glds_ring = env_int("SL_GLDS_RING", 4);
  if (e.glds_ring != 3 && e.glds_ring != 4 && e.glds_ring != 104 && e.glds_ring != 204) e.glds_ring = 0;
  e.splitk_slots = env_int("SL_SPLITK_SLOTS", 0);
  e.glds_dmab = env_int("SL_GLDS_DMAB", 0);
  e.tt_ring = env_int("SL_TT_RING", 1);
  e.split_k256 = env_int("SL_SPLIT_K256", 1);
  e.ring_max_tiles = env_int("SL_GLDS_RING_MAX_TILES", 0);
  e.attn_bwd_both = env_int("SL_ATTN_BWD_BOTH", 1);
  e.wgrad_stream_min_tok = env_int("SL_WGRAD_STREAM_MIN_TOK", 0);
  e.tt_batched = env_int("SL_TT_BATCHED", 1);
  e.ln_colred_inkernel = env_int("SL_LN_COLRED_INKERNEL", 0);
  e.stream_min_m = env_int("SL_STREAM_MIN_M", 26);
  if (e.stream_min_m < 16) e.stream_min_m = 26;
  e.disable_t256 = getenv("SL_DISABLE_T256") != nullptr;
  e.t256_min_tiles = env_int("SL_T256_MIN_TILES", 512);
  e.t256_min_k = env_int("SL_T256_MIN_K", 1024);
  e.t256_phased = env_int("SL_T256_PHASED", 1);
  e.t256_by_rounds_pad = env_int("SL_T256_BY_ROUNDS_PAD", 1);
  e.decode_tiled = env_int("SL_DECODE_TILED", 1);
  e.stream_k = env_int("SL_STREAM_K", 1);
  e.split_k = env_int("SL_SPLIT_K", 1);
  e.wgrad_tr = env_int("SL_WGRAD_TR", 1);
  e.tt_max_splits = env_int("SL_TT_MAX_SPLITS", 8);
  e.lnbwd_nw = env_int("SL_LNBWD_NW", 16);
  e.skinny_alt = env_int("SL_SKINNY_ALT", 0);
  e.prefill_share_prefix = env_int("SL_PREFILL_SHARE_PREFIX", 1);
  e.gemm_ko = env_int("SL_GEMM_KO", 0);
  { const char* sp = getenv("SL_GEMM_STAMP_PTR"); e.gemm_stamp_ptr = (sp && sp[0]) ? strtoull(sp, nullptr, 16) : 0ull; }
  const char* g = getenv("SL_DISABLE_GLDS");
  e.disable_glds = (g && g[0] == '1') ? 1 : ((g && g[0] == '2') ? 2 : 0);
  e.direct_epilogue = env_int("SL_DIRECT_EPILOGUE", 0);
  e.gemm_gm = env_int("SL_GEMM_GM", 8);
  e.attn_full_min = env_int("SL_ATTN_FULL_MIN", 32);
  e.attn_force_split = getenv("SL_ATTN_FORCE_SPLIT") != nullptr;
  { const int ks = env_int("SL_ATTN_DECODE_KS", 128); e.attn_decode_ks = (ks == 64 || ks == 65) ? ks : 128; }      // 65 = 64-key chunks + register prefetch of the next chunk
  e.attn_split_merge = env_int("SL_ATTN_SPLIT_MERGE", -1);
  e.attn_generic = env_int("SL_ATTN_GENERIC", 0);
  e.attn_qt = env_int("SL_ATTN_QT", 0);
  e.norm_single_row = env_int("SL_NORM_SINGLE_ROW", 0);
  e.no_ln_fold = env_int("SL_NO_LN_FOLD", 0);
  e.no_swap_epilogue = env_int("SL_NO_SWAP_EPILOGUE", 0);
  e.gemm_log = env_int("SL_GEMM_LOG", 0);
  e.no_wgrad_stream = env_int("SL_NO_WGRAD_STREAM", 0);
  e.stream_splits = e.stream_nwv = e.stream_mt = 0;
  e.stream_nl = env_int("SL_STREAM_NL", 0);
  e.stream_wide = env_int("SL_STREAM_WIDE", 1);
  e.stream_wsplits = env_int("SL_STREAM_WSPLITS", 0);
  e.stream_fixup = env_int("SL_STREAM_FIXUP", 0);
  const char* sc = getenv("SL_STREAM_CFG");
  if (sc && sc[0]) {
    int sp = 0, nwv = 0, mt = 0;
    const int n = sscanf(sc, "%d,%d,%d", &sp, &nwv, &mt);
    if (n >= 2) {
      if (sp >= 1 && sp <= 64) e.stream_splits = sp;
      if (nwv == 2 || nwv == 4) e.stream_nwv = nwv;
    }
    if (n == 3 && (mt == 8 || mt == 16)) e.stream_mt = mt;
  }
  return &e;
}

const SlEnv& sl_env() {
  const SlEnv* e = g_env.load(std::memory_order_acquire);
  if (!e) {
    std::lock_guard<std::mutex> lk(g_env_mu);
    e = g_env.load(std::memory_order_relaxed);
    if (!e) {
      e = env_load();
      g_env.store(e, std::memory_order_release);
    }
  }
  return *e;
}

static thread_local int g_family_rows = 0;
int sl_family_rows(int rows) { return g_family_rows > rows ? g_family_rows : rows; }
int sl_family_pin(int rows) {
  const int prev = g_family_rows;
  g_family_rows = rows > 0 ? rows : 0;
  return prev;
}

extern "C" int sl_tuning_reload(void) {
  std::lock_guard<std::mutex> lk(g_env_mu);
  g_env.store(env_load(), std::memory_order_release);
  return 0;
}
