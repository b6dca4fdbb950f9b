"""ctypes binding of libspeechllm.so (include/speechllm.h).

The product path has NO CPU fallback: if the shared library is missing or an entry point fails, an
exception is raised.  PyTorch is used only to own device memory and streams; raw pointers cross the
C ABI.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree build.  SL_LIB_PATH (A/B runs of two builds, tools/) is honoured only together with SL_DEV=1, and
# bench.py records the resolved path in its JSON line (`native_library`), so a foreign build can never pass for the in-tree one.
LIB_PATH = (os.environ.get("SL_LIB_PATH") if os.environ.get("SL_DEV") == "1" else None) or os.path.join(_HERE, "libspeechllm.so")

SL_F32, SL_BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_SILU_MUL, ACT_ROPE_KV = 0, 1, 2, 3
POST_NONE, POST_DROPOUT, POST_GELU_BWD, POST_SILU_MUL_BWD = 0, 1, 2, 3      # sl_gemm_ex_args.post_op
W_ROWMAJOR, W_PACKED = 0, 1
COMM_ID_BYTES = 128          # SL_COMM_ID_BYTES = sizeof(ncclUniqueId)
MAX_DECODE_BATCH = 2048      # SL_MAX_DECODE_BATCH: sequences per generate call / rows per decode step

c_i32, c_i64, c_f32, c_vp, c_sz = C.c_int32, C.c_int64, C.c_float, C.c_void_p, C.c_size_t


class SpeechLLMError(RuntimeError):
    pass


class GemmArgs(C.Structure):
    _fields_ = [("A", c_vp), ("lda", c_i64), ("strideA", c_i64),
                ("W", c_vp), ("ldw", c_i64), ("strideW", c_i64),
                ("C", c_vp), ("ldc", c_i64), ("strideC", c_i64),
                ("bias", c_vp), ("strideBias", c_i64),
                ("residual", c_vp), ("ldr", c_i64), ("strideR", c_i64),
                ("M", c_i32), ("N", c_i32), ("K", c_i32), ("batch", c_i32),
                ("dtype", c_i32), ("act", c_i32), ("out_f32", c_i32), ("w_layout", c_i32)]


class GemmFused(C.Structure):
    _fields_ = [("fuse_rms", c_i32), ("rms_eps", c_f32), ("rope_cos", c_vp), ("rope_sin", c_vp), ("tok_pos", c_vp), ("tok_seq", c_vp),
                ("k_cache", c_vp), ("v_cache", c_vp), ("n_heads", c_i32), ("n_kv_heads", c_i32), ("max_ctx", c_i32), ("reserved", c_i32),
                ("split_ws", c_vp), ("split_ws_bytes", C.c_size_t), ("rstd_in", c_vp), ("rstd_out", c_vp), ("norm_out", c_vp), ("norm_gain", c_vp)]


class GemmEx(C.Structure):
    _fields_ = [("trans_a", c_i32), ("trans_w", c_i32), ("residual_f32", c_i32), ("w_mod", c_i32), ("aux_out", c_vp), ("groups", c_vp),
                ("groups_ext", c_i32), ("reserved", c_i32), ("amax_val", c_vp), ("amax_idx", c_vp),
                ("ln_mr", c_vp), ("ln_u", c_vp), ("ln_c", c_vp), ("stats_out", c_vp), ("sk_ws", c_vp), ("sk_ws_bytes", C.c_size_t),
                # training-tape epilogue fusions (ABI 7)
                ("post_op", c_i32), ("post_reserved", c_i32), ("drop_p", C.c_float), ("post_reserved_f", C.c_float), ("drop_seed", C.c_uint64),
                ("drop_ld", C.c_int64), ("post_in", c_vp), ("post_ld", C.c_int64), ("colsum_out", c_vp), ("deferred_splits", c_vp)]


class AttnArgs(C.Structure):
    _fields_ = [("q", c_vp), ("q_row_stride", c_i64), ("q_head_stride", c_i64),
                ("k", c_vp), ("k_row_stride", c_i64), ("k_head_stride", c_i64),
                ("v", c_vp), ("v_row_stride", c_i64), ("v_head_stride", c_i64),
                ("out", c_vp), ("o_row_stride", c_i64), ("o_head_stride", c_i64),
                ("cu_q", c_vp), ("cu_k", c_vp), ("klen", c_vp),
                ("nseq", c_i32), ("max_qlen", c_i32), ("n_heads", c_i32), ("n_kv_heads", c_i32),
                ("head_dim", c_i32), ("causal", c_i32), ("dtype", c_i32), ("reserved", c_i32),
                ("scale", c_f32), ("dropout_p", c_f32), ("dropout_seed", C.c_uint64), ("lse", c_vp)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("q", c_vp), ("q_row_stride", c_i64), ("q_head_stride", c_i64),
                ("k", c_vp), ("k_row_stride", c_i64), ("k_head_stride", c_i64),
                ("v", c_vp), ("v_row_stride", c_i64), ("v_head_stride", c_i64),
                ("out", c_vp), ("o_row_stride", c_i64), ("o_head_stride", c_i64),
                ("d_out", c_vp), ("do_row_stride", c_i64), ("do_head_stride", c_i64),
                ("dq", c_vp), ("dq_row_stride", c_i64), ("dq_head_stride", c_i64),
                ("dk", c_vp), ("dk_row_stride", c_i64), ("dk_head_stride", c_i64),
                ("dv", c_vp), ("dv_row_stride", c_i64), ("dv_head_stride", c_i64),
                ("lse", c_vp), ("delta", c_vp), ("cu_q", c_vp), ("cu_k", c_vp), ("klen", c_vp), ("n_tok_q", c_i64),
                ("nseq", c_i32), ("max_qlen", c_i32), ("max_klen", c_i32), ("n_heads", c_i32), ("n_kv_heads", c_i32),
                ("head_dim", c_i32), ("causal", c_i32), ("dtype", c_i32),
                ("scale", c_f32), ("dropout_p", c_f32), ("dropout_seed", C.c_uint64)]


class HubertLayer(C.Structure):
    _fields_ = [(n, c_vp) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b",
                                    "w1", "b1", "w2", "b2")]


class HubertFold(C.Structure):
    _fields_ = [(n, c_vp) for n in ("wqkv_f", "uqkv", "cqkv", "w1_f", "u1", "c1")]


class HubertModel(C.Structure):
    _fields_ = [("dtype", c_i32), ("n_conv", c_i32), ("hidden", c_i32), ("n_layers", c_i32),
                ("n_heads", c_i32), ("ffn", c_i32), ("pos_k", c_i32), ("pos_groups", c_i32),
                ("conv_dim", c_i32 * 8), ("conv_kernel", c_i32 * 8), ("conv_stride", c_i32 * 8),
                ("ln_eps", c_f32),
                ("pool_kernel", c_i32), ("pool_stride", c_i32), ("llm_dim", c_i32), ("reserved", c_i32),
                ("conv0_w", c_vp), ("conv0_b", c_vp), ("conv0_g", c_vp), ("conv0_beta", c_vp),
                ("conv_w", c_vp * 8), ("conv_b", c_vp * 8), ("conv_g", c_vp * 8), ("conv_beta", c_vp * 8),
                ("fp_ln_g", c_vp), ("fp_ln_b", c_vp), ("fp_w", c_vp), ("fp_b", c_vp),
                ("pos_w", c_vp), ("pos_b", c_vp),
                ("layers", C.POINTER(HubertLayer)),
                ("final_ln_g", c_vp), ("final_ln_b", c_vp),
                ("proj_w", c_vp), ("proj_b", c_vp),
                ("fold", C.POINTER(HubertFold))]


class LlamaLayer(C.Structure):
    _fields_ = [(n, c_vp) for n in ("norm1", "wqkv", "wo", "norm2", "wgu", "wdown", "wqkv_dec", "wo_dec", "wgu_dec", "wdown_dec")]


class LlamaModel(C.Structure):
    _fields_ = [("dtype", c_i32), ("hidden", c_i32), ("n_layers", c_i32), ("n_heads", c_i32),
                ("n_kv_heads", c_i32), ("head_dim", c_i32), ("ffn", c_i32), ("vocab", c_i32),
                ("rms_eps", c_f32), ("rope_len", c_i32),
                ("embed", c_vp), ("lm_head", c_vp), ("final_norm", c_vp),
                ("rope_cos", c_vp), ("rope_sin", c_vp),
                ("layers", C.POINTER(LlamaLayer)),
                ("lm_head_dec", c_vp), ("dec_fused_norm", c_i32), ("reserved", c_i32)]


class EncStackCfg(C.Structure):
    _fields_ = [("dtype", c_i32), ("hidden", c_i32), ("n_heads", c_i32), ("ffn", c_i32), ("n_layers", c_i32), ("nseq", c_i32), ("max_len", c_i32),
                ("reserved", c_i32), ("n_tok", c_i64), ("ln_eps", c_f32), ("p_hidden", c_f32), ("p_act", c_f32), ("p_attn", c_f32),
                ("cu", c_vp), ("klen", c_vp), ("skip", c_vp), ("seeds", c_vp)]


class EncLayerSaved(C.Structure):
    _fields_ = [(n, c_vp) for n in ("x", "ln1", "qkv", "att", "x_mid", "ln2", "pre1", "mid", "x_out", "lse")]


class EncLayerGrads(C.Structure):
    _fields_ = [(n, c_vp) for n in ("ln1_g", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_g", "ln2_b", "w1", "b1", "w2", "b2")]


class LlamaStackCfg(C.Structure):
    _fields_ = [("dtype", c_i32), ("hidden", c_i32), ("n_heads", c_i32), ("n_kv_heads", c_i32), ("head_dim", c_i32), ("ffn", c_i32),
                ("n_layers", c_i32), ("nseq", c_i32), ("max_len", c_i32), ("reserved", c_i32), ("n_tok", c_i64), ("rms_eps", c_f32),
                ("cu", c_vp), ("klen", c_vp), ("pos", c_vp), ("rope_cos", c_vp), ("rope_sin", c_vp)]


class LlamaTrainLayer(C.Structure):
    _fields_ = [(n, c_vp) for n in ("norm1", "wqkv", "wo", "norm2", "wgu", "wdown", "wqkv_t", "wo_t", "wgu_t", "wdown_t")]


class LlamaLayerSaved(C.Structure):
    _fields_ = [(n, c_vp) for n in ("qkv", "x2", "gu", "att", "lse")]


class AdamWTensor(C.Structure):
    _fields_ = [("p", c_vp), ("g", c_vp), ("m", c_vp), ("v", c_vp), ("dst", c_vp), ("n", c_i64), ("dst_dtype", c_i32), ("reserved", c_i32)]


class KVCache(C.Structure):
    _fields_ = [("k_cache", c_vp), ("v_cache", c_vp), ("slots", c_i32), ("max_ctx", c_i32), ("shared_prefix", c_i32), ("reserved", c_i32)]


class GenerateOpts(C.Structure):
    _fields_ = [("eos_ids_host", C.POINTER(c_i32)), ("row_limits_host", C.POINTER(c_i32)), ("seed", C.c_uint64), ("max_new_tokens", c_i32),
                ("n_eos", c_i32), ("pad_id", c_i32), ("use_eos", c_i32), ("check_every", c_i32), ("sample", c_i32), ("temperature", c_f32),
                ("top_k", c_i32), ("top_p", c_f32), ("compact", c_i32)]


class GenerateStats(C.Structure):
    _fields_ = [("row_steps", c_i64), ("n_steps", c_i32), ("decode_launches", c_i32), ("compactions", c_i32), ("final_rows", c_i32),
                ("prefill_ms", c_f32), ("decode_ms", c_f32)]


_PROTOS = {
    "sl_last_error": (C.c_char_p, []),
    "sl_version": (c_i32, []),
    "sl_device_arch": (c_i32, [C.c_char_p, c_i32]),
    "sl_tuning_reload": (c_i32, []),
    "sl_decode_graph_cache_clear": (c_i32, []),
    "sl_gemm": (c_i32, [C.POINTER(GemmArgs), c_vp]),
    "sl_gemm_streamk_workspace_bytes": (C.c_size_t, []),
    "sl_layernorm_fold_build": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "sl_weight_norm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "sl_weight_norm_bwd_workspace_bytes": (C.c_size_t, [c_i32]),
    "sl_comm_unique_id": (c_i32, [c_vp]),
    "sl_comm_init": (c_i32, [C.POINTER(c_vp), c_vp, c_i32, c_i32]),
    "sl_allreduce_sum": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp]),
    "sl_comm_destroy": (c_i32, [c_vp]),
    "sl_comm_abort": (c_i32, [c_vp]),
    "sl_comm_rank": (c_i32, [c_vp]),
    "sl_comm_world": (c_i32, [c_vp]),
    "sl_pack_weight": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "sl_gemm_fused_decode": (c_i32, [C.POINTER(GemmArgs), C.POINTER(GemmFused), c_vp]),
    "sl_gemm_split_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "sl_gemm_split_count": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "sl_attn_decode_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "sl_attn_decode_split": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "sl_gemm_ex": (c_i32, [C.POINTER(GemmArgs), C.POINTER(GemmEx), c_vp]),
    "sl_gemm_ln_fold_ok": (c_i32, [c_i32, c_i32, c_i32, c_i32]),
    "sl_layernorm_stats_finalize": (c_i32, [c_vp, c_i32, c_i64, c_i32, c_f32, c_vp, c_vp]),
    "sl_layernorm_stats": (c_i32, [c_vp, c_i64, c_i32, c_f32, c_vp, c_i32, c_vp]),
    "sl_gelu_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp]),
    "sl_axpby": (c_i32, [c_vp, c_vp, c_f32, c_f32, c_i64, c_i32, c_vp]),
    "sl_adamw_blocks": (C.c_size_t, [c_i64]),
    "sl_adamw_step": (c_i32, [c_vp, c_vp, c_i32, c_i64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, c_i64, c_vp]),
    "sl_attn_dropout_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_i32, c_i32, c_i32, c_f32, C.c_uint64, c_i32, c_vp]),
    "sl_transpose_pad": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sl_dropout": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_f32, C.c_uint64, c_i32, c_vp]),
    "sl_silu_mul": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "sl_silu_mul_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "sl_rope_inplace": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sl_layernorm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_i32, c_i32, c_vp]),
    "sl_layernorm_bwd_ws_bytes": (c_sz, [c_i64, c_i32]),
    "sl_layernorm_bwd_ws": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_i32, c_i32, c_vp, c_sz, c_vp]),
    "sl_rmsnorm_bwd": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_i32, c_vp]),
    "sl_colsum": (c_i32, [c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "sl_softmax_rows": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i64, c_f32, c_i32, c_i32, c_vp]),
    "sl_softmax_bwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i64, c_f32, c_i32, c_vp]),
    "sl_softmax_rows_var": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_vp, c_i64, c_f32, c_i32, c_i32, c_vp]),
    "sl_softmax_bwd_var": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_i64, c_f32, c_i32, c_vp]),
    "sl_ce_loss": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "sl_soft_ce_loss": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_f32, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "sl_mse_loss": (c_i32, [c_vp, c_vp, c_i64, c_f32, c_vp, c_vp, c_i32, c_i32, c_vp]),
    "sl_kd_logit_losses": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_i32, c_vp, c_i32, c_vp]),
    "sl_kd_mse_rows": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, c_i32, c_i32, c_vp, c_i32, c_vp]),
    "sl_avgpool_bwd": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i64, c_i32, c_vp]),
    "sl_col2im": (c_i32, [c_vp, c_vp, c_i64, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sl_col2im_batch": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sl_avgpool_bwd_batch": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sl_hubert_conv0_bwd_batch": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp,
                                          c_vp, c_i32, c_vp]),
    "sl_hubert_conv0_bwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp, c_vp,
                                    c_i32, c_vp]),
    "sl_layernorm": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_i32, c_i32, c_vp]),
    "sl_rmsnorm": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_f32, c_i32, c_vp]),
    "sl_hubert_conv0": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "sl_posconv_stage": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sl_avgpool_rows": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp, c_i64, c_i32, c_vp]),
    "sl_hubert_conv0_batch": (c_i32, [c_vp, c_vp, c_vp, c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "sl_posconv_stage_batch": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sl_avgpool_batch": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sl_embed_gather": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_vp]),
    "sl_attn_fwd": (c_i32, [C.POINTER(AttnArgs), c_vp]),
    "sl_attn_bwd": (c_i32, [C.POINTER(AttnBwdArgs), c_vp]),
    "sl_rope_kv_append": (c_i32, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "sl_attn_decode": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_vp]),
    "sl_greedy_select_partial": (c_i32, [c_vp, c_vp, c_i32, c_i32, C.POINTER(c_i32), c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp]),
    "sl_greedy_select": (c_i32, [c_vp, c_i32, c_i32, C.POINTER(c_i32), c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                 c_i32, c_vp]),
    "sl_encoder_stack_train_workspace_bytes": (c_sz, [C.POINTER(EncStackCfg)]),
    "sl_encoder_stack_train_fwd": (c_i32, [C.POINTER(HubertLayer), C.POINTER(EncStackCfg), c_vp, C.POINTER(EncLayerSaved), C.POINTER(c_vp), c_vp, c_sz, c_vp]),
    "sl_encoder_stack_train_bwd": (c_i32, [C.POINTER(HubertLayer), C.POINTER(EncStackCfg), C.POINTER(EncLayerSaved), C.POINTER(EncLayerGrads), c_i32, c_i32,
                                           c_vp, c_vp, c_sz, c_vp]),
    "sl_llama_stack_train_workspace_bytes": (c_sz, [C.POINTER(LlamaStackCfg)]),
    "sl_llama_stack_train_fwd": (c_i32, [C.POINTER(LlamaTrainLayer), C.POINTER(LlamaStackCfg), C.POINTER(c_vp), C.POINTER(LlamaLayerSaved), c_vp, c_sz, c_vp]),
    "sl_llama_stack_train_bwd": (c_i32, [C.POINTER(LlamaTrainLayer), C.POINTER(LlamaStackCfg), C.POINTER(c_vp), C.POINTER(LlamaLayerSaved), C.POINTER(c_vp),
                                         c_vp, c_vp, c_sz, c_vp]),
    "sl_hubert_workspace_bytes": (c_sz, [C.POINTER(HubertModel), C.POINTER(c_i64), c_i32]),
    "sl_hubert_num_frames": (c_i32, [C.POINTER(HubertModel), c_i64]),
    "sl_hubert_forward": (c_i32, [C.POINTER(HubertModel), c_vp, C.POINTER(c_i64), c_i32, c_vp, c_i64, C.POINTER(c_i64), c_vp,
                                  c_vp, c_sz, c_vp]),
    "sl_whisper_logmel_workspace_bytes": (c_sz, [c_i32, c_i32, c_i32, c_i32]),
    "sl_whisper_logmel": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_sz, c_i32, c_vp]),
    "sl_whisper_workspace_bytes": (c_sz, [C.POINTER(HubertModel), c_i32]),
    "sl_whisper_forward": (c_i32, [C.POINTER(HubertModel), c_vp, c_i32, c_vp, c_i64, C.POINTER(c_i64), c_vp, c_vp, c_sz, c_vp]),
    "sl_llama_workspace_bytes": (c_sz, [C.POINTER(LlamaModel), c_i64, c_i32]),
    "sl_llama_prefill": (c_i32, [C.POINTER(LlamaModel), C.POINTER(KVCache), c_vp, C.POINTER(c_i32), c_i32, c_vp, c_vp, c_vp,
                                 c_vp, c_sz, c_vp]),
    "sl_llama_decode_step": (c_i32, [C.POINTER(LlamaModel), C.POINTER(KVCache), c_vp, c_vp, c_i32, c_vp, c_vp, c_sz, c_vp]),
    "sl_generate_workspace_bytes": (c_sz, [C.POINTER(LlamaModel), c_i64, c_i32, c_i32]),
    "sl_generate": (c_i32, [C.POINTER(LlamaModel), C.POINTER(KVCache), c_vp, C.POINTER(c_i32), c_i32, C.POINTER(GenerateOpts), C.POINTER(c_i32),
                            C.POINTER(GenerateStats), c_vp, c_sz, c_vp]),
    "sl_sample_select": (c_i32, [c_vp, c_i32, c_i32, c_f32, c_i32, c_f32, C.c_uint64, C.POINTER(c_i32), c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp,
                                 c_i32, c_vp, c_vp]),
    "sl_sample_generate": (c_i32, [C.POINTER(LlamaModel), C.POINTER(KVCache), c_vp, C.POINTER(c_i32), c_i32, c_i32, C.POINTER(c_i32), c_i32, c_i32, c_i32,
                                   c_i32, c_f32, c_i32, c_f32, C.c_uint64, C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_f32), c_vp, c_sz, c_vp]),
    "sl_greedy_generate": (c_i32, [C.POINTER(LlamaModel), C.POINTER(KVCache), c_vp, C.POINTER(c_i32), c_i32, c_i32,
                                   C.POINTER(c_i32), c_i32, c_i32, c_i32, c_i32, C.POINTER(c_i32), C.POINTER(c_i32),
                                   C.POINTER(c_f32), c_vp, c_sz, c_vp]),
}

EXPORTS = tuple(_PROTOS)
_lib = None


def lib():
    """Load the shared library once; fail loudly if it is absent (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SpeechLLMError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C llm-speech-summarization_amd/csrc`). There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOS.items():
            fn = getattr(l, name)  # AttributeError if an export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().sl_last_error().decode("utf-8", "replace")
        raise SpeechLLMError(f"{what or 'libspeechllm'} failed ({rc}): {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return SL_F32
    if dt == torch.bfloat16:
        return SL_BF16
    raise SpeechLLMError(f"unsupported dtype {dt}: the HIP path computes in float32 or bfloat16")


def ptr(t) -> int:
    """Device pointer of a tensor (None -> NULL)."""
    return 0 if t is None else t.data_ptr()


def h2d(data, dtype, device) -> torch.Tensor:
    """Host values -> device tensor WITHOUT stalling the stream: torch.tensor(list, device='cuda') copies from pageable memory, which
    blocks the host until everything queued on the stream has run (measured on the per-rank KD window: 19 such uploads per window = 18 of
    its 36 ms spent waiting, the GPU idling while the host refills the queue behind each).  Here the values go through PyTorch's pinned
    host allocator and an asynchronous copy on the current stream; the allocator keeps the pinned block alive until the copy has run."""
    t = data if isinstance(data, torch.Tensor) else torch.tensor(data, dtype=dtype)
    if t.dtype != dtype:
        t = t.to(dtype)
    device = torch.device(device)
    if device.type != "cuda":
        return t.to(device)
    return t.contiguous().pin_memory().to(device, non_blocking=True)


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_gpu(t: torch.Tensor, name: str = "tensor") -> None:
    if not t.is_cuda:
        raise SpeechLLMError(f"{name} must live on the GPU: the hot path is HIP-only")
    if not t.is_contiguous():
        raise SpeechLLMError(f"{name} must be contiguous")
