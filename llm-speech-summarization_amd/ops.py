"""Thin tensor-level wrappers over the per-op C-ABI entry points (used by the kernel parity tests and by
host code that composes ops outside the whole-model entry points).  Every function launches HIP kernels
on the current torch stream; nothing here computes on the CPU.
"""
from __future__ import annotations

import os
import ctypes as C
from typing import Optional, Sequence

import torch

from . import _lib as L


def gemm(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
         act: int = L.ACT_NONE, out_f32: bool = False, *, M: Optional[int] = None, K: Optional[int] = None,
         lda: Optional[int] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """C = act(A W^T + bias) + residual.  `M/K/lda` override the view for implicit-GEMM convolutions."""
    L.require_gpu(A, "A"); L.require_gpu(W, "W")
    N = W.shape[0]
    K = W.shape[1] if K is None else K
    M = A.shape[0] if M is None else M
    lda = A.stride(0) if lda is None else lda
    n_out = N // 2 if act == L.ACT_SILU_MUL else N
    if out is None:
        out = torch.empty((M, n_out), device=A.device, dtype=torch.float32 if out_f32 else A.dtype)
    a = L.GemmArgs()
    a.A, a.lda, a.strideA = L.ptr(A), lda, 0
    a.W, a.ldw, a.strideW = L.ptr(W), W.stride(0), 0
    a.C, a.ldc, a.strideC = L.ptr(out), out.stride(0), 0
    a.bias, a.strideBias = L.ptr(bias), 0
    a.residual, a.ldr, a.strideR = L.ptr(residual), (residual.stride(0) if residual is not None else 0), 0
    a.M, a.N, a.K, a.batch = M, N, K, 1
    a.dtype, a.act, a.out_f32 = L.dtype_code(A.dtype), act, int(out_f32)
    L.check(L.lib().sl_gemm(C.byref(a), L.stream_ptr()), "sl_gemm")
    return out


def gemm_batched(args: "L.GemmArgs") -> None:
    L.check(L.lib().sl_gemm(C.byref(args), L.stream_ptr()), "sl_gemm")


def layernorm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float, gelu: bool = False) -> torch.Tensor:
    L.require_gpu(x, "x")
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    L.check(L.lib().sl_layernorm(L.ptr(x), L.ptr(y), L.ptr(g), L.ptr(b), rows, x.shape[-1], eps, int(gelu),
                                 L.dtype_code(x.dtype), L.stream_ptr()), "sl_layernorm")
    return y


def rmsnorm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    L.require_gpu(x, "x")
    y = torch.empty_like(x)
    rows = x.numel() // x.shape[-1]
    L.check(L.lib().sl_rmsnorm(L.ptr(x), L.ptr(y), L.ptr(w), rows, x.shape[-1], eps, L.dtype_code(x.dtype), L.stream_ptr()),
            "sl_rmsnorm")
    return y


def hubert_conv0(wave: torch.Tensor, w: torch.Tensor, bias, gamma, beta, dtype: torch.dtype, k: int = 10, stride: int = 5,
                 eps: float = 1e-5) -> torch.Tensor:
    L.require_gpu(wave, "wave")
    n = wave.numel()
    Cout = w.shape[0]
    Lout = (n - k) // stride + 1
    out = torch.empty((Lout, Cout), device=wave.device, dtype=dtype)
    L.check(L.lib().sl_hubert_conv0(L.ptr(wave), n, L.ptr(w), L.ptr(bias), L.ptr(gamma), L.ptr(beta), L.ptr(out), Cout, k, stride,
                                    eps, L.dtype_code(dtype), L.stream_ptr()), "sl_hubert_conv0")
    return out


def posconv_stage(x: torch.Tensor, groups: int, k: int) -> torch.Tensor:
    T, H = x.shape
    xg = torch.empty((groups, T + k, H // groups), device=x.device, dtype=x.dtype)
    L.check(L.lib().sl_posconv_stage(L.ptr(x), L.ptr(xg), T, H, groups, k, L.dtype_code(x.dtype), L.stream_ptr()), "sl_posconv_stage")
    return xg


def avgpool_rows(x: torch.Tensor, kernel: int = 8, stride: int = 4, ranges: Optional[torch.Tensor] = None) -> torch.Tensor:
    T, H = x.shape
    P = ranges.shape[0] if ranges is not None else (T - kernel) // stride + 1
    y = torch.empty((P, H), device=x.device, dtype=x.dtype)
    L.check(L.lib().sl_avgpool_rows(L.ptr(x), L.ptr(y), T, H, kernel, stride, L.ptr(ranges), P, L.dtype_code(x.dtype), L.stream_ptr()),
            "sl_avgpool_rows")
    return y


def hubert_conv0_batch(flat_waves: torch.Tensor, sample_offsets: Sequence[int], row_offsets: Sequence[int], w, bias, gamma, beta, dtype: torch.dtype,
                       k: int = 10, stride: int = 5, eps: float = 1e-5) -> torch.Tensor:
    """conv0 + LayerNorm + GELU of every utterance of a ragged batch in one launch -> packed (sum L_u, C) rows."""
    dev = flat_waves.device
    n_utt = len(sample_offsets) - 1
    desc = L.h2d([list(sample_offsets), list(row_offsets)], torch.int64, dev)
    out = torch.empty((row_offsets[-1], w.shape[0]), device=dev, dtype=dtype)
    max_L = max(row_offsets[u + 1] - row_offsets[u] for u in range(n_utt))
    L.check(L.lib().sl_hubert_conv0_batch(L.ptr(flat_waves), desc[0].data_ptr(), desc[1].data_ptr(), n_utt, max_L, L.ptr(w), L.ptr(bias), L.ptr(gamma),
                                          L.ptr(beta), L.ptr(out), w.shape[0], k, stride, eps, L.dtype_code(dtype), L.stream_ptr()), "sl_hubert_conv0_batch")
    return out


def posconv_stage_batch(x: torch.Tensor, seqlens: Sequence[int], groups: int, k: int) -> torch.Tensor:
    """sl_posconv_stage for every utterance of a packed batch in one launch; utterance u's (groups, T_u + k, H/groups) block
    starts (cu[u] + u k) * H elements into the flat result."""
    cu, klen = seq_descriptors(seqlens, x.device)
    H = x.shape[1]
    xg = torch.empty((x.shape[0] + len(seqlens) * k) * H, device=x.device, dtype=x.dtype)
    L.check(L.lib().sl_posconv_stage_batch(L.ptr(x), L.ptr(xg), cu.data_ptr(), klen.data_ptr(), len(seqlens), max(int(n) for n in seqlens), H, groups, k,
                                           L.dtype_code(x.dtype), L.stream_ptr()), "sl_posconv_stage_batch")
    return xg


def avgpool_batch(x: torch.Tensor, seqlens: Sequence[int], kernel: int, stride: int):
    """AvgPool1d over time of every utterance of a packed batch in one launch -> (packed pooled rows, P list)."""
    cu, klen = seq_descriptors(seqlens, x.device)
    H = x.shape[1]
    P = [(int(t) - kernel) // stride + 1 for t in seqlens]
    rec, row = [], 0
    for p_ in P:
        rec.append([p_, row * H, 0, 0])
        row += p_
    rec_t = L.h2d(rec, torch.int64, x.device)
    y = torch.empty((row, H), device=x.device, dtype=x.dtype)
    L.check(L.lib().sl_avgpool_batch(L.ptr(x), L.ptr(y), cu.data_ptr(), klen.data_ptr(), rec_t.data_ptr(), len(seqlens), max(P), H, kernel, stride,
                                     L.dtype_code(x.dtype), L.stream_ptr()), "sl_avgpool_batch")
    return y, P


def embed_gather(table: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    ids32 = ids.to(device=table.device, dtype=torch.int32).contiguous().view(-1)
    out = torch.empty((ids32.numel(), table.shape[1]), device=table.device, dtype=table.dtype)
    L.check(L.lib().sl_embed_gather(L.ptr(table), L.ptr(ids32), L.ptr(out), ids32.numel(), table.shape[1], L.dtype_code(table.dtype),
                                    L.stream_ptr()), "sl_embed_gather")
    return out


def attn_fwd(q, k, v, out, cu_q, cu_k, klen, *, q_strides, k_strides, v_strides, o_strides, nseq, max_qlen, n_heads, n_kv_heads,
             head_dim, causal, scale) -> torch.Tensor:
    a = L.AttnArgs()
    a.q, a.q_row_stride, a.q_head_stride = L.ptr(q), q_strides[0], q_strides[1]
    a.k, a.k_row_stride, a.k_head_stride = L.ptr(k), k_strides[0], k_strides[1]
    a.v, a.v_row_stride, a.v_head_stride = L.ptr(v), v_strides[0], v_strides[1]
    a.out, a.o_row_stride, a.o_head_stride = L.ptr(out), o_strides[0], o_strides[1]
    a.cu_q, a.cu_k, a.klen = L.ptr(cu_q), L.ptr(cu_k), L.ptr(klen)
    a.nseq, a.max_qlen, a.n_heads, a.n_kv_heads = nseq, max_qlen, n_heads, n_kv_heads
    a.head_dim, a.causal, a.dtype, a.scale = head_dim, int(causal), L.dtype_code(out.dtype), scale
    L.check(L.lib().sl_attn_fwd(C.byref(a), L.stream_ptr()), "sl_attn_fwd")
    return out


_SEQ_DESC = {}


def seq_descriptors(seqlens: Sequence[int], device):
    """(cu_seqlens int32 (n+1), lens int32 (n)) device tensors of a packed batch, cached per length tuple (the KD step calls
    attention forward + backward for every layer with the same lengths)."""
    key = (tuple(int(n) for n in seqlens), str(device))
    d = _SEQ_DESC.get(key)
    if d is None:
        if len(_SEQ_DESC) > 64:
            _SEQ_DESC.clear()
        cu = [0]
        for n in key[0]:
            cu.append(cu[-1] + n)
        d = (L.h2d(cu, torch.int32, device), L.h2d(list(key[0]), torch.int32, device))
        _SEQ_DESC[key] = d
    return d


def attn_packed_qkv(qkv: torch.Tensor, seqlens: Sequence[int], n_heads: int, n_kv_heads: int, head_dim: int, causal: bool,
                    scale: float, dropout_p: float = 0.0, dropout_seed: int = 0, lse: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention over a packed fused (tokens, (nh+2nkv)*D) activation (the HuBERT layout).  dropout_p > 0: training-mode
    dropout of the attention probabilities (mask index ((token * nh + head) << 16) | key, see sl_attn_args).
    lse: optional fp32 (tokens, n_heads) buffer that receives the log-sum-exp (what attn_packed_qkv_bwd needs)."""
    dev = qkv.device
    cu, klen = seq_descriptors(seqlens, dev)
    ntok = qkv.shape[0]
    rs = qkv.stride(0)
    out = torch.empty((ntok, n_heads * head_dim), device=dev, dtype=qkv.dtype)
    qv = qkv
    kv = qkv[:, n_heads * head_dim:]
    vv = qkv[:, (n_heads + n_kv_heads) * head_dim:]
    a = L.AttnArgs()
    a.q, a.q_row_stride, a.q_head_stride = qv.data_ptr(), rs, head_dim
    a.k, a.k_row_stride, a.k_head_stride = kv.data_ptr(), rs, head_dim
    a.v, a.v_row_stride, a.v_head_stride = vv.data_ptr(), rs, head_dim
    a.out, a.o_row_stride, a.o_head_stride = out.data_ptr(), n_heads * head_dim, head_dim
    a.cu_q, a.cu_k, a.klen = cu.data_ptr(), cu.data_ptr(), klen.data_ptr()
    a.nseq, a.max_qlen, a.n_heads, a.n_kv_heads = len(seqlens), max(seqlens), n_heads, n_kv_heads
    a.head_dim, a.causal, a.dtype, a.scale = head_dim, int(causal), L.dtype_code(qkv.dtype), scale
    a.dropout_p, a.dropout_seed = float(dropout_p), int(dropout_seed) & 0xFFFFFFFFFFFFFFFF
    a.lse = L.ptr(lse)
    L.check(L.lib().sl_attn_fwd(C.byref(a), L.stream_ptr()), "sl_attn_fwd")
    return out


def attn_packed_qkv_bwd(qkv: torch.Tensor, out: torch.Tensor, d_out: torch.Tensor, lse: torch.Tensor, d_qkv: torch.Tensor, seqlens: Sequence[int],
                        n_heads: int, n_kv_heads: int, head_dim: int, causal: bool, scale: float, dropout_p: float = 0.0,
                        dropout_seed: int = 0) -> torch.Tensor:
    """Flash-style backward of attn_packed_qkv (sl_attn_bwd): d_qkv (same fused layout as qkv) <- [dQ | dK | dV]."""
    dev = qkv.device
    cu, klen = seq_descriptors(seqlens, dev)
    ntok, rs, D = qkv.shape[0], qkv.stride(0), head_dim
    esz = qkv.element_size()
    koff, voff = n_heads * D, (n_heads + n_kv_heads) * D
    delta = torch.empty((ntok, n_heads), device=dev, dtype=torch.float32)
    a = L.AttnBwdArgs()
    a.q, a.q_row_stride, a.q_head_stride = qkv.data_ptr(), rs, D
    a.k, a.k_row_stride, a.k_head_stride = qkv.data_ptr() + koff * esz, rs, D
    a.v, a.v_row_stride, a.v_head_stride = qkv.data_ptr() + voff * esz, rs, D
    a.out, a.o_row_stride, a.o_head_stride = out.data_ptr(), out.stride(0), D
    a.d_out, a.do_row_stride, a.do_head_stride = d_out.data_ptr(), d_out.stride(0), D
    drs = d_qkv.stride(0)
    a.dq, a.dq_row_stride, a.dq_head_stride = d_qkv.data_ptr(), drs, D
    a.dk, a.dk_row_stride, a.dk_head_stride = d_qkv.data_ptr() + koff * esz, drs, D
    a.dv, a.dv_row_stride, a.dv_head_stride = d_qkv.data_ptr() + voff * esz, drs, D
    a.lse, a.delta = lse.data_ptr(), delta.data_ptr()
    a.cu_q, a.cu_k, a.klen = cu.data_ptr(), cu.data_ptr(), klen.data_ptr()
    a.n_tok_q, a.nseq = ntok, len(seqlens)
    a.max_qlen = a.max_klen = max(int(n) for n in seqlens)
    a.n_heads, a.n_kv_heads, a.head_dim, a.causal, a.dtype = n_heads, n_kv_heads, D, int(causal), L.dtype_code(qkv.dtype)
    a.scale, a.dropout_p, a.dropout_seed = scale, float(dropout_p), int(dropout_seed) & 0xFFFFFFFFFFFFFFFF
    L.check(L.lib().sl_attn_bwd(C.byref(a), L.stream_ptr()), "sl_attn_bwd")
    return d_qkv


def attn_dropout_bwd(P: torch.Tensor, Pd: torch.Tensor, dP: torch.Tensor, n_mat: int, smax: int, dims: torch.Tensor, ld: int, cu_q: torch.Tensor,
                     n_heads: int, n_kv_heads: int, r: int, p: float, seed: int) -> None:
    L.check(L.lib().sl_attn_dropout_bwd(L.ptr(P), L.ptr(Pd), L.ptr(dP), n_mat, smax, L.ptr(dims), ld, L.ptr(cu_q), n_heads, n_kv_heads, r, float(p),
                                        int(seed) & 0xFFFFFFFFFFFFFFFF, L.dtype_code(P.dtype), L.stream_ptr()), "sl_attn_dropout_bwd")


def rope_kv_append(qkv, k_cache, v_cache, tok_seq, tok_pos, cos, sin, n_heads, n_kv, D, max_ctx) -> None:
    L.check(L.lib().sl_rope_kv_append(L.ptr(qkv), L.ptr(k_cache), L.ptr(v_cache), L.ptr(tok_seq), L.ptr(tok_pos), L.ptr(cos), L.ptr(sin),
                                      qkv.shape[0], n_heads, n_kv, D, max_ctx, L.dtype_code(qkv.dtype), L.stream_ptr()),
            "sl_rope_kv_append")


def attn_decode(q, q_stride, k_cache, v_cache, ctx_len, n_heads, n_kv, D, max_ctx, scale) -> torch.Tensor:
    B = ctx_len.shape[0]
    out = torch.empty((B, n_heads * D), device=q.device, dtype=q.dtype)
    L.check(L.lib().sl_attn_decode(L.ptr(q), q_stride, L.ptr(k_cache), L.ptr(v_cache), L.ptr(out), L.ptr(ctx_len), B, n_heads, n_kv, D,
                                   max_ctx, scale, L.dtype_code(q.dtype), L.stream_ptr()), "sl_attn_decode")
    return out


def greedy_select(logits, eos_ids, pad_id, use_eos, unfinished, ctx_len, gen_count, finish_len, next_ids, out_ids) -> None:
    B, V = logits.shape
    eos = (C.c_int32 * max(1, len(eos_ids)))(*eos_ids)
    L.check(L.lib().sl_greedy_select(L.ptr(logits), B, V, eos, len(eos_ids), pad_id, int(use_eos), L.ptr(unfinished), L.ptr(ctx_len),
                                     L.ptr(gen_count), L.ptr(finish_len), L.ptr(next_ids), L.ptr(out_ids), out_ids.shape[1],
                                     L.stream_ptr()), "sl_greedy_select")


def gemm_top1(A: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor] = None):
    """Row-wise top-1 of A W^T (+ bias) without storing the product (sl_gemm_ex_args.amax_*): returns (val, idx), each
    (ceil(N / 64), M): the largest value of every 64-column group and its column."""
    L.require_gpu(A, "A"); L.require_gpu(W, "W")
    M, K = A.shape
    N = W.shape[0]
    ng = (N + 63) // 64
    val = torch.empty((ng, M), device=A.device, dtype=torch.float32)
    idx = torch.empty((ng, M), device=A.device, dtype=torch.int32)
    a = L.GemmArgs()
    a.A, a.lda, a.W, a.ldw, a.C, a.ldc = L.ptr(A), A.stride(0), L.ptr(W), W.stride(0), None, N
    a.bias = L.ptr(bias)
    a.M, a.N, a.K, a.batch = M, N, K, 1
    a.dtype, a.act, a.out_f32 = L.dtype_code(A.dtype), L.ACT_NONE, 1
    ex = L.GemmEx()
    ex.w_mod, ex.amax_val, ex.amax_idx = 1, L.ptr(val), L.ptr(idx)
    L.check(L.lib().sl_gemm_ex(C.byref(a), C.byref(ex), L.stream_ptr()), "sl_gemm_ex")
    return val, idx


def greedy_select_partial(val, idx, eos_ids, pad_id, use_eos, unfinished, ctx_len, gen_count, finish_len, next_ids, out_ids, advance_ctx=True) -> None:
    ng, B = val.shape
    eos = (C.c_int32 * max(1, len(eos_ids)))(*eos_ids)
    L.check(L.lib().sl_greedy_select_partial(L.ptr(val), L.ptr(idx), ng, B, eos, len(eos_ids), pad_id, int(use_eos), int(advance_ctx), L.ptr(unfinished),
                                             L.ptr(ctx_len), L.ptr(gen_count), L.ptr(finish_len), L.ptr(next_ids), L.ptr(out_ids), out_ids.shape[1],
                                             L.stream_ptr()), "sl_greedy_select_partial")


def pack_weight(W: torch.Tensor) -> torch.Tensor:
    """(N,K) row-major -> fragment-packed buffer for the decode kernel (sl_pack_weight)."""
    L.require_gpu(W, "W")
    n, k = W.shape
    out = torch.empty(((n + 15) // 16 * 16, k), device=W.device, dtype=W.dtype)
    L.check(L.lib().sl_pack_weight(L.ptr(W), W.stride(0), L.ptr(out), n, k, L.dtype_code(W.dtype), L.stream_ptr()), "sl_pack_weight")
    return out


def gemm_decode(A: torch.Tensor, Wp: torch.Tensor, N: int, *, residual=None, act: int = L.ACT_NONE, out_f32: bool = False,
                fuse_rms: bool = False, eps: float = 1e-5, rope=None, out: Optional[torch.Tensor] = None,
                split_k: bool = True, rstd_in: Optional[torch.Tensor] = None, rstd_out: Optional[torch.Tensor] = None,
                norm_out: Optional[torch.Tensor] = None, norm_gain: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Decode GEMM on packed weights.  rope = dict(cos, sin, pos, seq, k_cache, v_cache, n_heads, n_kv, max_ctx) for ACT_ROPE_KV.
    split_k: hand the kernel a scratch buffer so that row counts > 64 may split K over blocks (gemm_stream.hip)."""
    M, K = A.shape
    if act == L.ACT_SILU_MUL:
        n_out = N // 2
    elif act == L.ACT_ROPE_KV:
        n_out = rope["n_heads"] * 128
    else:
        n_out = N
    if out is None:
        out = torch.empty((M, n_out), device=A.device, dtype=torch.float32 if out_f32 else A.dtype)
    a = L.GemmArgs()
    a.A, a.lda = L.ptr(A), A.stride(0)
    a.W, a.ldw = L.ptr(Wp), K
    a.C, a.ldc = L.ptr(out), out.stride(0)
    a.residual, a.ldr = L.ptr(residual), (residual.stride(0) if residual is not None else 0)
    a.M, a.N, a.K, a.batch = M, N, K, 1
    a.dtype, a.act, a.out_f32, a.w_layout = L.dtype_code(A.dtype), act, int(out_f32), L.W_PACKED
    f = L.GemmFused()
    f.fuse_rms, f.rms_eps = int(fuse_rms), eps
    f.rstd_in, f.rstd_out = L.ptr(rstd_in), L.ptr(rstd_out)
    f.norm_out, f.norm_gain = L.ptr(norm_out), L.ptr(norm_gain)     # the reduce pass also writes the RMS-normalised rows (sl_gemm_fused.norm_out)
    if rope is not None:
        f.rope_cos, f.rope_sin, f.tok_pos, f.tok_seq = L.ptr(rope["cos"]), L.ptr(rope["sin"]), L.ptr(rope["pos"]), L.ptr(rope["seq"])
        f.k_cache, f.v_cache = L.ptr(rope["k_cache"]), L.ptr(rope["v_cache"])
        f.n_heads, f.n_kv_heads, f.max_ctx = rope["n_heads"], rope["n_kv"], rope["max_ctx"]
    if split_k:
        need = int(L.lib().sl_gemm_split_workspace_bytes(M, N, K, a.dtype))
        if need > 0:
            ws = _split_ws(A.device, need)
            f.split_ws, f.split_ws_bytes = L.ptr(ws), ws.numel()
    L.check(L.lib().sl_gemm_fused_decode(C.byref(a), C.byref(f), L.stream_ptr()), "sl_gemm_fused_decode")
    return out


_SPLIT_WS = {}


def _split_ws(device, nbytes: int) -> torch.Tensor:
    ws = _SPLIT_WS.get(device)
    if ws is None or ws.numel() < nbytes:
        ws = torch.zeros(nbytes, dtype=torch.uint8, device=device)   # the first 8 KiB hold the K-split fix-up's counters: zero on first use
        _SPLIT_WS[device] = ws
    return ws


def attn_decode_split(q, q_stride, k_cache, v_cache, ctx_len, n_heads, n_kv, D, max_ctx, scale, out=None, ws=None) -> torch.Tensor:
    B = ctx_len.shape[0]
    if out is None:
        out = torch.empty((B, n_heads * D), device=q.device, dtype=q.dtype)
    if ws is None:
        ws = torch.empty(int(L.lib().sl_attn_decode_workspace_bytes(B, n_heads, n_kv, max_ctx)), dtype=torch.uint8, device=q.device)
    L.check(L.lib().sl_attn_decode_split(L.ptr(q), q_stride, L.ptr(k_cache), L.ptr(v_cache), L.ptr(out), L.ptr(ws), L.ptr(ctx_len), B,
                                         n_heads, n_kv, D, max_ctx, scale, L.dtype_code(q.dtype), L.stream_ptr()), "sl_attn_decode_split")
    return out


# ---------------------------------------------------------------------------------------------
# training-side wrappers (sl_gemm_ex and train_ops.hip)
# ---------------------------------------------------------------------------------------------
def gemm_ex(A, W, *, M: int, N: int, K: int, lda: int, ldw: int, out: torch.Tensor, ldc: Optional[int] = None, bias=None,
            residual=None, ldr: int = 0, act: int = L.ACT_NONE, out_f32: bool = False, trans_a: bool = False, trans_w: bool = False,
            residual_f32: bool = False, aux_out=None, batch: int = 1, strideA: int = 0, strideW: int = 0, strideC: int = 0,
            strideBias: int = 0, strideR: int = 0, a_off: int = 0, w_off: int = 0, c_off: int = 0, r_off: int = 0,
            dtype: Optional[torch.dtype] = None, groups: Optional[torch.Tensor] = None, w_mod: int = 1, groups_ext: bool = False,
            ln_mr=None, ln_u=None, ln_c=None, stats_out=None, sk_ws: Optional[torch.Tensor] = None,
            post_op: int = 0, drop_p: float = 0.0, drop_seed: int = 0, drop_ld: int = 0, post_in=None, post_ld: int = 0, colsum_out=None,
            deferred_splits=None):
    """Raw-pointer GEMM with every backward feature; *_off are element offsets into the tensors.  ln_mr/ln_u/ln_c: the
    LayerNorm-folded consumer epilogue; stats_out: per-64-column {sum, sum of squares} of the stored rows (speechllm.h).
    post_op / drop_* / post_in / colsum_out: the training tape's epilogue fusions (sl_gemm_ex_args, ABI 7).
    deferred_splits: a ctypes c_int32 the library sets to the number of K runs it left un-reduced in sk_ws (0: `out` was written)."""
    dt = dtype or A.dtype
    esz = 4 if dt == torch.float32 else 2
    a = L.GemmArgs()
    a.A, a.lda, a.strideA = A.data_ptr() + a_off * esz, lda, strideA
    a.W, a.ldw, a.strideW = W.data_ptr() + w_off * esz, ldw, strideW
    a.C, a.ldc, a.strideC = out.data_ptr() + c_off * (4 if out_f32 else esz), (ldc if ldc is not None else out.stride(0)), strideC
    a.bias, a.strideBias = L.ptr(bias), strideBias
    a.residual = 0 if residual is None else residual.data_ptr() + r_off * (4 if residual_f32 else esz)
    a.ldr, a.strideR = ldr, strideR
    a.M, a.N, a.K, a.batch = M, N, K, batch
    a.dtype, a.act, a.out_f32, a.w_layout = L.dtype_code(dt), act, int(out_f32), L.W_ROWMAJOR
    e = L.GemmEx()
    e.trans_a, e.trans_w, e.residual_f32, e.aux_out = int(trans_a), int(trans_w), int(residual_f32), L.ptr(aux_out)
    e.groups, e.w_mod, e.groups_ext = L.ptr(groups), w_mod, int(groups_ext)
    e.ln_mr, e.ln_u, e.ln_c, e.stats_out = L.ptr(ln_mr), L.ptr(ln_u), L.ptr(ln_c), L.ptr(stats_out)
    if sk_ws is not None:       # stream-K workspace (streamk_workspace): zeroed once, one per stream
        e.sk_ws, e.sk_ws_bytes = sk_ws.data_ptr(), sk_ws.numel() * sk_ws.element_size()
    e.post_op, e.drop_p, e.drop_seed, e.drop_ld = int(post_op), float(drop_p), int(drop_seed) & 0xFFFFFFFFFFFFFFFF, int(drop_ld)
    e.post_in, e.post_ld, e.colsum_out = L.ptr(post_in), int(post_ld), L.ptr(colsum_out)
    if deferred_splits is not None:
        e.deferred_splits = C.cast(C.byref(deferred_splits), C.c_void_p)
    L.check(L.lib().sl_gemm_ex(C.byref(a), C.byref(e), L.stream_ptr()), "sl_gemm_ex")
    return out


def streamk_workspace(device) -> torch.Tensor:
    """A zeroed stream-K workspace (sl_gemm_ex_args.sk_ws): one per stream that launches GEMMs."""
    return torch.zeros(int(L.lib().sl_gemm_streamk_workspace_bytes()), device=device, dtype=torch.uint8)


def layernorm_stats(x: torch.Tensor, eps: float) -> torch.Tensor:
    """(rows, 2) fp32 {mean, rstd} of every row of x (sl_layernorm_stats) — the first layer's input of the folded encoder."""
    rows, cols = x.shape
    mr = torch.empty((rows, 2), device=x.device, dtype=torch.float32)
    L.check(L.lib().sl_layernorm_stats(L.ptr(x), rows, cols, eps, L.ptr(mr), L.dtype_code(x.dtype), L.stream_ptr()), "sl_layernorm_stats")
    return mr


def layernorm_stats_finalize(stats: torch.Tensor, cols: int, eps: float) -> torch.Tensor:
    """(rows, 2) fp32 {mean, rstd} from a producer GEMM's (rows, cols/64, 2) segment sums (sl_layernorm_stats_finalize)."""
    rows, segs = stats.shape[0], stats.shape[1]
    mr = torch.empty((rows, 2), device=stats.device, dtype=torch.float32)
    L.check(L.lib().sl_layernorm_stats_finalize(L.ptr(stats), segs, rows, cols, eps, L.ptr(mr), L.stream_ptr()), "sl_layernorm_stats_finalize")
    return mr


def transpose_pad(x: torch.Tensor, rows: int, cols: int, ld_out: Optional[int] = None) -> torch.Tensor:
    """(cols, ld_out) = x[:rows, :cols]^T, zero-filled beyond `rows` (sl_transpose_pad)."""
    ld_out = rows if ld_out is None else ld_out
    vec = 4 if x.dtype == torch.float32 else 8
    ldy = (ld_out + vec - 1) // vec * vec
    y = torch.empty((cols, ldy), device=x.device, dtype=x.dtype)
    L.check(L.lib().sl_transpose_pad(L.ptr(x), x.stride(0), L.ptr(y), ldy, rows, cols, ld_out, L.dtype_code(x.dtype), L.stream_ptr()), "sl_transpose_pad")
    return y if ldy == ld_out else y[:, :ld_out]


def dgrad(dY: torch.Tensor, W: torch.Tensor, out: Optional[torch.Tensor] = None, wt: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dX (M, K_in) = dY (M, N_out) . W (N_out, K_in).  wt: W stored transposed (K_in, N_out), contiguous — then the
    product is a plain K-contiguous GEMM and runs the LDS-DMA tiled kernels instead of the transposed-operand loader."""
    if wt is not None:
        return gemm(dY, wt, out=out)
    M, Nout = dY.shape
    Kin = W.shape[1]
    if out is None:
        out = torch.empty((M, Kin), device=dY.device, dtype=dY.dtype)
    return gemm_ex(dY, W, M=M, N=Kin, K=Nout, lda=dY.stride(0), ldw=W.stride(0), out=out, trans_w=True)


def wgrad_acc(dY: torch.Tensor, X: torch.Tensor, dW: torch.Tensor, *, ldx: Optional[int] = None, Kin: Optional[int] = None,
              M: Optional[int] = None, db: Optional[torch.Tensor] = None) -> torch.Tensor:
    """dW (N_out, K_in) fp32 += dY^T (N_out, M) . X (M, K_in); X rows may overlap (ldx < K_in: implicit-GEMM conv windows).
    db (fp32, N_out): the bias gradient db += colsum(dY) — inside the token-major product where that kernel runs, else a sl_colsum launch."""
    M = dY.shape[0] if M is None else M
    Nout = dY.shape[1]
    Kin = X.shape[1] if Kin is None else Kin
    tt = (ldx is None and M >= 256 and dY.dtype == torch.bfloat16 and Nout % 128 == 0 and Kin % 128 == 0 and dY.stride(0) % 8 == 0 and X.stride(0) % 8 == 0
          and dY.data_ptr() % 16 == 0 and X.data_ptr() % 16 == 0 and os.environ.get("SL_WGRAD_TR", "1") != "0")
    if db is not None and not (tt and os.environ.get("SL_TAPE_FUSE", "1") != "0"):
        colsum_acc(dY[:M], db)
        db = None
    if tt:
        return gemm_ex(dY, X, M=Nout, N=Kin, K=M, lda=dY.stride(0), ldw=X.stride(0), out=dW, ldc=dW.stride(0), residual=dW, ldr=dW.stride(0),
                       out_f32=True, residual_f32=True, trans_a=True, trans_w=True, dtype=dY.dtype, colsum_out=db)
    if ldx is None and M >= 256:
        # plain Linear: contract over token rows on K-contiguous copies (dY^T, X^T zero-padded to a whole number of K slabs) so
        # the product runs the LDS-DMA tiled kernels; the doubly-transposed register loader measured ~100 TF/s on these shapes
        Mp = (M + 63) // 64 * 64
        dYT = transpose_pad(dY, M, Nout, Mp)
        XT = transpose_pad(X, M, Kin, Mp)
        return gemm_ex(dYT, XT, M=Nout, N=Kin, K=Mp, lda=Mp, ldw=Mp, out=dW, ldc=dW.stride(0), residual=dW, ldr=dW.stride(0), out_f32=True,
                       residual_f32=True, dtype=dY.dtype)
    return gemm_ex(dY, X, M=Nout, N=Kin, K=M, lda=dY.stride(0), ldw=(X.stride(0) if ldx is None else ldx), out=dW, ldc=dW.stride(0),
                   residual=dW, ldr=dW.stride(0), out_f32=True, residual_f32=True, trans_a=True, trans_w=True, dtype=dY.dtype)


def gelu_bwd(dy, pre):
    dx = torch.empty_like(dy)
    L.check(L.lib().sl_gelu_bwd(L.ptr(dy), L.ptr(pre), L.ptr(dx), dy.numel(), L.dtype_code(dy.dtype), L.stream_ptr()), "sl_gelu_bwd")
    return dx


def axpby(x, y, a=1.0, b=1.0):
    L.check(L.lib().sl_axpby(L.ptr(x), L.ptr(y), a, b, y.numel(), L.dtype_code(y.dtype), L.stream_ptr()), "sl_axpby")
    return y


def dropout(x: torch.Tensor, p: float, seed: int, residual: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """out = (residual or 0) + keep * x / (1 - p) with the counter-based mask of sl_dropout (same seed -> same mask: the
    backward pass calls this on the gradient).  `out` may alias x."""
    L.require_gpu(x, "x")
    out = torch.empty_like(x) if out is None else out
    L.check(L.lib().sl_dropout(L.ptr(x), L.ptr(residual), L.ptr(out), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, L.dtype_code(x.dtype),
                               L.stream_ptr()), "sl_dropout")
    return out


def dropout_keep_mask(n: int, p: float, seed: int) -> "torch.Tensor":
    """Host restatement of sl_dropout's mask (bool, n elements) — used by the tests' oracle to apply identical masks."""
    import numpy as np
    m32 = np.uint64(0xFFFFFFFF)

    def lowbias32(v):
        v = v & m32
        v ^= v >> np.uint64(16); v = (v * np.uint64(0x7feb352d)) & m32
        v ^= v >> np.uint64(15); v = (v * np.uint64(0x846ca68b)) & m32
        v ^= v >> np.uint64(16)
        return v

    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    i = np.arange(n, dtype=np.uint64)
    h = lowbias32((i & m32) ^ lowbias32((i >> np.uint64(32)) ^ np.uint64(seed & 0xFFFFFFFF)) ^ np.uint64(seed >> 32))
    thr = np.uint64(int(float(np.float32(p)) * 16777216.0))
    return torch.from_numpy((h >> np.uint64(8)) >= thr)


def silu_mul(gu):
    M, F2 = gu.shape
    out = torch.empty((M, F2 // 2), device=gu.device, dtype=gu.dtype)
    L.check(L.lib().sl_silu_mul(L.ptr(gu), L.ptr(out), M, F2 // 2, L.dtype_code(gu.dtype), L.stream_ptr()), "sl_silu_mul")
    return out


def silu_mul_bwd(gu, dy):
    dgu = torch.empty_like(gu)
    L.check(L.lib().sl_silu_mul_bwd(L.ptr(gu), L.ptr(dy), L.ptr(dgu), gu.shape[0], gu.shape[1] // 2, L.dtype_code(gu.dtype), L.stream_ptr()),
            "sl_silu_mul_bwd")
    return dgu


def rope_inplace(x, tok_pos, cos, sin, heads, n_rot, D, inverse=False):
    L.check(L.lib().sl_rope_inplace(L.ptr(x), L.ptr(tok_pos), L.ptr(cos), L.ptr(sin), x.shape[0], heads, n_rot, D, int(inverse),
                                    L.dtype_code(x.dtype), L.stream_ptr()), "sl_rope_inplace")
    return x


_LN_WS = {}


def layernorm_bwd(x, g, b, dy, eps, dgamma=None, dbeta=None, gelu=False):
    dx = torch.empty_like(dy)
    rows = x.numel() // x.shape[-1]
    need = int(L.lib().sl_layernorm_bwd_ws_bytes(rows, x.shape[-1])) if dgamma is not None and dbeta is not None else 0
    ws = None
    if need:                                  # scratch for the blocks' dgamma / dbeta records (one buffer per device, grown on demand)
        ws = _LN_WS.get(x.device)
        if ws is None or ws.numel() < need:
            ws = _LN_WS[x.device] = torch.empty(need, dtype=torch.uint8, device=x.device)
    L.check(L.lib().sl_layernorm_bwd_ws(L.ptr(x), L.ptr(g), L.ptr(b), L.ptr(dy), L.ptr(dx), L.ptr(dgamma), L.ptr(dbeta), rows, x.shape[-1], eps,
                                        int(gelu), L.dtype_code(x.dtype), L.ptr(ws), ws.numel() if ws is not None else 0, L.stream_ptr()),
            "sl_layernorm_bwd_ws")
    return dx


def rmsnorm_bwd(x, w, dy, eps):
    dx = torch.empty_like(dy)
    rows = x.numel() // x.shape[-1]
    L.check(L.lib().sl_rmsnorm_bwd(L.ptr(x), L.ptr(w), L.ptr(dy), L.ptr(dx), rows, x.shape[-1], eps, L.dtype_code(x.dtype), L.stream_ptr()),
            "sl_rmsnorm_bwd")
    return dx


def colsum_acc(x, out_f32):
    L.check(L.lib().sl_colsum(L.ptr(x), x.stride(0), L.ptr(out_f32), x.shape[0], x.shape[1], L.dtype_code(x.dtype), L.stream_ptr()), "sl_colsum")
    return out_f32


def softmax_rows(S, n_mats, rows, cols, ld, scale, causal, dtype):
    P = torch.empty((n_mats, rows, ld), device=S.device, dtype=dtype)
    L.check(L.lib().sl_softmax_rows(L.ptr(S), L.ptr(P), n_mats, rows, cols, ld, scale, int(causal), L.dtype_code(dtype), L.stream_ptr()),
            "sl_softmax_rows")
    return P


def softmax_bwd(P, dP, cols, scale):
    dS = torch.empty_like(P)
    nrows = P.shape[0] * P.shape[1]
    L.check(L.lib().sl_softmax_bwd(L.ptr(P), L.ptr(dP), L.ptr(dS), nrows, cols, P.shape[2], scale, L.dtype_code(P.dtype), L.stream_ptr()),
            "sl_softmax_bwd")
    return dS


def softmax_rows_var(S, P, n_mats, rows_per_mat, mat_dim, ld, scale, causal, dtype):
    L.check(L.lib().sl_softmax_rows_var(L.ptr(S), L.ptr(P), n_mats, rows_per_mat, L.ptr(mat_dim), ld, scale, int(causal), L.dtype_code(dtype),
                                        L.stream_ptr()), "sl_softmax_rows_var")
    return P


def softmax_bwd_var(P, dP, dS, n_mats, rows_per_mat, mat_dim, ld, scale):
    L.check(L.lib().sl_softmax_bwd_var(L.ptr(P), L.ptr(dP), L.ptr(dS), n_mats, rows_per_mat, L.ptr(mat_dim), ld, scale, L.dtype_code(P.dtype),
                                       L.stream_ptr()), "sl_softmax_bwd_var")
    return dS


def ce_loss(logits, labels, coef, loss, dlogits=None, accumulate=False, dtype=torch.float32):
    L.check(L.lib().sl_ce_loss(L.ptr(logits), L.ptr(labels), logits.shape[0], logits.shape[1], coef, L.ptr(loss), L.ptr(dlogits),
                               int(accumulate), L.dtype_code(dtype), L.stream_ptr()), "sl_ce_loss")


def soft_ce_loss(student, teacher, coef, loss, dstudent=None, accumulate=False, dtype=torch.float32):
    L.check(L.lib().sl_soft_ce_loss(L.ptr(student), L.ptr(teacher), student.shape[0], student.shape[1], coef, L.ptr(loss), L.ptr(dstudent),
                                    int(accumulate), L.dtype_code(dtype), L.stream_ptr()), "sl_soft_ce_loss")


def mse_loss(a, b, coef, loss, da=None, accumulate=False):
    L.check(L.lib().sl_mse_loss(L.ptr(a), L.ptr(b), a.numel(), coef, L.ptr(loss), L.ptr(da), int(accumulate), L.dtype_code(a.dtype),
                                L.stream_ptr()), "sl_mse_loss")


def avgpool_bwd(dy, T, kernel, stride, out=None):
    """out (optional): the (T, H) rows of a packed buffer the gradient is written into (no temporary + copy per utterance)."""
    P, H = dy.shape
    dx = torch.empty((T, H), device=dy.device, dtype=dy.dtype) if out is None else out
    assert dx.is_contiguous() and tuple(dx.shape) == (T, H)
    L.check(L.lib().sl_avgpool_bwd(L.ptr(dy), L.ptr(dx), T, H, kernel, stride, P, L.dtype_code(dy.dtype), L.stream_ptr()), "sl_avgpool_bwd")
    return dx


def col2im(dcol, Lin, Cc, k, s, out=None):
    """out (optional): the (Lin, Cc) rows of a packed buffer the gradient is written into (no temporary + copy per utterance)."""
    dx = torch.empty((Lin, Cc), device=dcol.device, dtype=dcol.dtype) if out is None else out
    assert dx.is_contiguous() and tuple(dx.shape) == (Lin, Cc)
    L.check(L.lib().sl_col2im(L.ptr(dcol), L.ptr(dx), Lin, dcol.shape[0], Cc, k, s, L.dtype_code(dcol.dtype), L.stream_ptr()), "sl_col2im")
    return dx


def col2im_batch(dcol, out, src_rows: Sequence[int], dst_rows: Sequence[int], src_off: Sequence[int], dst_off: Sequence[int], Cc: int, k: int, s: int):
    """sl_col2im for every utterance of a packed batch in one launch: utterance u's windows are rows src_off[u] .. + src_rows[u] of dcol (k * Cc wide), its
    gradient rows dst_off[u] .. + dst_rows[u] of `out` (Cc wide)."""
    desc = L.h2d([[src_rows[u], dst_rows[u], src_off[u], dst_off[u]] for u in range(len(src_rows))], torch.int64, dcol.device)
    L.check(L.lib().sl_col2im_batch(L.ptr(dcol), L.ptr(out), desc.data_ptr(), len(src_rows), max(int(n) for n in dst_rows), Cc, k, s, L.dtype_code(dcol.dtype),
                                    L.stream_ptr()), "sl_col2im_batch")
    return out


def avgpool_bwd_batch(dy, out, P: Sequence[int], T: Sequence[int], src_off: Sequence[int], dst_off: Sequence[int], kernel: int, stride: int):
    """sl_avgpool_bwd for every utterance of a packed batch in one launch (rows src_off[u] .. + P[u] of dy -> rows dst_off[u] .. + T[u] of out)."""
    desc = L.h2d([[P[u], T[u], src_off[u], dst_off[u]] for u in range(len(P))], torch.int64, dy.device)
    L.check(L.lib().sl_avgpool_bwd_batch(L.ptr(dy), L.ptr(out), desc.data_ptr(), len(P), max(int(n) for n in T), dy.shape[1], kernel, stride,
                                         L.dtype_code(dy.dtype), L.stream_ptr()), "sl_avgpool_bwd_batch")
    return out


def hubert_conv0_bwd_batch(waves: Sequence[torch.Tensor], w, bias, gamma, beta, dy, row_offsets: Sequence[int], dw, dbias, dgamma, dbeta, k=10, stride=5,
                           eps=1e-5):
    """conv0 + LayerNorm + GELU backward of every utterance of a packed batch in ONE launch (dy rows row_offsets[u]..)."""
    dev = dy.device
    flat = torch.cat([wv.reshape(-1) for wv in waves]).contiguous()
    soff, spref = [0], [0]
    for u, wv in enumerate(waves):
        soff.append(soff[-1] + wv.numel())
        spref.append(spref[-1] + (row_offsets[u + 1] - row_offsets[u] + 23) // 24)        # strips of 24 time steps (k=10, s=5)
    desc = L.h2d([soff, list(row_offsets), spref], torch.int64, dev)
    L.check(L.lib().sl_hubert_conv0_bwd_batch(L.ptr(flat), desc[0].data_ptr(), desc[1].data_ptr(), desc[2].data_ptr(), len(waves), spref[-1], L.ptr(w),
                                              L.ptr(bias), L.ptr(gamma), L.ptr(beta), L.ptr(dy), w.shape[0], k, stride, eps, L.ptr(dw), L.ptr(dbias),
                                              L.ptr(dgamma), L.ptr(dbeta), L.dtype_code(dy.dtype), L.stream_ptr()), "sl_hubert_conv0_bwd_batch")


def hubert_conv0_bwd(wave, w, bias, gamma, beta, dy, dw, dbias, dgamma, dbeta, k=10, stride=5, eps=1e-5):
    L.check(L.lib().sl_hubert_conv0_bwd(L.ptr(wave), wave.numel(), L.ptr(w), L.ptr(bias), L.ptr(gamma), L.ptr(beta), L.ptr(dy), w.shape[0], k,
                                        stride, eps, L.ptr(dw), L.ptr(dbias), L.ptr(dgamma), L.ptr(dbeta), L.dtype_code(dy.dtype),
                                        L.stream_ptr()), "sl_hubert_conv0_bwd")


def kd_logit_losses(student, teacher, labels, row_coef, row_slot, losses, dstudent, dtype):
    """All next-token / soft cross-entropy terms of a window in one launch (sl_kd_logit_losses)."""
    L.check(L.lib().sl_kd_logit_losses(L.ptr(student), L.ptr(teacher), L.ptr(labels), L.ptr(row_coef), L.ptr(row_slot), student.shape[0], student.shape[1],
                                       L.ptr(losses), losses.stride(0), L.ptr(dstudent), L.dtype_code(dtype), L.stream_ptr()), "sl_kd_logit_losses")


def kd_mse_rows(a, b, row_coef, row_slot, losses, loss_col, da=None):
    """Feature-distillation MSE of one tap over every utterance's tail rows in one launch (sl_kd_mse_rows)."""
    L.check(L.lib().sl_kd_mse_rows(L.ptr(a), L.ptr(b), L.ptr(row_coef), L.ptr(row_slot), a.shape[0], a.shape[1], L.ptr(losses), losses.stride(0), loss_col,
                                   L.ptr(da), L.dtype_code(a.dtype), L.stream_ptr()), "sl_kd_mse_rows")


def sample_select(logits, temperature, top_k, top_p, seed, eos_ids, pad_id, use_eos, unfinished, ctx_len, gen_count, finish_len, next_ids, out_ids) -> None:
    """sl_sample_select: HF's temperature / top-k / top-p warpers + one reproducible draw per row, then greedy_select's bookkeeping."""
    B, V = logits.shape
    eos = (C.c_int32 * max(1, len(eos_ids)))(*eos_ids)
    choice = torch.empty(B, dtype=torch.int32, device=logits.device)
    L.check(L.lib().sl_sample_select(L.ptr(logits), B, V, float(temperature), int(top_k), float(top_p), int(seed) & 0xFFFFFFFFFFFFFFFF, eos, len(eos_ids), pad_id,
                                     int(use_eos), L.ptr(unfinished), L.ptr(ctx_len), L.ptr(gen_count), L.ptr(finish_len), L.ptr(next_ids), L.ptr(out_ids),
                                     out_ids.shape[1], L.ptr(choice), L.stream_ptr()), "sl_sample_select")


def sample_uniform(seed: int, row: int, step: int) -> float:
    """Host restatement of the uniform sl_sample_select draws for (seed, row, step) — the tests rebuild the inverse-CDF pick with it."""
    m32 = 0xFFFFFFFF

    def lowbias32(v):
        v &= m32
        v ^= v >> 16; v = (v * 0x7feb352d) & m32
        v ^= v >> 15; v = (v * 0x846ca68b) & m32
        v ^= v >> 16
        return v

    seed &= 0xFFFFFFFFFFFFFFFF
    h = lowbias32((step & m32) ^ lowbias32((row & m32) ^ (seed & m32)) ^ (seed >> 32))
    return (h >> 8) / 16777216.0
