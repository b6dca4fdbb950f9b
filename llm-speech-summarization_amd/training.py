"""Knowledge-distillation step on the HIP path: host-side mirror of ref:trainer.py:270-384.

One micro-step = one utterance (the reference's batch-size-1 semantics, ref:README.md:86):
    encoder(audio) -> [prefix | audio | suffix[1:] | response[1:]] -> frozen LLM (student pass, activations kept)
    [prefix | text | suffix[1:] | response[1:]]                     -> frozen LLM (teacher pass, no grad)
    total = w_ntp * CE(response-only) + w_ld * soft-CE(student, teacher) + w_fd * sum_taps MSE(hidden)
    backward: dgrad-only through the LLM (its weights are frozen, ref:trainer.py:63-64), dgrad + wgrad
    through the audio encoder; fp32 gradient accumulation over `grad_accum_interval` micro-steps.
Every FLOP is a libspeechllm kernel (forward kernels of the inference path + sl_gemm_ex + train_ops.hip);
this module is the tape: it decides what to keep from the forward and in which order to call the backward
kernels.  torch supplies buffers, the optimizer (AdamW on fp32 master weights, ref:trainer.py:98-105) and the
collective (RCCL all-reduce of the fp32 gradient buckets, overlapped with the remaining backward).

Training-mode regularisers (the reference trains with `audio_encoder.train()`, ref:trainer.py:258, so HF's HuBERT applies
feature-projection / hidden / activation dropout, LayerDrop and SpecAugment time masking): `TrainRegularizers` switches
them on (`KDTrainer(..., regularizers=...)`).  Masks come from a counter-based hash (sl_dropout), so forward and backward
agree without stored masks and the test oracle can rebuild them; SpecAugment spans follow HF's `_compute_mask_indices`
on numpy's global RNG.  Attention-probability dropout is applied inside the attention kernels (the mask index is a
function of (token, head, key), so the flash-style backward, which recomputes the probabilities tile by tile, rebuilds it).  Deterministic KD-step parity
(golden fixtures) is defined with the regularisers off.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L
from . import ops
from .audio_encoder import AudioEncoder
from .audio_llama import AudioLlamaForCausalLM
from .weights import HubertDeviceWeights, fold_pos_conv_weight


# ------------------------------------------------------------------------------------------------
# training-mode regularisers of the HF HuBERT encoder (hf:models/hubert/modeling_hubert.py: feature projection dropout,
# _mask_hidden_states, encoder dropout + LayerDrop, layer dropouts)
# ------------------------------------------------------------------------------------------------
@dataclass
class TrainRegularizers:
    """hubert-large-ls960-ft config values (SURVEY.md §8 model constants)."""
    feat_proj_dropout: float = 0.1
    hidden_dropout: float = 0.1
    activation_dropout: float = 0.1
    attention_dropout: float = 0.1
    layerdrop: float = 0.1
    apply_spec_augment: bool = True
    mask_time_prob: float = 0.05
    mask_time_length: int = 10
    mask_time_min_masks: int = 2
    seed: int = 1234                      # base of the dropout / LayerDrop counters (ref:config/llama3_hubert.yaml:1 seed_everything)

    @staticmethod
    def from_hf_config(d: dict, seed: int = 1234) -> "TrainRegularizers":
        return TrainRegularizers(d.get("feat_proj_dropout", 0.0), d.get("hidden_dropout", 0.1), d.get("activation_dropout", 0.1),
                                 d.get("attention_dropout", 0.1), d.get("layerdrop", 0.1), d.get("apply_spec_augment", True), d.get("mask_time_prob", 0.05),
                                 d.get("mask_time_length", 10), d.get("mask_time_min_masks", 2), seed)

    @staticmethod
    def from_whisper_config(d: dict, seed: int = 1234) -> "TrainRegularizers":
        """hf:models/whisper/configuration_whisper.py names: `dropout` (after attention / FFN), `activation_dropout`,
        `attention_dropout`, `encoder_layerdrop`; whisper-medium ships all of them at 0 and apply_spec_augment false.
        WhisperEncoder has no feature-projection dropout, and its SpecAugment acts on the log-mel input (not built: off)."""
        return TrainRegularizers(0.0, d.get("dropout", 0.0), d.get("activation_dropout", 0.0), d.get("attention_dropout", 0.0),
                                 d.get("encoder_layerdrop", 0.0), False, 0.0, 10, 2, seed)


def compute_mask_indices(shape: Tuple[int, int], mask_prob: float, mask_length: int, min_masks: int = 0) -> np.ndarray:
    """SpecAugment span mask, restating transformers 4.47 `_compute_mask_indices` (hf:models/hubert/modeling_hubert.py,
    no attention mask) call for call on numpy's GLOBAL RNG, so a seeded run picks the spans the reference would."""
    batch_size, sequence_length = shape
    if mask_length < 1:
        raise ValueError("`mask_length` has to be bigger than 0.")
    if mask_length > sequence_length:
        raise ValueError(f"`mask_length` has to be smaller than `sequence_length`, but got `mask_length`: {mask_length}"
                         f" and `sequence_length`: {sequence_length}`")
    epsilon = np.random.rand(1).item()

    def num_spans(input_length):
        n = int(mask_prob * input_length / mask_length + epsilon)
        n = max(n, min_masks)
        if n * mask_length > sequence_length:
            n = sequence_length // mask_length
        if input_length - (mask_length - 1) < n:
            n = max(input_length - (mask_length - 1), 0)
        return n

    mask = np.zeros((batch_size, sequence_length), dtype=bool)
    max_spans = num_spans(sequence_length)
    if max_spans == 0:
        return mask
    idxs = []
    for _ in range(batch_size):
        n = num_spans(sequence_length)
        idx = np.random.choice(np.arange(sequence_length - (mask_length - 1)), n, replace=False)
        dummy = sequence_length - 1 if len(idx) == 0 else idx[0]
        idxs.append(np.concatenate([idx, np.ones(max_spans - n, dtype=np.int32) * dummy]))
    idxs = np.array(idxs)
    idxs = np.broadcast_to(idxs[:, :, None], (batch_size, max_spans, mask_length)).reshape(batch_size, max_spans * mask_length)
    offsets = np.broadcast_to(np.arange(mask_length)[None, None, :], (batch_size, max_spans, mask_length)).reshape(batch_size, max_spans * mask_length)
    idxs = idxs + offsets
    if idxs.max() > sequence_length - 1:
        idxs[idxs > sequence_length - 1] = sequence_length - 1
    np.put_along_axis(mask, idxs, 1, -1)
    return mask


_SITES = {"fp": 1, "pos": 2, "attn_out": 3, "act": 4, "ffn_out": 5, "layerdrop": 6, "attn_prob": 7}


def _site_seed(base: int, site: str, layer: int = 0) -> int:
    """64-bit seed of one dropout site of one micro-batch (splitmix-style mixing of (base, site, layer))."""
    z = (base * 0x9E3779B97F4A7C15 + _SITES[site] * 0xBF58476D1CE4E5B9 + (layer + 1) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 30; z = (z * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z ^= z >> 27; z = (z * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def _vec(dt) -> int:
    return 4 if dt == torch.float32 else 8


def _rup(n: int, m: int) -> int:
    return (n + m - 1) // m * m


def _offsets(lens: Sequence[int]) -> List[int]:
    o = [0]
    for n in lens:
        o.append(o[-1] + int(n))
    return o


# ------------------------------------------------------------------------------------------------
# frozen LLM: forward with a tape, data-gradient backward — over a PACKED batch of sequences
# ------------------------------------------------------------------------------------------------
class LlamaTape:
    """Frozen LLM over a packed batch: forward with a tape and data-gradient backward.  The per-layer launch sequences live in
    the library's C++ runtime (sl_llama_stack_train_fwd / _bwd, csrc/train_tape.hip); this class owns the buffers."""

    def __init__(self, llm: AudioLlamaForCausalLM):
        self.llm = llm
        self.w = llm._dev()
        self.a = llm.arch
        self._wt: Dict[tuple, torch.Tensor] = {}   # frozen weights stored transposed for the data-gradient GEMMs (+1 copy)
        n = self.a.num_hidden_layers
        self._layers = (L.LlamaTrainLayer * n)()
        for li in range(n):
            lw = self.w.layer_t[li]
            lay = self._layers[li]
            lay.norm1, lay.wqkv, lay.wo, lay.norm2, lay.wgu, lay.wdown = (lw[k].data_ptr() for k in ("norm1", "wqkv", "wo", "norm2", "wgu", "wdown"))
        self._have_t = False
        self._ws: Optional[torch.Tensor] = None

    def _t(self, li: int, name: str) -> torch.Tensor:
        key = (li, name)
        if key not in self._wt:
            self._wt[key] = self.w.layer_t[li][name].t().contiguous()
        return self._wt[key]

    def _cfg(self, n_tok: int, seqlens: Sequence[int], pos: torch.Tensor, dt: torch.dtype):
        a, w = self.a, self.w
        cu, klen = ops.seq_descriptors(seqlens, pos.device)
        c = L.LlamaStackCfg()
        c.dtype, c.hidden, c.n_heads, c.n_kv_heads, c.head_dim, c.ffn = L.dtype_code(dt), a.hidden_size, a.num_attention_heads, a.num_key_value_heads, a.head_dim, a.intermediate_size
        c.n_layers, c.nseq, c.max_len, c.n_tok, c.rms_eps = a.num_hidden_layers, len(seqlens), max(int(n) for n in seqlens), n_tok, a.rms_norm_eps
        c.cu, c.klen, c.pos, c.rope_cos, c.rope_sin = cu.data_ptr(), klen.data_ptr(), pos.data_ptr(), w.rope_cos.data_ptr(), w.rope_sin.data_ptr()
        return c, (cu, klen)

    def _workspace(self, cfg, device) -> torch.Tensor:
        need = int(L.lib().sl_llama_stack_train_workspace_bytes(C.byref(cfg)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def forward(self, x: torch.Tensor, seqlens: Sequence[int], save: bool = True):
        """x: (sum S_i, H) packed embeddings.  Returns (hidden_states: L+1 packed tensors [hidden_states[l] = input of
        layer l, last = post-norm], tape).  With save=False nothing is kept for backward (teacher pass)."""
        a, w = self.a, self.w
        dt, dev, n = x.dtype, x.device, x.shape[0]
        nh, nkv, D, H, F_ = a.num_attention_heads, a.num_key_value_heads, a.head_dim, a.hidden_size, a.intermediate_size
        nl = a.num_hidden_layers
        pos = L.h2d(torch.cat([torch.arange(s_, dtype=torch.int32) for s_ in seqlens]), torch.int32, dev)
        cfg, keep = self._cfg(n, seqlens, pos, dt)
        states = torch.empty((nl, n, H), device=dev, dtype=dt)                 # outputs of the layers = hidden[1..L]
        hidden = [x] + [states[l] for l in range(nl)]
        hptr = (L.c_vp * (nl + 1))(*[h.data_ptr() for h in hidden])
        saved, bufs = None, None
        if save:
            qkv_w, att_w = (nh + 2 * nkv) * D, nh * D
            row = qkv_w + H + 2 * F_ + att_w
            bufs = (torch.empty((nl, n * row), device=dev, dtype=dt), torch.empty((nl, n, nh), device=dev, dtype=torch.float32))
            esz = bufs[0].element_size()
            saved = (L.LlamaLayerSaved * nl)()
            for l in range(nl):
                base = bufs[0][l].data_ptr()
                sv = saved[l]
                sv.qkv, sv.x2, sv.gu, sv.att = base, base + n * qkv_w * esz, base + n * (qkv_w + H) * esz, base + n * (qkv_w + H + 2 * F_) * esz
                sv.lse = bufs[1][l].data_ptr()
        ws = self._workspace(cfg, dev)
        L.check(L.lib().sl_llama_stack_train_fwd(self._layers, C.byref(cfg), hptr, saved, ws.data_ptr(), ws.numel(), L.stream_ptr()), "sl_llama_stack_train_fwd")
        xn = ops.rmsnorm(hidden[nl], w.final_norm, a.rms_norm_eps)
        return hidden[:nl] + [xn], dict(saved=saved, bufs=bufs, hidden=hidden, hptr=hptr, cfg=cfg, keep=keep, x_final=hidden[nl], seqlens=list(seqlens), pos=pos)

    def logits(self, xn_rows: torch.Tensor) -> torch.Tensor:
        return ops.gemm(xn_rows, self.w.lm_head, out_f32=True)

    LM_HEAD_DGRAD_SPLITS = (12, 8, 6, 4, 3, 2)

    def _lm_head_dgrad(self, d_logits: torch.Tensor) -> torch.Tensor:
        """d_xn (n, H) = d_logits (n, V) . lm_head (V, H): few output tiles (n ~ 1 k rows x H) under a 128 k-long reduction.  As one
        transposed-operand product it ran 48 tiles for 2.5 ms (KD window, profiles/r03_ep_kd_gemm_shapes.txt); here the vocabulary is
        cut into slices (the first count of LM_HEAD_DGRAD_SPLITS that leaves whole 128-byte K slabs: 12 for V = 128 256, 0.88 ms;
        tools/time_lm_head_dgrad.py) — one batched launch on a K-contiguous (H, V) copy of the frozen matrix, fp32 partials summed
        in slice order."""
        w = self.w
        n, V = d_logits.shape
        H = w.lm_head.shape[1]
        vec = 4 if d_logits.dtype == torch.float32 else 8
        S = next((s_ for s_ in self.LM_HEAD_DGRAD_SPLITS if V % (s_ * 8 * vec) == 0), 1)
        # from one 128-row tile up (the per-rank window of an 8-rank run has 128 loss rows: as a transposed-operand product it took 3.6 ms
        # of a 39.8 ms window, profiles/r05_e_kd_window2_ops.txt)
        if S <= 1 or n < 96 or not d_logits.is_contiguous():
            return ops.dgrad(d_logits, w.lm_head)
        if getattr(self, "_lm_head_t", None) is None:
            self._lm_head_t = w.lm_head.t().contiguous()        # frozen (ref:trainer.py:63-64): built once
        part = torch.empty((S, n, H), device=d_logits.device, dtype=torch.float32)
        Ks = V // S
        ops.gemm_ex(d_logits, self._lm_head_t, M=n, N=H, K=Ks, lda=V, ldw=V, out=part, ldc=H, out_f32=True, batch=S, strideA=Ks, strideW=Ks,
                    strideC=n * H, dtype=d_logits.dtype)
        return part.sum(0).to(d_logits.dtype)

    def backward(self, tape, tail_rows: torch.Tensor, d_logits_tail: torch.Tensor, d_hidden: Dict[int, torch.Tensor],
                 n_seq: Optional[int] = None) -> torch.Tensor:
        """tail_rows: int64 indices (packed) of the rows whose logits carry loss; d_logits_tail: (len(tail_rows), V)
        gradient (dtype T); d_hidden[l]: (N, H) gradient of hidden_states[l].  Returns d(input embeddings) (N, H).
        n_seq: differentiate only the FIRST n_seq sequences of the forward's packed batch (N = their rows): the KD window runs
        the student and the no-grad teacher sequences through one forward pass, student first; rows are packed sequence by
        sequence, so the student's share of every saved buffer is its leading rows and the teacher rows are simply not visited."""
        a, w = self.a, self.w
        nl = a.num_hidden_layers
        if not self._have_t:                                   # (in, out) copies of the frozen weights, built once
            for li in range(nl):
                lay = self._layers[li]
                lay.wqkv_t, lay.wo_t, lay.wgu_t, lay.wdown_t = (self._t(li, k).data_ptr() for k in ("wqkv", "wo", "wgu", "wdown"))
            self._have_t = True
        cfg = tape["cfg"]
        x_final = tape["x_final"]
        if n_seq is not None and n_seq < len(tape["seqlens"]):
            lens = tape["seqlens"][:n_seq]
            n_rows = int(sum(lens))
            cfg = L.LlamaStackCfg.from_buffer_copy(cfg)           # same descriptors (cu / klen / pos are prefixes too), fewer sequences
            cfg.nseq, cfg.n_tok, cfg.max_len = n_seq, n_rows, max(int(n) for n in lens)
            x_final = x_final[:n_rows]
        dx = torch.zeros_like(x_final)
        d_xn = self._lm_head_dgrad(d_logits_tail)   # (n_tail, H)
        if nl in d_hidden:
            d_xn += d_hidden[nl].index_select(0, tail_rows)
        dx.index_copy_(0, tail_rows, ops.rmsnorm_bwd(x_final.index_select(0, tail_rows), w.final_norm, d_xn, a.rms_norm_eps))
        d_tap = (L.c_vp * nl)(*[(d_hidden[l].data_ptr() if l in d_hidden else None) for l in range(nl)])
        ws = self._workspace(cfg, dx.device)
        L.check(L.lib().sl_llama_stack_train_bwd(self._layers, C.byref(cfg), tape["hptr"], tape["saved"], d_tap, dx.data_ptr(), ws.data_ptr(), ws.numel(),
                                                 L.stream_ptr()), "sl_llama_stack_train_bwd")
        return dx


# ------------------------------------------------------------------------------------------------
# audio encoder: forward with a tape, full backward (data + parameter gradients) — ragged batch of utterances
# ------------------------------------------------------------------------------------------------
class EncoderTape:
    """HuBERT + pool + projection for a batch of utterances of arbitrary lengths, composed op by op from the same
    kernels the fused C runtime (sl_hubert_forward) launches, keeping what the backward needs.  Token-wise work
    (norms, linear layers) runs on the packed frames of the whole batch; convolutions and attention respect
    utterance boundaries (grouped GEMMs / varlen attention), so results equal per-utterance runs."""

    def __init__(self, enc: AudioEncoder):
        if enc.downsample_method != "pool":
            raise L.SpeechLLMError("the KD step is built for the `pool` downsample (all shipped configs use it)")
        self.enc = enc

    @property
    def W(self) -> HubertDeviceWeights:
        return self.enc.weights

    # encoder geometry under the names both architectures' tapes use
    @property
    def hidden(self) -> int:
        return self.enc.arch.hidden_size

    @property
    def n_heads(self) -> int:
        return self.enc.arch.num_attention_heads

    @property
    def ffn(self) -> int:
        return self.enc.arch.intermediate_size

    @property
    def ln_eps(self) -> float:
        return self.enc.arch.layer_norm_eps

    # kernel-layout fp32 gradient buffers, one per weight tensor role: views into ONE flat arena laid out in the order the
    # backward pass finishes them, so that the data-parallel all-reduce runs in place on contiguous slices (dist.GradArena)
    def grad_order(self) -> List[Tuple[str, tuple]]:
        a, W = self.enc.arch, self.W
        names = ["proj_w", "proj_b", "final_ln_g", "final_ln_b"]
        for li in reversed(range(len(W.layer_t))):
            names += [f"l{li}.{k}" for k in W.layer_t[li]]
        names += ["pos_w", "pos_b", "fp_w", "fp_b", "fp_ln_g", "fp_ln_b", "masked_spec_embed"]
        for i in reversed(range(1, len(a.conv_dim))):
            names += [f"conv{i}_w", f"conv{i}_b", f"conv{i}_g", f"conv{i}_beta"]
        names += ["conv0_w", "conv0_b", "conv0_g", "conv0_beta"]
        shapes = {k: tuple(v.shape) for k, v in W.t.items()}
        for li, lt in enumerate(W.layer_t):
            shapes.update({f"l{li}.{k}": tuple(v.shape) for k, v in lt.items()})
        shapes["masked_spec_embed"] = (a.hidden_size,)
        names += [k for k in shapes if k not in names]
        return [(n, shapes[n]) for n in names]

    def new_grads(self) -> Dict[str, torch.Tensor]:
        from .dist import GradArena
        self.arena = GradArena(self.grad_order(), self.enc.device)
        return self.arena.views

    def forward(self, waves: Sequence[torch.Tensor], reg: Optional[TrainRegularizers] = None, step: int = 0,
                masked_spec_embed: Optional[torch.Tensor] = None):
        """`reg` switches the training-mode regularisers on; `step` (micro-batch counter) salts their seeds."""
        W, a, enc = self.W, self.enc.arch, self.enc
        t, dt, dev = W.t, enc.dtype, enc.device
        B = len(waves)
        waves = [wv.reshape(-1).to(device=dev, dtype=torch.float32).contiguous() for wv in waves]
        nc = len(a.conv_dim)
        # per-layer lengths and packed row offsets
        Ls = [[0] * B for _ in range(nc)]
        for u, wv in enumerate(waves):
            n = wv.numel()
            for i in range(nc):
                n = (n - a.conv_kernel[i]) // a.conv_stride[i] + 1
                Ls[i][u] = n
        offs = [_offsets(Ls[i]) for i in range(nc)]
        tape = dict(waves=waves, Ls=Ls, offs=offs)
        flat = torch.cat(waves) if B > 1 else waves[0]
        tape["flat_waves"], tape["soff"] = flat, _offsets([wv.numel() for wv in waves])
        x = ops.hubert_conv0_batch(flat, tape["soff"], offs[0], t["conv0_w"], t["conv0_b"], t["conv0_g"], t["conv0_beta"], dt,
                                   k=a.conv_kernel[0], stride=a.conv_stride[0])      # one launch for the ragged batch
        acts, pres = [x], [None]
        for i in range(1, nc):
            Cin, Cout, k, s = a.conv_dim[i - 1], a.conv_dim[i], a.conv_kernel[i], a.conv_stride[i]
            grp = L.h2d([[Ls[i][u], offs[i - 1][u] * Cin, offs[i][u] * Cout, 0] for u in range(B)], torch.int64, dev)
            c = torch.empty((offs[i][B], Cout), device=dev, dtype=dt)
            ops.gemm_ex(x, t[f"conv{i}_w"], M=max(Ls[i]), N=Cout, K=k * Cin, lda=s * Cin, ldw=k * Cin, out=c, bias=t[f"conv{i}_b"], batch=B,
                        groups=grp, dtype=dt)
            x = ops.layernorm(c, t[f"conv{i}_g"], t[f"conv{i}_beta"], 1e-5, gelu=True)
            acts.append(x); pres.append(c)
        tape.update(acts=acts, pres=pres)
        T = Ls[nc - 1]
        toff = offs[nc - 1]
        NT, H = toff[B], a.hidden_size
        fp_ln = ops.layernorm(x, t["fp_ln_g"], t["fp_ln_b"], a.layer_norm_eps)
        x0 = ops.gemm(fp_ln, t["fp_w"], bias=t["fp_b"])
        base = None if reg is None else (int(reg.seed) * 1000003 + int(step)) & 0xFFFFFFFFFFFFFFFF
        spec_rows = None
        if reg is not None:
            if reg.feat_proj_dropout > 0:
                ops.dropout(x0, reg.feat_proj_dropout, _site_seed(base, "fp"), out=x0)
            if reg.apply_spec_augment and reg.mask_time_prob > 0:          # hf `_mask_hidden_states`: one (1, T) draw per utterance
                rows = []
                for u in range(B):
                    m = compute_mask_indices((1, T[u]), reg.mask_time_prob, reg.mask_time_length, reg.mask_time_min_masks)[0]
                    rows.append(torch.from_numpy(np.nonzero(m)[0]) + toff[u])
                spec_rows = L.h2d(torch.cat(rows), torch.int64, dev)
                if spec_rows.numel():
                    if masked_spec_embed is None:
                        raise L.SpeechLLMError("SpecAugment needs the encoder's masked_spec_embed")
                    x0.index_copy_(0, spec_rows, masked_spec_embed.to(device=dev, dtype=dt).reshape(1, -1).expand(spec_rows.numel(), -1).contiguous())
        G, k = a.num_conv_pos_embedding_groups, a.num_conv_pos_embeddings
        Hg = H // G
        xg_off = _offsets([(T[u] + k) * H for u in range(B)])
        xg = ops.posconv_stage_batch(x0, T, G, k)                                   # one launch; utterance u at xg_off[u]
        pos_grp = L.h2d([[T[u], xg_off[u] + g * (T[u] + k) * Hg, toff[u] * H + g * Hg, toff[u] * H + g * Hg] for u in range(B) for g in range(G)],
                        torch.int64, dev)
        pre_pos = torch.empty_like(x0)
        x1 = torch.empty_like(x0)
        ops.gemm_ex(xg, t["pos_w"], M=max(T), N=Hg, K=k * Hg, lda=Hg, ldw=k * Hg, out=x1, ldc=H, bias=t["pos_b"], residual=x0, ldr=H, act=L.ACT_GELU,
                    aux_out=pre_pos, batch=B * G, strideW=Hg * k * Hg, strideBias=Hg, groups=pos_grp, w_mod=G, dtype=dt)
        if reg is not None and reg.hidden_dropout > 0:                     # encoder: hidden_states = dropout(hidden_states + pos_conv_embed(...))
            ops.dropout(x1, reg.hidden_dropout, _site_seed(base, "pos"), out=x1)
        tape.update(fp_ln=fp_ln, xg=xg, xg_off=xg_off, pre_pos=pre_pos, T=T, toff=toff, reg=reg, base=base, spec_rows=spec_rows)
        x, layers = self._stack_forward(x1, T, reg, base)
        out, head = self._head_forward(x, T, toff)
        tape.update(layers=layers, x_last=x, **head)
        return out, tape

    # -- the pre-LN transformer layers shared by the HuBERT and Whisper encoders (hf:...hubert.py:504-547 stable-LN layer,
    #    hf:models/whisper/modeling_whisper.py:360-414), on the packed frames of the batch: ONE call into the library's C++ tape
    #    runtime (sl_encoder_stack_train_fwd, csrc/train_tape.hip) issues the launches of all layers
    def _stack_forward(self, x: torch.Tensor, T: Sequence[int], reg: Optional[TrainRegularizers], base: Optional[int]):
        W, enc = self.W, self.enc
        dt, dev = enc.dtype, enc.device
        H, nh, F_, eps = self.hidden, self.n_heads, self.ffn, self.ln_eps
        NT, nl = x.shape[0], len(W.layer_t)
        skip = (C.c_uint8 * nl)()
        seeds = (C.c_uint64 * (4 * nl))()
        for li in range(nl):
            if reg is not None and reg.layerdrop > 0:                      # hf: skip the layer when a uniform draw < layerdrop
                u01 = (_site_seed(base, "layerdrop", li) >> 11) * (1.0 / 9007199254740992.0)
                skip[li] = 1 if u01 < reg.layerdrop else 0
            if reg is not None:
                for j, site in enumerate(("attn_prob", "attn_out", "act", "ffn_out")):
                    seeds[4 * li + j] = _site_seed(base, site, li)
        cu, klen = ops.seq_descriptors(T, dev)
        cfg = L.EncStackCfg()
        cfg.dtype, cfg.hidden, cfg.n_heads, cfg.ffn, cfg.n_layers, cfg.nseq, cfg.max_len, cfg.n_tok = L.dtype_code(dt), H, nh, F_, nl, len(T), max(int(t_) for t_ in T), NT
        cfg.ln_eps = eps
        cfg.p_hidden = reg.hidden_dropout if reg is not None else 0.0
        cfg.p_act = reg.activation_dropout if reg is not None else 0.0
        cfg.p_attn = reg.attention_dropout if reg is not None else 0.0
        cfg.cu, cfg.klen = cu.data_ptr(), klen.data_ptr()
        cfg.skip, cfg.seeds = C.cast(skip, C.c_void_p).value, C.cast(seeds, C.c_void_p).value
        live = [li for li in range(nl) if not skip[li]]
        row = 8 * H + 2 * F_                                               # ln1, qkv (3H), att, x_mid, ln2, x_out + pre1, mid
        bufs = torch.empty((max(1, len(live)), NT * row), device=dev, dtype=dt)
        lses = torch.empty((max(1, len(live)), NT, nh), device=dev, dtype=torch.float32)
        esz = bufs.element_size()
        saved = (L.EncLayerSaved * nl)()
        for j, li in enumerate(live):
            p0, sv = bufs[j].data_ptr(), saved[li]
            o = 0
            for name, width in (("ln1", H), ("qkv", 3 * H), ("att", H), ("x_mid", H), ("ln2", H), ("pre1", F_), ("mid", F_), ("x_out", H)):
                setattr(sv, name, p0 + o * NT * esz)
                o += width
            sv.lse = lses[j].data_ptr()
        need = int(L.lib().sl_encoder_stack_train_workspace_bytes(C.byref(cfg)))
        if getattr(self, "_ws", None) is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        x_out = C.c_void_p(0)
        L.check(L.lib().sl_encoder_stack_train_fwd(W._layers, C.byref(cfg), x.data_ptr(), saved, C.byref(x_out), self._ws.data_ptr(), self._ws.numel(),
                                                   L.stream_ptr()), "sl_encoder_stack_train_fwd")
        if live:                                                           # the last live layer's x_out, as a tensor view
            off = 7 * H + 2 * F_                                           # x_out is the last block of the layer's buffer
            out = bufs[len(live) - 1][off * NT:(off + H) * NT].view(NT, H)
            assert out.data_ptr() == x_out.value
        else:
            out = x
        return out, dict(cfg=cfg, saved=saved, keep=(skip, seeds, cu, klen, bufs, lses, x), skip=[bool(v) for v in skip])

    # -- final LayerNorm, AvgPool1d over time, projection (ref:model/audio_encoder.py:56-63,87)
    def _head_forward(self, x: torch.Tensor, T: Sequence[int], toff: Sequence[int]):
        W, enc = self.W, self.enc
        t, dt, dev = W.t, enc.dtype, enc.device
        B, H = len(T), self.hidden
        lnf = ops.layernorm(x, t["final_ln_g"], t["final_ln_b"], self.ln_eps)
        pooled, P = ops.avgpool_batch(lnf, T, enc.pool_kernel, enc.pool_stride)      # one launch for the ragged batch
        poff = _offsets(P)
        out = ops.gemm(pooled, t["proj_w"], bias=t["proj_b"])
        return out, dict(pooled=pooled, P=P, poff=poff)

    def backward(self, tape, d_out: torch.Tensor, g: Dict[str, torch.Tensor], on_bucket=None) -> None:
        """Accumulates fp32 parameter gradients into `g` (kernel layouts).  `on_bucket(names)` is called as soon as
        a group of gradient buffers is final for this optimizer step (used to launch their all-reduce early)."""
        W, a, enc = self.W, self.enc.arch, self.enc
        t, dt = W.t, enc.dtype
        H, T, toff, poff = a.hidden_size, tape["T"], tape["toff"], tape["poff"]
        B = len(T)
        NT = toff[B]
        nh = a.num_attention_heads
        done = on_bucket or (lambda names: None)
        dx = self._head_backward(tape, d_out, g, done)
        # `on_bucket` is what the chunks are for (finished gradient buffers go to the data-parallel reducer while the rest still runs); with nobody listening
        # the whole stack is ONE call: every call boundary joins the side stream of the parameter-gradient products (main stream idle until the lagging
        # products of the chunk have finished) and costs the first layer of the next call its fused dropout.  stack_chunk forces a size (bench.py's
        # per-rank probe keeps the reducer's hand-over points).
        chunk = getattr(self, "stack_chunk", None) or (4 if on_bucket is not None else len(W.layer_t))
        dx = self._stack_backward(tape, dx, g, done, chunk=chunk)
        reg, base = tape.get("reg"), tape.get("base")
        p_h = reg.hidden_dropout if reg is not None else 0.0
        # positional conv: x1 = x0 + gelu(conv(x0) + b)
        G, k = a.num_conv_pos_embedding_groups, a.num_conv_pos_embeddings
        Hg = H // G
        if p_h > 0:
            dx = ops.dropout(dx, p_h, _site_seed(base, "pos"))
        d_pre = ops.gelu_bwd(dx, tape["pre_pos"])
        ops.colsum_acc(d_pre, g["pos_b"])
        xg, xg_off = tape["xg"], tape["xg_off"]
        wd = t["pos_w"].view(G, Hg, k, Hg).flip(2).permute(0, 3, 2, 1).contiguous().view(G, Hg, k * Hg)  # [g][c][jj][n]
        dx0 = torch.empty_like(dx)
        for u in range(B):
            Tu, r0 = T[u], toff[u]
            # wgrad: dW[g][n][j*Hg + c] += sum_t d_pre[t][g*Hg + n] * xg[g][t + j][c]   (overlapping windows as the W operand)
            ops.gemm_ex(d_pre, xg, M=Hg, N=k * Hg, K=Tu, lda=H, ldw=Hg, out=g["pos_w"], ldc=k * Hg, residual=g["pos_w"], ldr=k * Hg,
                        out_f32=True, residual_f32=True, trans_a=True, trans_w=True, batch=G, strideA=Hg, strideW=(Tu + k) * Hg, strideC=Hg * k * Hg,
                        strideR=Hg * k * Hg, a_off=r0 * H, w_off=xg_off[u], dtype=dt)
        # dgrad: correlation with the flipped, transposed weight over the staged d_pre (windows start one row later) — every utterance and group in ONE
        # staging launch + ONE grouped product (the forward's records shifted by a row; was a staging launch + a 16-group product per utterance)
        dpg = ops.posconv_stage_batch(d_pre, T, G, k)
        grp = L.h2d([[T[u], xg_off[u] + g * (T[u] + k) * Hg + Hg, toff[u] * H + g * Hg, toff[u] * H + g * Hg] for u in range(B) for g in range(G)], torch.int64, dx.device)
        ops.gemm_ex(dpg, wd, M=max(T), N=Hg, K=k * Hg, lda=Hg, ldw=k * Hg, out=dx0, ldc=H, residual=dx, ldr=H, batch=B * G, strideW=Hg * k * Hg, groups=grp,
                    w_mod=G, dtype=dt)
        # SpecAugment rows took the learned mask embedding; feature-projection dropout
        spec_rows = tape.get("spec_rows")
        if spec_rows is not None and spec_rows.numel():
            g["masked_spec_embed"] += dx0.index_select(0, spec_rows).float().sum(0)
            dx0.index_fill_(0, spec_rows, 0.0)
        if reg is not None and reg.feat_proj_dropout > 0:
            ops.dropout(dx0, reg.feat_proj_dropout, _site_seed(base, "fp"), out=dx0)
        # feature projection
        ops.wgrad_acc(dx0, tape["fp_ln"], g["fp_w"], db=g["fp_b"])
        d_fpln = ops.dgrad(dx0, t["fp_w"])
        acts, pres, offs, Ls = tape["acts"], tape["pres"], tape["offs"], tape["Ls"]
        d_act = ops.layernorm_bwd(acts[-1], t["fp_ln_g"], t["fp_ln_b"], d_fpln, a.layer_norm_eps, g["fp_ln_g"], g["fp_ln_b"])
        done(["pos_w", "pos_b", "fp_w", "fp_b", "fp_ln_g", "fp_ln_b", "masked_spec_embed"])
        # conv stack
        for i in reversed(range(1, len(a.conv_dim))):
            Cin, Cout, kk, s = a.conv_dim[i - 1], a.conv_dim[i], a.conv_kernel[i], a.conv_stride[i]
            d_c = ops.layernorm_bwd(pres[i], t[f"conv{i}_g"], t[f"conv{i}_beta"], d_act, 1e-5, g[f"conv{i}_g"], g[f"conv{i}_beta"], gelu=True)
            ops.colsum_acc(d_c, g[f"conv{i}_b"])
            dcol = ops.dgrad(d_c, t[f"conv{i}_w"], wt=ops.transpose_pad(t[f"conv{i}_w"], t[f"conv{i}_w"].shape[0], t[f"conv{i}_w"].shape[1]))
            d_prev = torch.empty((offs[i - 1][B], Cin), device=d_c.device, dtype=dt)
            # weight gradient: every utterance's windows in ONE grouped launch (per-group reduction length = its frame count),
            # fp32 partials per utterance summed afterwards — per-utterance launches had 48 tiles each and ran 0.6 ms apiece
            Kin = kk * Cin
            part = torch.empty((B, Cout, Kin), device=d_c.device, dtype=torch.float32)
            recs = [[Cout, offs[i][u] * Cout, u * Cout * Kin, 0, offs[i - 1][u] * Cin, Kin, offs[i][u + 1] - offs[i][u], 0] for u in range(B)]
            grp = L.h2d(recs, torch.int64, d_c.device)
            ops.gemm_ex(d_c, acts[i - 1], M=Cout, N=Kin, K=max(r_[6] for r_ in recs), lda=Cout, ldw=s * Cin, out=part, ldc=Kin, out_f32=True,
                        trans_a=True, trans_w=True, batch=B, dtype=dt, groups=grp, groups_ext=True)
            g[f"conv{i}_w"] += part.sum(0)
            # every utterance's windows folded back in ONE launch, straight into the packed rows (was a launch + a copy per utterance)
            ops.col2im_batch(dcol, d_prev, [offs[i][u + 1] - offs[i][u] for u in range(B)], [offs[i - 1][u + 1] - offs[i - 1][u] for u in range(B)],
                             [offs[i][u] for u in range(B)], [offs[i - 1][u] for u in range(B)], Cin, kk, s)
            d_act = d_prev
            done([f"conv{i}_w", f"conv{i}_b", f"conv{i}_g", f"conv{i}_beta"])
        ops.hubert_conv0_bwd_batch(tape["waves"], t["conv0_w"], t["conv0_b"], t["conv0_g"], t["conv0_beta"], d_act, offs[0], g["conv0_w"], g["conv0_b"],
                                   g["conv0_g"], g["conv0_beta"], k=a.conv_kernel[0], stride=a.conv_stride[0])
        done(["conv0_w", "conv0_b", "conv0_g", "conv0_beta"])


    def _head_backward(self, tape, d_out: torch.Tensor, g: Dict[str, torch.Tensor], done) -> torch.Tensor:
        W, enc = self.W, self.enc
        t, dt = W.t, enc.dtype
        H, T, toff, poff = self.hidden, tape["T"], tape["toff"], tape["poff"]
        B, NT = len(T), tape["toff"][len(T)]
        ops.wgrad_acc(d_out, tape["pooled"], g["proj_w"], db=g["proj_b"])
        d_pooled = ops.dgrad(d_out, t["proj_w"])
        d_lnf = torch.empty((NT, H), device=d_out.device, dtype=dt)
        ops.avgpool_bwd_batch(d_pooled, d_lnf, [poff[u + 1] - poff[u] for u in range(B)], T, [poff[u] for u in range(B)], [toff[u] for u in range(B)],
                              enc.pool_kernel, enc.pool_stride)
        dx = ops.layernorm_bwd(tape["x_last"], t["final_ln_g"], t["final_ln_b"], d_lnf, self.ln_eps, g["final_ln_g"], g["final_ln_b"])
        done(["proj_w", "proj_b", "final_ln_g", "final_ln_b"])
        return dx

    def _stack_backward(self, tape, dx: torch.Tensor, g: Dict[str, torch.Tensor], done, chunk: int = 4) -> torch.Tensor:
        """Backward of the layer stack in the C++ tape runtime (sl_encoder_stack_train_bwd), `chunk` layers per call so that
        `done` can hand finished gradient buffers to the data-parallel reducer while the rest still runs.  dx is updated in place."""
        W = self.W
        st = tape["layers"]
        nl = len(W.layer_t)
        grads = (L.EncLayerGrads * nl)()
        for li in range(nl):
            for name, _ in L.EncLayerGrads._fields_:
                setattr(grads[li], name, g[f"l{li}.{name}"].data_ptr())
        dx = dx.contiguous()
        ws = self._ws
        for hi in range(nl, 0, -chunk):
            lo = max(0, hi - chunk)
            L.check(L.lib().sl_encoder_stack_train_bwd(W._layers, C.byref(st["cfg"]), st["saved"], grads, lo, hi, dx.data_ptr(), ws.data_ptr(), ws.numel(),
                                                       L.stream_ptr()), "sl_encoder_stack_train_bwd")
            for li in reversed(range(lo, hi)):
                done([f"l{li}.{n}" for n in W.layer_t[li]])
        return dx


def kernel_grads_to_state_dict(enc: AudioEncoder, g: Dict[str, torch.Tensor], master: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Map kernel-layout fp32 gradients back onto the reference's state-dict parameter names/shapes
    (inverse of HubertDeviceWeights' re-layouts; weight-norm backward for the positional conv)."""
    a = enc.arch
    H, G, k = a.hidden_size, a.num_conv_pos_embedding_groups, a.num_conv_pos_embeddings
    Hg = H // G
    out: Dict[str, torch.Tensor] = {}
    p = "encoder.feature_extractor.conv_layers."
    out[p + "0.conv.weight"] = g["conv0_w"].view(a.conv_dim[0], 1, -1)
    out[p + "0.conv.bias"], out[p + "0.layer_norm.weight"], out[p + "0.layer_norm.bias"] = g["conv0_b"], g["conv0_g"], g["conv0_beta"]
    for i in range(1, len(a.conv_dim)):
        Cin, Cout, kk = a.conv_dim[i - 1], a.conv_dim[i], a.conv_kernel[i]
        out[p + f"{i}.conv.weight"] = g[f"conv{i}_w"].view(Cout, kk, Cin).permute(0, 2, 1).contiguous()
        out[p + f"{i}.conv.bias"], out[p + f"{i}.layer_norm.weight"], out[p + f"{i}.layer_norm.bias"] = g[f"conv{i}_b"], g[f"conv{i}_g"], g[f"conv{i}_beta"]
    p = "encoder.feature_projection."
    out[p + "layer_norm.weight"], out[p + "layer_norm.bias"] = g["fp_ln_g"], g["fp_ln_b"]
    out[p + "projection.weight"], out[p + "projection.bias"] = g["fp_w"], g["fp_b"]
    # positional conv: d folded weight (H, Hg, k) -> weight-norm backward (g = original0 (1,1,k), v = original1)
    p = "encoder.encoder.pos_conv_embed.conv."
    k0 = p + "parametrizations.weight.original0" if p + "parametrizations.weight.original0" in master else p + "weight_g"
    k1 = p + "parametrizations.weight.original1" if p + "parametrizations.weight.original1" in master else p + "weight_v"
    gw, v = master[k0], master[k1]
    if g["pos_w"].is_cuda and g["pos_w"].dtype == torch.float32 and v.dtype == torch.float32 and v.is_contiguous() and k <= 256:
        # sl_weight_norm_bwd: dW stays in the kernel layout (H, k, Hg); two HIP launches, fixed summation order
        out[k0], out[k1] = torch.empty_like(gw), torch.empty_like(v)
        ws = torch.empty(int(L.lib().sl_weight_norm_bwd_workspace_bytes(k)) // 4, device=v.device, dtype=torch.float32)
        L.check(L.lib().sl_weight_norm_bwd(L.ptr(g["pos_w"]), L.ptr(v), L.ptr(gw.contiguous()), L.ptr(out[k0]), L.ptr(out[k1]), L.ptr(ws), H, Hg, k,
                                           L.stream_ptr()), "sl_weight_norm_bwd")
    else:                    # host tensors / probing images of direct_refresh_map (float64 index images): the defining formulas
        dW = g["pos_w"].view(H, k, Hg).permute(0, 2, 1).contiguous()
        gw, v = gw.to(dW.dtype), v.to(dW.dtype)
        norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
        dot = (dW * v).sum(dim=(0, 1), keepdim=True)
        out[k0] = dot / norm
        out[k1] = (gw / norm) * (dW - v * dot / norm.pow(2))
    out[p + "bias"] = g["pos_b"]
    for li in range(a.num_hidden_layers):
        p, q = f"encoder.encoder.layers.{li}.", f"l{li}."
        wq, wk, wv = g[q + "wqkv"].split(H, dim=0)
        bq, bk, bv = g[q + "bqkv"].split(H, dim=0)
        out[p + "attention.q_proj.weight"], out[p + "attention.k_proj.weight"], out[p + "attention.v_proj.weight"] = wq, wk, wv
        out[p + "attention.q_proj.bias"], out[p + "attention.k_proj.bias"], out[p + "attention.v_proj.bias"] = bq, bk, bv
        out[p + "attention.out_proj.weight"], out[p + "attention.out_proj.bias"] = g[q + "wo"], g[q + "bo"]
        out[p + "layer_norm.weight"], out[p + "layer_norm.bias"] = g[q + "ln1_g"], g[q + "ln1_b"]
        out[p + "final_layer_norm.weight"], out[p + "final_layer_norm.bias"] = g[q + "ln2_g"], g[q + "ln2_b"]
        out[p + "feed_forward.intermediate_dense.weight"], out[p + "feed_forward.intermediate_dense.bias"] = g[q + "w1"], g[q + "b1"]
        out[p + "feed_forward.output_dense.weight"], out[p + "feed_forward.output_dense.bias"] = g[q + "w2"], g[q + "b2"]
    out["encoder.encoder.layer_norm.weight"], out["encoder.encoder.layer_norm.bias"] = g["final_ln_g"], g["final_ln_b"]
    out["embed_projection.weight"], out["embed_projection.bias"] = g["proj_w"], g["proj_b"]
    out["encoder.masked_spec_embed"] = g["masked_spec_embed"]
    return out


class WhisperEncoderTape(EncoderTape):
    """Whisper encoder + pool + projection with a tape (hf:models/whisper/modeling_whisper.py:592-646; the reference trains
    it through the same loop, ref:trainer.py:168-199, 278-291): log-mel on the GPU (no gradient), conv1(k3,p1)+GELU and
    conv2(k3,s2,p1)+GELU as implicit GEMMs over zero-haloed channel-last rows, + the fixed positional table, then the shared
    pre-LN layer stack.  Whisper's own dropouts / LayerDrop / SpecAugment are 0 / off in the shipped configs, so train() and
    eval() coincide; `reg` is still honoured by the layer stack.  Every utterance is one padded 30 s window; the embeddings
    are cropped to `compute_num_audio_embeds(len(audio))` rows as the reference does."""

    def __init__(self, enc: AudioEncoder):
        if enc.downsample_method != "pool":
            raise L.SpeechLLMError("the KD step is built for the `pool` downsample (all shipped configs use it)")
        self.enc = enc

    @property
    def hidden(self) -> int:
        return self.enc.arch.d_model

    @property
    def n_heads(self) -> int:
        return self.enc.arch.encoder_attention_heads

    @property
    def ffn(self) -> int:
        return self.enc.arch.encoder_ffn_dim

    @property
    def ln_eps(self) -> float:
        return 1e-5

    def grad_order(self) -> List[Tuple[str, tuple]]:
        W = self.W
        names = ["proj_w", "proj_b", "final_ln_g", "final_ln_b"]
        for li in reversed(range(len(W.layer_t))):
            names += [f"l{li}.{k}" for k in W.layer_t[li]]
        names += ["conv1_w", "conv1_b", "conv2_w", "conv2_b"]
        shapes = {k: tuple(v.shape) for k, v in W.t.items() if k != "pos"}
        for li, lt in enumerate(W.layer_t):
            shapes.update({f"l{li}.{k}": tuple(v.shape) for k, v in lt.items()})
        names += [k for k in shapes if k not in names]
        return [(n, shapes[n]) for n in names]

    def forward(self, waves: Sequence[torch.Tensor], reg: Optional[TrainRegularizers] = None, step: int = 0, masked_spec_embed=None):
        from .utils import compute_num_audio_embeds
        W, a, enc = self.W, self.enc.arch, self.enc
        t, dt, dev = W.t, enc.dtype, enc.device
        B, H, nm, F_, T1 = len(waves), a.d_model, a.num_mel_bins, a.n_frames, a.max_source_positions
        lens = [int(torch.as_tensor(wv).numel()) for wv in waves]
        feats = enc.feature_extractor(list(waves), return_tensors="pt", sampling_rate=a.sampling_rate).input_features   # (B, nm, F) view of (B, F, nm)
        a1 = torch.zeros((B, F_ + 2, nm), device=dev, dtype=dt)                 # one zero row before / after each window: Conv1d padding=1
        a1[:, 1:F_ + 1] = feats.transpose(1, 2).to(dt)
        b1 = torch.zeros((B, F_ + 2, H), device=dev, dtype=dt)
        pre1 = torch.empty((B, F_ + 2, H), device=dev, dtype=dt)
        ops.gemm_ex(a1, t["conv1_w"], M=F_, N=H, K=3 * nm, lda=nm, ldw=3 * nm, out=b1, ldc=H, bias=t["conv1_b"], act=L.ACT_GELU,
                    aux_out=pre1.view(-1)[H:], batch=B, strideA=(F_ + 2) * nm, strideC=(F_ + 2) * H, c_off=H, dtype=dt)
        x0 = torch.empty((B * T1, H), device=dev, dtype=dt)
        pre2 = torch.empty_like(x0)
        ops.gemm_ex(b1, t["conv2_w"], M=T1, N=H, K=3 * H, lda=2 * H, ldw=3 * H, out=x0, ldc=H, bias=t["conv2_b"], act=L.ACT_GELU, aux_out=pre2,
                    residual=t["pos"], ldr=H, batch=B, strideA=(F_ + 2) * H, strideC=T1 * H, strideR=0, dtype=dt)
        T = [T1] * B
        toff = _offsets(T)
        base = None if reg is None else (int(reg.seed) * 1000003 + int(step)) & 0xFFFFFFFFFFFFFFFF
        tape = dict(a1=a1, b1=b1, pre1=pre1, pre2=pre2, T=T, toff=toff, reg=reg, base=base)
        x, layers = self._stack_forward(x0, T, reg, base)
        full, head = self._head_forward(x, T, toff)
        Pfull = head["P"][0]
        keep = [min(Pfull, max(0, compute_num_audio_embeds(n, sr=a.sampling_rate))) for n in lens]       # ref:trainer.py:283-289
        rows = L.h2d(torch.cat([torch.arange(u * Pfull, u * Pfull + keep[u]) for u in range(B)]), torch.int64, dev)
        out = full.index_select(0, rows)
        tape.update(layers=layers, x_last=x, pooled=head["pooled"], P=keep, poff=_offsets(keep), P_full=head["P"], poff_full=head["poff"], rows=rows)
        return out, tape

    def backward(self, tape, d_out: torch.Tensor, g: Dict[str, torch.Tensor], on_bucket=None) -> None:
        W, a, enc = self.W, self.enc.arch, self.enc
        t, dt = W.t, enc.dtype
        B, H, nm, F_, T1 = len(tape["T"]), a.d_model, a.num_mel_bins, a.n_frames, a.max_source_positions
        done = on_bucket or (lambda names: None)
        d_full = torch.zeros((tape["poff_full"][B], d_out.shape[1]), device=d_out.device, dtype=dt)      # cropped rows carry the gradient
        d_full.index_copy_(0, tape["rows"], d_out)
        full = dict(tape, P=tape["P_full"], poff=tape["poff_full"])
        dx = self._head_backward(full, d_full, g, done)
        dx = self._stack_backward(tape, dx, g, done, chunk=getattr(self, "stack_chunk", None) or (4 if on_bucket is not None else len(W.layer_t)))
        # x0 = gelu(conv2(b1) + bias) + pos  (the positional table is frozen, hf:...whisper.py:606)
        d_pre2 = ops.gelu_bwd(dx, tape["pre2"])
        ops.colsum_acc(d_pre2, g["conv2_b"])
        b1, a1 = tape["b1"], tape["a1"]
        part = torch.empty((B, H, 3 * H), device=dx.device, dtype=torch.float32)
        ops.gemm_ex(d_pre2, b1, M=H, N=3 * H, K=T1, lda=H, ldw=2 * H, out=part, ldc=3 * H, out_f32=True, trans_a=True, trans_w=True, batch=B,
                    strideA=T1 * H, strideW=(F_ + 2) * H, strideC=H * 3 * H, dtype=dt)
        g["conv2_w"] += part.sum(0)
        dcol = ops.dgrad(d_pre2, t["conv2_w"], wt=ops.transpose_pad(t["conv2_w"], t["conv2_w"].shape[0], t["conv2_w"].shape[1]))                        # (B*T1, 3H): windows of the haloed rows
        d_b1 = torch.empty((B, F_, H), device=dx.device, dtype=dt)
        for u in range(B):
            d_b1[u] = ops.col2im(dcol[u * T1:(u + 1) * T1], F_ + 2, H, 3, 2)[1:F_ + 1]
        d_pre1 = ops.gelu_bwd(d_b1.view(B * F_, H), tape["pre1"][:, 1:F_ + 1].reshape(B * F_, H))
        ops.colsum_acc(d_pre1, g["conv1_b"])
        part1 = torch.empty((B, H, 3 * nm), device=dx.device, dtype=torch.float32)
        ops.gemm_ex(d_pre1, a1, M=H, N=3 * nm, K=F_, lda=H, ldw=nm, out=part1, ldc=3 * nm, out_f32=True, trans_a=True, trans_w=True, batch=B,
                    strideA=F_ * H, strideW=(F_ + 2) * nm, strideC=H * 3 * nm, dtype=dt)
        g["conv1_w"] += part1.sum(0)
        done(["conv1_w", "conv1_b", "conv2_w", "conv2_b"])


def whisper_grads_to_state_dict(enc: AudioEncoder, g: Dict[str, torch.Tensor], master: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Kernel-layout gradients -> the reference's Whisper state-dict names (inverse of WhisperDeviceWeights' re-layouts).
    k_proj has no bias (its slot of the fused qkv bias is dropped); embed_positions is frozen in HF."""
    a = enc.arch
    H = a.d_model
    out: Dict[str, torch.Tensor] = {}
    for name, cin in (("conv1", a.num_mel_bins), ("conv2", H)):
        out[f"encoder.{name}.weight"] = g[f"{name}_w"].view(H, 3, cin).permute(0, 2, 1).contiguous()
        out[f"encoder.{name}.bias"] = g[f"{name}_b"]
    for li in range(a.encoder_layers):
        p, q = f"encoder.layers.{li}.", f"l{li}."
        wq, wk, wv = g[q + "wqkv"].split(H, dim=0)
        bq, _, bv = g[q + "bqkv"].split(H, dim=0)
        out[p + "self_attn.q_proj.weight"], out[p + "self_attn.k_proj.weight"], out[p + "self_attn.v_proj.weight"] = wq, wk, wv
        out[p + "self_attn.q_proj.bias"], out[p + "self_attn.v_proj.bias"] = bq, bv
        out[p + "self_attn.out_proj.weight"], out[p + "self_attn.out_proj.bias"] = g[q + "wo"], g[q + "bo"]
        out[p + "self_attn_layer_norm.weight"], out[p + "self_attn_layer_norm.bias"] = g[q + "ln1_g"], g[q + "ln1_b"]
        out[p + "final_layer_norm.weight"], out[p + "final_layer_norm.bias"] = g[q + "ln2_g"], g[q + "ln2_b"]
        out[p + "fc1.weight"], out[p + "fc1.bias"] = g[q + "w1"], g[q + "b1"]
        out[p + "fc2.weight"], out[p + "fc2.bias"] = g[q + "w2"], g[q + "b2"]
    out["encoder.layer_norm.weight"], out["encoder.layer_norm.bias"] = g["final_ln_g"], g["final_ln_b"]
    out["embed_projection.weight"], out["embed_projection.bias"] = g["proj_w"], g["proj_b"]
    return out


# ------------------------------------------------------------------------------------------------
# optimizer step on the device: AdamW + the kernels' weight copies in one launch
# ------------------------------------------------------------------------------------------------
def direct_refresh_map(enc: AudioEncoder, to_state_dict, grads: Dict[str, torch.Tensor], master: Dict[str, torch.Tensor]):
    """Which state-dict parameters sit in the kernels' weight tensors in the SAME element order (so that the optimizer kernel can
    write their compute-dtype copy itself), found by probing rather than by a hand-kept table: every kernel-layout role is filled
    with its own element indices and sent through the gradient re-layout `to_state_dict` (the exact inverse of the weight
    re-layout); a parameter whose image is a run of consecutive indices of one role is a contiguous slice of that role's device
    tensor.  Returns ({state-dict key: (role, element offset)}, {roles that hold anything else})."""
    dev = enc.device
    roles = list(grads)
    fake = {k: (torch.arange(grads[k].numel(), dtype=torch.float64, device=dev) + float(i << 40)).view(grads[k].shape) for i, k in enumerate(roles)}
    image = to_state_dict(enc, fake, master)
    direct, covered = {}, {}
    for key, t in image.items():
        if key not in master or t.dtype != torch.float64:
            continue
        o = t.reshape(-1)
        first = int(o[0].item())
        rid, off = first >> 40, first & ((1 << 40) - 1)
        if 0 <= rid < len(roles) and o.numel() == master[key].numel() and off + o.numel() <= grads[roles[rid]].numel() and \
                bool(torch.equal(o, torch.arange(o.numel(), dtype=torch.float64, device=dev) + float(first))):
            direct[key] = (roles[rid], off)
            covered[roles[rid]] = covered.get(roles[rid], 0) + o.numel()
    indirect = {k for k in roles if covered.get(k, 0) != grads[k].numel()}
    return direct, indirect


class FusedAdamW:
    """torch.optim.AdamW.step() as ONE HIP launch (sl_adamw_step) over the optimizer's own state tensors — the state dict stays
    torch's, so checkpoints keep the reference's layout (ref:trainer.py:97-105, 516-528) — writing, in the same pass, the
    compute-dtype kernel copy of every weight that is laid out like its parameter."""

    def __init__(self, optimizer: torch.optim.AdamW, named_params: Sequence[Tuple[str, torch.nn.Parameter]], dst_of: Dict[str, Tuple[torch.Tensor, int]]):
        self.opt, self.named, self.dst_of = optimizer, list(named_params), dst_of
        g = optimizer.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or len(optimizer.param_groups) != 1:
            raise L.SpeechLLMError("FusedAdamW covers plain AdamW (one param group, amsgrad / maximize off)")
        lib = L.lib()
        self._blocks = [int(lib.sl_adamw_blocks(p.numel())) for _, p in self.named]
        self._cache = {}                        # record tables of the `only` sets of step(), by name tuple

    def step(self, only=None) -> None:
        """only: a set of parameter names — the step of just those (KDTrainer._early_step); launched on the CURRENT stream.  The record
        table of an `only` set is built and uploaded once and reused (masters, moments, arena gradients and the kernels' weight copies
        keep their addresses; `invalidate()` after anything that replaces them, e.g. loading an optimizer checkpoint)."""
        g = self.opt.param_groups[0]
        lib = L.lib()
        ckey = None if only is None else (only if isinstance(only, tuple) else tuple(sorted(only)))
        hit = self._cache.get(ckey) if ckey is not None else None
        if hit is not None:
            t_dev, f_dev, n_live, total, steps = hit
            t_now = int(steps[0]) + 1
        else:
            live = [(k, p, nb) for (k, p), nb in zip(self.named, self._blocks) if p.grad is not None and (only is None or k in only)]
            if not live:
                return
            dev = live[0][1].device
            table = (L.AdamWTensor * len(live))()
            first = np.zeros(len(live), dtype=np.int64)
            steps, total, t_now = [], 0, None
            for i, (k, p, nb) in enumerate(live):
                st = self.opt.state[p]
                if len(st) == 0:                     # torch creates the state on a parameter's first step
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                t_i = int(st["step"]) + 1
                if t_now is None:
                    t_now = t_i
                elif t_i != t_now:
                    raise L.SpeechLLMError("FusedAdamW: parameters disagree on the step count")
                grad = p.grad
                if not (grad.is_contiguous() and grad.dtype == torch.float32 and p.dtype == torch.float32 and p.is_contiguous()):
                    raise L.SpeechLLMError(f"FusedAdamW: parameter {k} / its gradient must be contiguous fp32")
                rec = table[i]
                rec.p, rec.g, rec.m, rec.v, rec.n = p.data_ptr(), grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()
                d = self.dst_of.get(k)
                if d is not None:
                    rec.dst, rec.dst_dtype = d[0].data_ptr() + d[1] * d[0].element_size(), L.dtype_code(d[0].dtype)
                first[i] = total
                total += nb
                steps.append(st["step"])
            # (pinned, asynchronous: a pageable upload here made the host wait for the whole backward before it could queue the step)
            t_dev = L.h2d(torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8), torch.uint8, dev)
            f_dev = L.h2d(torch.from_numpy(first), torch.int64, dev)
            n_live = len(live)
            if ckey is not None:
                self._cache[ckey] = (t_dev, f_dev, n_live, total, steps)
        L.check(lib.sl_adamw_step(t_dev.data_ptr(), f_dev.data_ptr(), n_live, total, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                  float(g["eps"]), float(g["weight_decay"]), t_now, L.stream_ptr()), "sl_adamw_step")
        torch._foreach_add_(steps, 1.0)
        self._keep = getattr(self, "_keep", [])[-15:] + [(t_dev, f_dev)]      # alive until the launches have consumed them
        self.opt._opt_called = True             # what lr_scheduler's order check looks at (it wraps optimizer.step to set it)

    def invalidate(self) -> None:
        self._cache.clear()


# ------------------------------------------------------------------------------------------------
# the KD step
# ------------------------------------------------------------------------------------------------
def effective_accum(train_cfg, world: int):
    """(samples per optimizer step over all ranks, samples per optimizer step on one rank).
    Default (strong scaling, equal to the reference's single-GPU run): `grad_accum_interval` samples per step (ref:config/
    llama3_hubert.yaml:34, ref:trainer.py:373-377) dealt to the ranks, grad_accum_interval / world each.
    `train.per_rank_accum: k` (new key, weak scaling): every rank packs k samples per step, so a step averages k x world samples —
    the per-rank GEMMs keep the row counts of the single-GPU run (k = 16: ~3 200 / 5 072 rows instead of ~400 / 634 at 8 ranks),
    the optimizer sees a `world` times larger batch and the PolynomialLR horizon (total steps = epochs x samples / (k x world),
    ref:trainer.py:106-110) shrinks accordingly; each loss is divided by k x world where the reference divides by
    grad_accum_interval (ref:trainer.py:373)."""
    k = int(getattr(train_cfg, "per_rank_accum", 0) or 0)
    if k > 0:
        return k * world, k
    accum = int(train_cfg.grad_accum_interval)
    if accum % world:
        raise L.SpeechLLMError(f"grad_accum_interval={accum} must be a multiple of the world size {world} (or set train.per_rank_accum)")
    return accum, accum // world


class KDTrainer:
    """ref:trainer.py:23-398 restricted to the optimisation step: losses, backward, accumulation, AdamW + PolynomialLR,
    and (new) data parallelism: rank r runs `grad_accum_interval / world` micro-steps per optimizer step, gradients are
    summed with an all-reduce of the fp32 buckets on a side stream while the rest of the backward still runs."""

    def __init__(self, config, encoder: AudioEncoder, llm: AudioLlamaForCausalLM, prefix_ids: torch.Tensor, suffix_ids: torch.Tensor,
                 total_optimizer_steps: int = 1000, process_group=None, regularizers: Optional[TrainRegularizers] = None,
                 overlap_optimizer: Optional[bool] = None):
        tr = config.train
        self.enc, self.llm = encoder, llm
        self.ntp_w, self.ld_w, self.fd_w = tr.ntp_loss_weight, tr.ld_loss_weight, tr.fd_loss_weight
        self.use_ld, self.use_fd = bool(tr.use_ld_loss), bool(tr.use_fd_loss)
        self.taps = list(tr.fd_loss_connector_layers)
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        self.accum, self.local_accum = effective_accum(tr, self.world)
        dev = encoder.device
        self.prefix_ids, self.suffix_ids = prefix_ids.to(dev), suffix_ids.to(dev)
        self.is_whisper = getattr(encoder, "encoder_base", "hubert") == "whisper"
        self.enc_tape = WhisperEncoderTape(encoder) if self.is_whisper else EncoderTape(encoder)
        self.llm_tape = LlamaTape(llm)
        self.to_state_dict = whisper_grads_to_state_dict if self.is_whisper else kernel_grads_to_state_dict
        # fp32 master weights (the reference keeps fp32 params under fp16 autocast, ref:trainer.py:252,270)
        self.master = {k: v.detach().to(dev, torch.float32).clone() for k, v in encoder.state_dict().items()}
        self.reg = regularizers
        # masked_spec_embed only receives a gradient under SpecAugment (the reference's train() mode); with the regularisers
        # off it stays out of the optimizer, as its gradient is identically zero
        spec = regularizers is not None and regularizers.apply_spec_augment and regularizers.mask_time_prob > 0
        frozen = {"encoder.embed_positions.weight"} if self.is_whisper else (set() if spec else {"encoder.masked_spec_embed"})
        # every encoder parameter, in the order the reference's optimizer indexes them (ref:trainer.py:98-105, first param group =
        # audio_encoder.parameters()); the ones frozen here never get a .grad, which AdamW skips, exactly like the reference's
        # requires_grad=False / never-touched parameters
        from .weights import reference_param_order
        self.param_names = reference_param_order(self.master.keys())
        self.trainable = [k for k in self.param_names if k not in frozen]
        self._param = {k: torch.nn.Parameter(self.master[k], requires_grad=(k not in frozen)) for k in self.param_names}
        self.params = [self._param[k] for k in self.trainable]
        opt = tr.optimizer
        self.optimizer = torch.optim.AdamW([self._param[k] for k in self.param_names], lr=float(opt.lr), betas=(float(opt.beta1), float(opt.beta2)))
        self.scheduler = torch.optim.lr_scheduler.PolynomialLR(self.optimizer, total_iters=total_optimizer_steps, power=1.0)
        self.grads = self.enc_tape.new_grads()
        self.micro = 0                 # micro-steps of the open accumulation window on this rank
        self.micro_total = 0
        self.micro_batches = 0
        self.optimizer_steps = 0
        self.keep_last_grads = False
        from .dist import BucketedAllReduce
        self.reducer = BucketedAllReduce(self.enc_tape.arena, group=process_group) if self.world > 1 else None
        # AdamW + the refresh of the kernels' weight copies as one launch (sl_adamw_step); `use_fused_adamw = False` falls back
        # to torch.optim's foreach step + a full re-derivation of the device weights (what the equivalence test compares with)
        self.merge_teacher_pass = True             # teacher + student sequences through one ragged LLM forward (False: two passes)
        self.use_fused_adamw = True
        self._fused: Optional[FusedAdamW] = None
        self._fused_indirect = None
        # AdamW of the gradient buckets that are already final (and, under data parallel, summed) while the rest of the backward still runs
        # (_early_step).  OFF by default — measured: per-rank window 32.0 vs 31.4 ms, 16-sample window 89.1 vs 89.1 ms
        # (profiles/r06_ac_overlap_optimizer_ab.txt): the step's 1.6 ms of HBM-bound blocks take the CUs from the backward's small kernels for
        # as long as they save at the end.  `train.overlap_optimizer: true` / overlap_optimizer=True turn it on (SL_KD_OVERLAP_OPT=0 overrides).
        self.overlap_optimizer = bool(getattr(tr, "overlap_optimizer", False)) if overlap_optimizer is None else bool(overlap_optimizer)
        self._opt_stream = None
        self._early_plan_ = None
        self._early_names = {}
        self.early_min_bytes = 64 << 20            # a prefix of the arena steps early once this much of it has become final since the last early step
        self.early_launches = 0                    # early AdamW launches so far (tests, bench)
        self._early_reset()
        if self.reducer is not None and self.overlap_optimizer:      # (the default exchange path stays exactly as it was: no hook, no extra stream waits)
            self.reducer.after_bucket = self._early_after_bucket

    # -- the optimizer step of finished gradient buckets, beside the rest of the backward -----------------------------------
    # (Opt-in, see __init__.)  AdamW is element-wise: a parameter's update needs its own (final) gradient only.  The arena is laid out in the order the backward
    # finishes its buffers (projector, layer N-1 .. 0, ...), so once a prefix of it is final its parameters can step — 1.6 ms of HBM-bound
    # work per window (30 B per parameter) that otherwise sits between the last backward kernel and the next forward.  Bit-identical to the
    # step behind the backward (same kernel, same hyper-parameters; the scheduler moves after the whole step).  Only parameters whose
    # state-dict gradient IS a slice of the arena take part (the transformer layers, the projector: 95 % of HuBERT-large); the re-laid-out
    # ones (conv taps, weight-norm) step at the end as before.
    def _early_reset(self) -> None:
        self._early_armed = False
        self._early_next = 0
        self._early_done = set()
        self._early_final = None
        self._early_n_final = 0
        self._early_upto = 0

    def _early_plan(self):
        """[(name, arena offset, elements)] by offset, of the parameters whose gradient is a contiguous slice of the arena."""
        if self._early_plan_ is None:
            flat = self.enc_tape.arena.flat
            sd = self.to_state_dict(self.enc, self.grads, self.master)
            base, store = flat.data_ptr(), flat.untyped_storage().data_ptr()
            plan = []
            for k in self.trainable:
                g = sd[k]
                if g.is_contiguous() and g.dtype == torch.float32 and g.untyped_storage().data_ptr() == store and g.numel() == self._param[k].numel():
                    plan.append((k, (g.data_ptr() - base) // 4, g.numel()))
            self._early_plan_ = sorted(plan, key=lambda r: r[1])
        return self._early_plan_

    def _early_step(self, upto: int, after_stream=None) -> None:
        """AdamW, on the optimizer's own stream, of every planned parameter not yet stepped whose gradient ends at or below arena offset `upto`;
        ordered behind `after_stream` (default: the current stream) at the time of the call."""
        fused = self._fused_optimizer() if self.use_fused_adamw else None
        if fused is None:
            return
        plan, i0 = self._early_plan(), self._early_next
        while self._early_next < len(plan) and plan[self._early_next][1] + plan[self._early_next][2] <= upto:
            self._early_next += 1
        i1 = self._early_next
        if i1 == i0:
            return
        names = self._early_names.get((i0, i1))          # the same chunks every window: their record tables are built once (FusedAdamW.step)
        if names is None:
            names = self._early_names[(i0, i1)] = tuple(sorted(k for k, _, _ in plan[i0:i1]))
        flat = self.enc_tape.arena.flat
        if names not in fused._cache:
            for k, off, n in plan[i0:i1]:
                p = self._param[k]
                p.grad = flat[off:off + n].view(p.shape)
        if self._opt_stream is None:
            self._opt_stream = self._make_opt_stream(flat.device)
        ev = torch.cuda.Event()
        ev.record(after_stream if after_stream is not None else torch.cuda.current_stream())
        self._opt_stream.wait_event(ev)
        with torch.cuda.stream(self._opt_stream):
            fused.step(only=names)
        self._early_done.update(names)
        self.early_launches += 1

    @staticmethod
    def _make_opt_stream(device):
        """The early steps' stream.  SL_KD_OPT_CUS=n (default 0: no mask) confines it to n CUs of every XCD (hipExtStreamCreateWithCUMask): the
        step's HBM-bound blocks then trickle beside the backward instead of taking every CU from its small kernels."""
        n = int(os.environ.get("SL_KD_OPT_CUS", "0") or 0)
        if n <= 0 or n >= 32:
            return torch.cuda.Stream(device=device)
        hip = C.CDLL("libamdhip64.so")
        words = (C.c_uint32 * 8)(*([(1 << n) - 1] * 8))           # bit 32 x + c = CU c of XCD x
        h = C.c_void_p()
        with torch.cuda.device(device):
            rc = hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words)
        if rc != 0:
            raise L.SpeechLLMError(f"hipExtStreamCreateWithCUMask failed ({rc})")
        return torch.cuda.ExternalStream(h.value, device=device)

    def _early_ready(self, names) -> None:
        """on_bucket callback of the backward without a reducer: the same final-prefix bookkeeping as BucketedAllReduce.ready."""
        arena = self.enc_tape.arena
        if self._early_final is None:
            self._early_final = [False] * len(arena.order)
        for n in names:
            self._early_final[arena.index[n]] = True
        while self._early_n_final < len(self._early_final) and self._early_final[self._early_n_final]:
            self._early_n_final += 1
        upto = arena.end_offset(self._early_n_final - 1) if self._early_n_final > 0 else 0
        if (upto - self._early_upto) * 4 >= self.early_min_bytes:
            self._early_upto = upto
            self._early_step(upto)

    def _early_after_bucket(self, upto: int) -> None:
        """BucketedAllReduce.after_bucket: the arena below `upto` has been summed over the ranks on the reducer's stream."""
        if self._early_armed and self.reducer is not None and self.reducer.stream is not None:
            self._early_step(upto, after_stream=self.reducer.stream)

    def _fused_optimizer(self) -> Optional["FusedAdamW"]:
        if self._fused is None:
            W = self.enc.weights
            direct, indirect = direct_refresh_map(self.enc, self.to_state_dict, self.grads, self.master)
            handled = set(W.indirect_roles())
            # roles without a device tensor (masked_spec_embed) have nothing to refresh; roles the weight class declares constant
            # (Whisper: the frozen position table, the k slot of the fused qkv bias — k_proj has no bias) never change
            const = set(W.constant_roles())
            need = {r for r in indirect if W.role_tensor(r) is not None and r not in const}
            if not need <= handled:
                self.use_fused_adamw = False            # a layout this shortcut does not know: keep the general path
                return None
            dst_of = {}
            for key, (role, off) in direct.items():
                t = W.role_tensor(role)
                if t is not None:
                    dst_of[key] = (t, off)
            self._fused = FusedAdamW(self.optimizer, [(k, self._param[k]) for k in self.param_names], dst_of)
            self._fused_indirect = sorted(need)
            self.enc._state = self.master               # state_dict() reads the masters themselves (updated in place)
        return self._fused

    # -- micro-steps ----------------------------------------------------------------------------
    def micro_step(self, wave: torch.Tensor, text_ids: torch.Tensor, response_ids: torch.Tensor) -> Dict[str, float]:
        """One utterance (the reference's loop body, ref:trainer.py:261-384)."""
        return self.micro_batch([wave], [text_ids], [response_ids])[0]

    def micro_batch(self, waves: Sequence[torch.Tensor], text_ids: Sequence[torch.Tensor], response_ids: Sequence[torch.Tensor],
                    close_window: Optional[bool] = None) -> List[Dict[str, float]]:
        """len(waves) micro-steps of ONE accumulation window processed together as a packed, ragged batch.  The encoder
        weights do not change inside a window, so this is the same arithmetic as the reference's sequential batch-size-1
        micro-steps (each loss still divided by grad_accum_interval, gradients summed) — but every GEMM sees all the
        window's tokens at once instead of ~200 rows.  ids are 1-D with the BOS already stripped (ref:trainer.py:155-156).
        Returns per-utterance losses (ref:trainer.py:325-370)."""
        enc, llm = self.enc, self.llm
        dev, dt = enc.device, enc.dtype
        emb = llm.model.embed_tokens
        B = len(waves)
        if self.micro + B > self.local_accum:
            raise L.SpeechLLMError("a micro-batch must not straddle an optimizer step")
        # close_window=True: the epoch's last, partial accumulation window (the reference steps on the last batch too,
        # ref:trainer.py:377); every rank then exchanges the whole arena in one piece (ranks may hold different sample counts,
        # so the early per-bucket launches, whose sequence must be identical on all ranks, are skipped)
        last = (self.micro + B) == self.local_accum if close_window is None else bool(close_window)
        early_buckets = last and close_window is None and self.reducer is not None
        response_ids = [r if r.is_cuda else L.h2d(r, r.dtype, dev) for r in response_ids]      # (asynchronous uploads: nothing in a window may stall the stream)
        text_ids = [t_ if t_.is_cuda else L.h2d(t_, t_.dtype, dev) for t_ in text_ids]
        ns = [int(r.shape[0]) for r in response_ids]
        audio, etape = self.enc_tape.forward(waves, self.reg, step=self.micro_batches,
                                             masked_spec_embed=self.master.get("encoder.masked_spec_embed"))   # (sum P, H) packed
        self.micro_batches += 1
        poff = etape["poff"]
        pre, suf = emb(self.prefix_ids)[0], emb(self.suffix_ids)[0, 1:]
        n_pre = pre.shape[0]
        resp = [emb(r[None])[0, 1:] for r in response_ids]
        a_parts, a_lens = [], []
        for u in range(B):                                                            # ref:utils.py:36-45
            a_parts += [pre, audio[poff[u]:poff[u + 1]], suf, resp[u]]
            a_lens.append(n_pre + (poff[u + 1] - poff[u]) + suf.shape[0] + resp[u].shape[0])
        a_seq = torch.cat(a_parts, 0).contiguous()
        aoff = _offsets(a_lens)
        # The teacher (text prompt, no gradient) and the student (audio prompt) run through the SAME frozen weights: when the
        # teacher is needed at all, both go through ONE ragged forward pass, student sequences first (M ~ 5 000 rows per GEMM
        # instead of 3 200 + 1 900; ref:trainer.py:299-323 runs them as two calls).  The backward visits the student rows only.
        need_teacher = self.use_ld or self.use_fd
        merged = need_teacher and self.merge_teacher_pass
        t_seq = t_lens = None
        if need_teacher:
            t_parts, t_lens = [], []
            for u in range(B):
                te = emb(text_ids[u][None])[0]
                t_parts += [pre, te, suf, resp[u]]
                t_lens.append(n_pre + te.shape[0] + suf.shape[0] + resp[u].shape[0])
            t_seq = torch.cat(t_parts, 0).contiguous()
        n_a = a_seq.shape[0]
        if merged:
            hidden_all, ltape = self.llm_tape.forward(torch.cat([a_seq, t_seq], 0), a_lens + t_lens)
            hidden_a = [h[:n_a] for h in hidden_all]
            hidden_t = [h[n_a:] for h in hidden_all]
        else:
            hidden_a, ltape = self.llm_tape.forward(a_seq, a_lens)
            hidden_t = self.llm_tape.forward(t_seq, t_lens, save=False)[0] if need_teacher else None   # teacher pass: same kernels, nothing kept
        # rows whose logits / hidden states the losses read: the last n_u rows of every sequence
        tail = L.h2d(torch.cat([torch.arange(aoff[u + 1] - ns[u], aoff[u + 1]) for u in range(B)]), torch.int64, dev)
        toffs = _offsets(ns)
        logits_a = self.llm_tape.logits(hidden_a[-1].index_select(0, tail))           # (sum n, V) fp32
        V = logits_a.shape[1]
        losses = torch.zeros((B, 3), device=dev, dtype=torch.float32)
        d_logits = torch.empty((toffs[B], V), device=dev, dtype=dt)
        inv_acc = 1.0 / self.accum                                                    # ref:trainer.py:373
        # per-row descriptors of the window's loss launches: label (the last row of an utterance predicts nothing:
        # logits[-n:-1] vs labels[1:], ref:model/audio_llama.py:84-89), weights of the loss values / gradients, utterance slot
        lab_rows, coef_rows, slot_rows, mse_rows = [], [], [], []
        H_llm = a_seq.shape[1]
        for u in range(B):
            n = ns[u]
            lab_rows.append(torch.cat([response_ids[u][1:].to(torch.int32), torch.full((1,), -1, dtype=torch.int32, device=dev)]))
            ld_on = 1.0 if self.use_ld else 0.0
            # a one-token response has no next-token target (its only row carries label -1): the reference's mean over zero
            # targets is NaN (ref:model/audio_llama.py:84-91) and poisons every weight; here the term contributes nothing — the
            # same on every rank, so a data-parallel window never stalls on it
            inv_n1 = 1.0 / (n - 1) if n > 1 else 0.0
            coef_rows += [[inv_n1, self.ntp_w * inv_acc * inv_n1, ld_on / n, ld_on * self.ld_w * inv_acc / n]] * n
            slot_rows += [u] * n
            mse_rows += [[1.0 / (n * H_llm), 2.0 * self.fd_w * inv_acc / (n * H_llm)]] * n
        labels = torch.cat(lab_rows).contiguous()
        row_coef = L.h2d(coef_rows, torch.float32, dev)
        row_slot = L.h2d(slot_rows, torch.int32, dev)
        d_hidden: Dict[int, torch.Tensor] = {}
        logits_t = None
        if need_teacher:
            tto = _offsets(t_lens)
            ttail = L.h2d(torch.cat([torch.arange(tto[u + 1] - ns[u], tto[u + 1]) for u in range(B)]), torch.int64, dev)
            if self.use_ld:
                logits_t = self.llm_tape.logits(hidden_t[-1].index_select(0, ttail))
            if self.use_fd:
                mse_coef = L.h2d(mse_rows, torch.float32, dev)
                for l in self.taps:                                                   # one launch per tap for the whole window
                    ha, ht = hidden_a[l].index_select(0, tail), hidden_t[l].index_select(0, ttail)
                    dtail = torch.empty_like(ha)
                    ops.kd_mse_rows(ha, ht, mse_coef, row_slot, losses, 2, dtail)
                    d = torch.zeros_like(hidden_a[l])
                    d.index_copy_(0, tail, dtail)
                    d_hidden[l] = d
        # next-token CE + soft CE of every utterance: one launch, the gradient written once
        ops.kd_logit_losses(logits_a, logits_t, labels, row_coef, row_slot, losses, d_logits, dt)
        d_seq = self.llm_tape.backward(ltape, tail, d_logits, d_hidden, n_seq=B if merged else None)
        d_audio = torch.cat([d_seq[aoff[u] + n_pre: aoff[u] + n_pre + (poff[u + 1] - poff[u])] for u in range(B)], 0).contiguous()
        # (an instance whose optimizer_step was replaced — gradient inspection in the tests — never steps early)
        sw = os.environ.get("SL_KD_OVERLAP_OPT", "")                  # "1" / "0" override the trainer's setting (A/B runs)
        self._early_armed = ((sw == "1" or (self.overlap_optimizer and sw != "0")) and last and close_window is None and self.use_fused_adamw and
                             d_audio.is_cuda and "optimizer_step" not in self.__dict__)
        on_bucket = self.reducer.ready if early_buckets else (self._early_ready if self._early_armed and self.reducer is None else None)
        self.enc_tape.backward(etape, d_audio, self.grads, on_bucket=on_bucket)
        self.micro += B
        self.micro_total += B
        self.last_d_audio = d_audio
        # The losses go to the host through a pinned asynchronous copy queued BEFORE the optimizer step's launches, and are waited for only after
        # those launches are queued: reading them first (losses.tolist()) left the GPU idle for the 1.4 ms the host needs to prepare the step
        # (profiles/r06_n_kd_timeline.txt: the largest gap of a window), and a read behind the step would wait for AdamW as well.
        ev = None
        if losses.is_cuda:
            host = torch.empty(losses.shape, dtype=losses.dtype, pin_memory=True)
            host.copy_(losses, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        else:
            host = losses
        if last:
            self.optimizer_step()
        if ev is not None:
            ev.synchronize()
        out = []
        for ntp, ld, fd in host.tolist():
            out.append(dict(ntp_loss=ntp, ld_loss=ld, fd_loss=fd, total=self.ntp_w * ntp + self.ld_w * ld + self.fd_w * fd))
        return out

    def close(self) -> None:
        """Tear the data-parallel reducer's communicator down (a collective over the ranks; BucketedAllReduce never does it from
        __del__).  Idempotent; the step must not be used for exchanges afterwards."""
        if self.reducer is not None:
            self.reducer.close()

    def abort(self) -> None:
        """Failure path: drop the reducer's communicator locally (ncclCommAbort), never waiting for the peers (BucketedAllReduce.abort)."""
        if self.reducer is not None:
            self.reducer.abort()

    def close_window(self) -> None:
        """Optimizer step of a window in which THIS rank held no sample (the tail of an epoch under data parallel): its
        gradients are zero, the all-reduce still has to be joined."""
        self.optimizer_step()

    # -- optimizer step ---------------------------------------------------------------------------
    def optimizer_step(self) -> None:
        if self.reducer is not None:
            self.reducer.finish()                  # in place on the arena: nothing to scatter back
        early = self._early_done
        if early:                                  # their AdamW ran beside the backward (_early_step): this stream owns weights and gradients again
            torch.cuda.current_stream().wait_stream(self._opt_stream)
        sd_grads = self.to_state_dict(self.enc, self.grads, self.master)
        if self.keep_last_grads:                   # tests / debugging only: a 1.27 GB clone per step otherwise
            self.last_grads = {k: sd_grads[k].reshape(p.shape).clone() for k, p in zip(self.trainable, self.params)}
        for k, p in zip(self.trainable, self.params):
            p.grad = None if k in early else sd_grads[k].reshape(p.shape)  # views of the arena wherever kernel and state-dict layouts coincide
        fused = self._fused_optimizer() if self.use_fused_adamw else None
        if fused is not None:
            fused.step()                           # p, m, v and the kernels' copy of p: one pass over 30 B per parameter
        else:
            self.optimizer.step()
        self.scheduler.step()
        self.optimizer.zero_grad(set_to_none=True)
        self.enc_tape.arena.zero_()
        if fused is not None:
            self.enc.weights.refresh_indirect(self.master)      # the few re-laid-out tensors (conv taps, folded weight norm)
        else:
            self.enc.refresh_weights(self.master)  # compute-dtype kernel weights follow the fp32 master, in place
        self.optimizer_steps += 1
        self.micro = 0
        self._early_reset()

    # -- optimizer state in the reference's checkpoint layout ------------------------------------------
    def _n_llm_params(self) -> int:
        a = self.llm.arch            # LlamaForCausalLM.parameters(): embed, 9 per layer, final norm (+ lm_head when untied)
        return 2 + 9 * a.num_hidden_layers + (0 if a.tie_word_embeddings else 1)

    def optimizer_state_dict(self) -> dict:
        """torch.optim.AdamW.state_dict() as the REFERENCE's optimizer would write it (ref:trainer.py:98-105, 516-528): two
        param groups — every encoder parameter in `audio_encoder.parameters()` order, then the (frozen, stateless) LLM's."""
        sd = self.optimizer.state_dict()
        g0 = dict(sd["param_groups"][0])
        n_enc = len(self.param_names)
        g0["params"] = list(range(n_enc))
        g1 = dict(g0)
        g1["params"] = list(range(n_enc, n_enc + self._n_llm_params()))
        return {"state": sd["state"], "param_groups": [g0, g1]}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        """Accepts the reference's two-group layout (or this build's): per-parameter state is taken by index for the encoder
        group; the LLM group carries no state (its parameters never had gradients)."""
        groups = sd["param_groups"]
        n_enc = len(self.param_names)
        if len(groups[0]["params"]) != n_enc:
            raise L.SpeechLLMError(f"optimizer checkpoint has {len(groups[0]['params'])} encoder parameters, this encoder has {n_enc}")
        g0 = dict(groups[0])
        g0["params"] = list(range(n_enc))
        state = {int(i): v for i, v in sd["state"].items() if int(i) < n_enc}
        self.optimizer.load_state_dict({"state": state, "param_groups": [g0]})
        if self._fused is not None:
            self._fused.invalidate()              # the moments were replaced: cached record tables point at the old ones
        dev = self.enc.device
        for st in self.optimizer.state.values():          # ref:trainer.py:124-130 moves the state to the GPU by hand
            for k, v in st.items():
                if torch.is_tensor(v) and k != "step":
                    st[k] = v.to(dev)
