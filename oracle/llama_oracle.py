"""Oracle (test infrastructure): Llama decoder, greedy decode, response-only loss — fp32 CPU.

Restates, op for op:
  * ref:model/audio_llama.py:22-113 (AudioLlamaForCausalLM.forward incl. per-sample response-only CE)
  * hf:models/llama/modeling_llama.py:52-67 (RMSNorm), :73-127 (rotary), :130-160 (rotate_half /
    apply_rotary_pos_emb), :163-176 (SwiGLU MLP), :179-213 (repeat_kv + eager attention),
    :217-281 (attention), :284-324 (decoder layer), :347-417 (LlamaModel.forward)
  * hf:modeling_rope_utils.py:580-662 (llama3 scaled inverse frequencies)
  * hf:generation/utils.py:2783-2972 (_sample in greedy mode: argmax of fp32 last-row logits, EOS
    stop, pad finished rows) as invoked at ref:inference.py:60-66 with `inputs_embeds` only
    (ids start empty: hf:generation/utils.py:736-744).

Weights are in the HF LlamaForCausalLM state-dict layout (`model.*`, `lm_head.weight`; the head is
tied to the embedding when `lm_head.weight` is absent).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


@dataclass
class LlamaCfg:
    hidden_size: int = 3072
    num_hidden_layers: int = 28
    num_attention_heads: int = 24
    num_key_value_heads: int = 8
    head_dim: int = 128
    intermediate_size: int = 8192
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Optional[dict] = None  # {"factor","low_freq_factor","high_freq_factor","original_max_position_embeddings"}
    tie_word_embeddings: bool = True
    eos_token_ids: Tuple[int, ...] = (128001, 128008, 128009)
    pad_token_id: Optional[int] = None  # HF: defaults to eos_token_ids[0] when unset


LLAMA32_3B = LlamaCfg(
    rope_scaling=dict(factor=32.0, low_freq_factor=1.0, high_freq_factor=4.0,
                      original_max_position_embeddings=8192))

# GeneZC/MiniChat-2-3B (ref:config/minichat_hubert.yaml:22): Llama arch, MHA, untied head.
MINICHAT2_3B = LlamaCfg(hidden_size=3072, num_hidden_layers=24, num_attention_heads=24,
                        num_key_value_heads=24, head_dim=128, intermediate_size=8192, vocab_size=49216,
                        rms_norm_eps=1e-5, rope_theta=10000.0, rope_scaling=None,
                        tie_word_embeddings=False, eos_token_ids=(2,), pad_token_id=None)


def rope_inv_freq(cfg: LlamaCfg) -> torch.Tensor:
    """Default and `llama3` inverse frequencies (hf:models/llama/modeling_llama.py:103-108,
    hf:modeling_rope_utils.py:636-660)."""
    dim = cfg.head_dim
    inv_freq = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(torch.float) / dim))
    if cfg.rope_scaling is None:
        return inv_freq
    factor = cfg.rope_scaling["factor"]
    low = cfg.rope_scaling["low_freq_factor"]
    high = cfg.rope_scaling["high_freq_factor"]
    old_len = cfg.rope_scaling["original_max_position_embeddings"]
    low_wl = old_len / low
    high_wl = old_len / high
    wavelen = 2 * math.pi / inv_freq
    inv_l = torch.where(wavelen > low_wl, inv_freq / factor, inv_freq)
    smooth = (old_len / wavelen - low) / (high - low)
    smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
    medium = ~(wavelen < high_wl) * ~(wavelen > low_wl)
    return torch.where(medium, smoothed, inv_l)


def rope_cos_sin(cfg: LlamaCfg, position_ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """hf:models/llama/modeling_llama.py:110-127: fp32 cos/sin of shape (..., head_dim)."""
    inv = rope_inv_freq(cfg)
    freqs = position_ids.float()[..., None] * inv  # == (inv_freq @ pos).T, single products, exact
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def rms_norm(x: torch.Tensor, w: torch.Tensor, eps: float) -> torch.Tensor:
    xf = x.float()
    var = xf.pow(2).mean(-1, keepdim=True)
    return w.float() * (xf * torch.rsqrt(var + eps))


def llama_forward(
    sd: Dict[str, torch.Tensor],
    cfg: LlamaCfg,
    inputs_embeds: torch.Tensor,
    attention_mask: Optional[torch.Tensor] = None,
    past: Optional[List[Tuple[torch.Tensor, torch.Tensor]]] = None,
    output_hidden_states: bool = False,
    last_logits_only: bool = False,
    logits_last_n: Optional[int] = None,
):
    """LlamaModel.forward + lm_head.
    `logits_last_n`: lm_head on the last n positions only (the rows ref:model/audio_llama.py:84-89 / ref:trainer.py:325-340
    read; the other rows of HF's all-position logits never reach a loss) — keeps a 128 256-way vocabulary affordable on CPU.

    inputs_embeds (B,S,h); attention_mask (B, past+S) 0/1 with LEFT padding or None.
    Position ids are arange(S)+past_len irrespective of padding (hf:...llama.py:386-389).
    Returns dict(logits, hidden_states (tuple of L+1, last one post-norm), past).
    """
    B, S, H = inputs_embeds.shape
    nh, nkv, hd = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    past_len = 0 if past is None else past[0][0].shape[2]
    pos = torch.arange(past_len, past_len + S)[None, :].expand(B, S)
    cos, sin = rope_cos_sin(cfg, pos)  # (B,S,hd)
    cos = cos[:, None]
    sin = sin[:, None]

    total = past_len + S
    # causal ∧ padding additive mask (hf:masking_utils.py; eager path adds it to the scores)
    qi = torch.arange(past_len, total)[:, None]
    kj = torch.arange(total)[None, :]
    allowed = (kj <= qi)[None, None].expand(B, 1, S, total)
    if attention_mask is not None:
        allowed = allowed & attention_mask.bool()[:, None, None, :]
    bias = torch.zeros(B, 1, S, total).masked_fill(~allowed, float("-inf"))

    x = inputs_embeds.float()
    hiddens = []
    new_past = []
    for li in range(cfg.num_hidden_layers):
        q = f"model.layers.{li}."
        if output_hidden_states:
            hiddens.append(x)
        res = x
        h = rms_norm(x, sd[q + "input_layernorm.weight"], cfg.rms_norm_eps)
        a = q + "self_attn."
        qs = F.linear(h, sd[a + "q_proj.weight"].float()).view(B, S, nh, hd).transpose(1, 2)
        ks = F.linear(h, sd[a + "k_proj.weight"].float()).view(B, S, nkv, hd).transpose(1, 2)
        vs = F.linear(h, sd[a + "v_proj.weight"].float()).view(B, S, nkv, hd).transpose(1, 2)
        qs = qs * cos + rotate_half(qs) * sin
        ks = ks * cos + rotate_half(ks) * sin
        if past is not None:
            ks = torch.cat([past[li][0], ks], dim=2)
            vs = torch.cat([past[li][1], vs], dim=2)
        new_past.append((ks, vs))
        rep = nh // nkv
        kr = ks[:, :, None].expand(B, nkv, rep, total, hd).reshape(B, nh, total, hd)
        vr = vs[:, :, None].expand(B, nkv, rep, total, hd).reshape(B, nh, total, hd)
        att = torch.matmul(qs, kr.transpose(2, 3)) * (hd ** -0.5) + bias
        att = F.softmax(att, dim=-1, dtype=torch.float32)
        # fully masked rows (left-pad query positions) are NaN in eager HF as well; zero them so the
        # padded positions stay finite (their outputs are never read: ref:model/audio_llama.py:84)
        att = torch.nan_to_num(att, nan=0.0)
        o = torch.matmul(att, vr).transpose(1, 2).reshape(B, S, nh * hd)
        x = res + F.linear(o, sd[a + "o_proj.weight"].float())
        res = x
        h = rms_norm(x, sd[q + "post_attention_layernorm.weight"], cfg.rms_norm_eps)
        m = q + "mlp."
        g = F.linear(h, sd[m + "gate_proj.weight"].float())
        u = F.linear(h, sd[m + "up_proj.weight"].float())
        x = res + F.linear(F.silu(g) * u, sd[m + "down_proj.weight"].float())
    x = rms_norm(x, sd["model.norm.weight"], cfg.rms_norm_eps)
    if output_hidden_states:
        hiddens.append(x)
    head = sd["lm_head.weight"] if "lm_head.weight" in sd else sd["model.embed_tokens.weight"]
    rows = x[:, -1:] if last_logits_only else (x[:, -int(logits_last_n):] if logits_last_n else x)
    logits = F.linear(rows, head.float())
    return dict(logits=logits, hidden_states=tuple(hiddens), past=new_past, last_hidden=x)


def response_only_loss(logits: torch.Tensor, labels: Sequence[torch.Tensor]) -> torch.Tensor:
    """ref:model/audio_llama.py:72-101: per-sample mean CE of logits[-n:-1] vs labels[1:], batch mean."""
    loss = 0.0
    for sample_logits, sample_labels in zip(logits, labels):
        n = sample_labels.shape[0]
        shift_logits = sample_logits[-n:-1, :]
        shift_labels = sample_labels[1:]
        loss = loss + F.cross_entropy(shift_logits.float(), shift_labels.long())
    return loss / logits.shape[0]


def greedy_generate(
    sd: Dict[str, torch.Tensor],
    cfg: LlamaCfg,
    inputs_embeds: torch.Tensor,
    max_new_tokens: int,
    use_eos: bool = True,
    return_margins: bool = False,
):
    """Greedy KV-cached decode from prompt embeddings (B,S,h), no padding.

    Returns LongTensor (B, n_new) holding ONLY new tokens (the prompt had no ids), padded with
    pad_token_id after a row's EOS, truncated at the step where every row has finished —
    hf:generation/utils.py:2876-2942.
    """
    B = inputs_embeds.shape[0]
    embed = sd["model.embed_tokens.weight"].float()
    eos = torch.tensor(cfg.eos_token_ids)
    pad = cfg.pad_token_id if cfg.pad_token_id is not None else cfg.eos_token_ids[0]
    out = llama_forward(sd, cfg, inputs_embeds, last_logits_only=True)
    past = out["past"]
    unfinished = torch.ones(B, dtype=torch.long)
    ids = []
    margins = []
    for step in range(max_new_tokens):
        logits = out["logits"][:, -1].float()
        nxt = logits.argmax(dim=-1)
        if return_margins:
            top2 = logits.topk(2, dim=-1).values
            margins.append(top2[:, 0] - top2[:, 1])
        if use_eos:
            nxt = nxt * unfinished + pad * (1 - unfinished)
        ids.append(nxt)
        if use_eos:
            unfinished = unfinished & ~torch.isin(nxt, eos)
            if unfinished.max() == 0:
                break
        if step + 1 < max_new_tokens:
            out = llama_forward(sd, cfg, embed[nxt][:, None, :], past=past, last_logits_only=True)
            past = out["past"]
    ids = torch.stack(ids, dim=1)
    if return_margins:
        return ids, torch.stack(margins, dim=1)
    return ids
