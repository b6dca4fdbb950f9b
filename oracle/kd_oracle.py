"""Oracle (test infrastructure): prompt/sequence assembly and knowledge-distillation losses, fp32 CPU.

Restates:
  * ref:utils.py:27-46  merge_prompt_response_tokens  [prefix | x | suffix[1:] | response[1:]]
  * ref:utils.py:49-73  merge_prompt_tokens           [prefix | x | suffix[1:]]
  * ref:utils.py:76-82  construct_attention_mask (left padding)
  * ref:utils.py:85-164 batch_full_embed_sequence
  * ref:utils.py:167-178 soft_cross_entropy
  * ref:trainer.py:325-370 loss mix (ntp / ld / fd on connector layers)
  * ref:inference.py:95-137 order of operations of generate_audio_response

Token ids are taken as given (the tokenizer is host-side text processing).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from .llama_oracle import LlamaCfg, llama_forward, response_only_loss, greedy_generate


def embed(sd, ids: torch.Tensor) -> torch.Tensor:
    return sd["model.embed_tokens.weight"].float()[ids.long()]


def merge_prompt_tokens(sd, prefix_ids: torch.Tensor, suffix_ids: torch.Tensor, inputs_embeds: torch.Tensor):
    """ref:utils.py:63-73; prefix_ids/suffix_ids are (1,n) including the BOS the tokenizer prepends."""
    return torch.cat([embed(sd, prefix_ids), inputs_embeds, embed(sd, suffix_ids)[:, 1:, :]], dim=1)


def merge_prompt_response_tokens(sd, prefix_ids, suffix_ids, inputs_embeds, response_ids):
    """ref:utils.py:33-46; response_ids (1,n): its first id is dropped again here."""
    return torch.cat([embed(sd, prefix_ids), inputs_embeds, embed(sd, suffix_ids)[:, 1:, :],
                      embed(sd, response_ids)[:, 1:, :]], dim=1)


def construct_attention_mask(seq_lens: Sequence[int]) -> torch.Tensor:
    max_len = max(seq_lens)
    return torch.stack([F.pad(torch.ones(n), (max_len - n, 0)) for n in seq_lens]).long()


def left_pad_batch(seqs: List[torch.Tensor]):
    lens = [s.shape[1] for s in seqs]
    m = max(lens)
    padded = torch.cat([F.pad(s, (0, 0, m - s.shape[1], 0)) for s in seqs])
    return padded, construct_attention_mask(lens)


def soft_cross_entropy(inp: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    return (-(F.softmax(target.float(), dim=-1) * F.log_softmax(inp.float(), dim=-1)).sum(-1)).mean()


def generate_audio_response_ids(sd_llm, cfg: LlamaCfg, audio_embeds, prefix_ids, suffix_ids,
                                additional_text_ids: Optional[torch.Tensor] = None,
                                max_new_tokens: int = 256, use_eos: bool = True, return_margins=False):
    """ref:inference.py:109-135: [prefix | (text[1:]) | audio | suffix[1:]] -> greedy ids."""
    combined = audio_embeds
    if additional_text_ids is not None and additional_text_ids.numel() > 0:
        combined = torch.cat([embed(sd_llm, additional_text_ids[:, 1:]), audio_embeds], dim=1)
    prompt = merge_prompt_tokens(sd_llm, prefix_ids, suffix_ids, combined)
    return greedy_generate(sd_llm, cfg, prompt, max_new_tokens, use_eos=use_eos, return_margins=return_margins)


def kd_losses(sd_llm, cfg: LlamaCfg, audio_embeds, text_ids, response_ids, prefix_ids, suffix_ids,
              connector_layers=(0, 5, 11, 17, 23), ntp_w=0.5, ld_w=0.5, fd_w=1.0, tail_logits_only=False):
    """One KD micro-step's losses for batch size 1 (ref:trainer.py:299-370).

    audio_embeds (1,P,h) may carry grad; text_ids / response_ids are 1-D id tensors already stripped of
    BOS by the collate (ref:trainer.py:155-156).
    """
    a_seq = merge_prompt_response_tokens(sd_llm, prefix_ids, suffix_ids, audio_embeds, response_ids[None])
    t_seq = merge_prompt_response_tokens(sd_llm, prefix_ids, suffix_ids, embed(sd_llm, text_ids[None]),
                                         response_ids[None])
    n = response_ids.shape[0]
    last_n = n if tail_logits_only else None      # every loss below reads logits[-n:] only
    a = llama_forward(sd_llm, cfg, a_seq, attention_mask=torch.ones(1, a_seq.shape[1], dtype=torch.long),
                      output_hidden_states=True, logits_last_n=last_n)
    ntp = response_only_loss(a["logits"], [response_ids])
    with torch.no_grad():
        t = llama_forward(sd_llm, cfg, t_seq, attention_mask=torch.ones(1, t_seq.shape[1], dtype=torch.long),
                          output_hidden_states=True, logits_last_n=last_n)
    ld = soft_cross_entropy(a["logits"][:, -n:, :], t["logits"][:, -n:, :])
    fd = 0.0
    for li in connector_layers:
        fd = fd + F.mse_loss(a["hidden_states"][li][:, -n:, :], t["hidden_states"][li][:, -n:, :])
    total = ntp_w * ntp + ld_w * ld + fd_w * fd
    return dict(ntp=ntp, ld=ld, fd=fd, total=total)
