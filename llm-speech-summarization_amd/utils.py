"""Host-side sequence assembly: mirror of the tensor parts of ref:utils.py.

Same names, argument meaning and error behaviour as the reference; embeddings come from the HIP row
gather behind `embed_tokens`, concatenation stays on the device.  Token ids come from the caller's
tokenizer object exactly as in the reference (`tokenizer(text, return_tensors="pt").input_ids`).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

SYSTEM_PROMPT = ""
MINICHAT_PROMPT_PREFIX = f"{SYSTEM_PROMPT}[|User|]"
MINICHAT_PROMPT_SUFFIX = "</s>[|Assistant|]"
LLAMA_PROMPT_PREFIX = (f"<|start_header_id|>system<|end_header_id|>{SYSTEM_PROMPT}<|eot_id|>"
                       "<|start_header_id|>user<|end_header_id|>\n\n")
LLAMA_PROMPT_SUFFIX = "<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n"

LLAMA_ID = "meta-llama/Llama-3.2-3B-Instruct"
MINICHAT_ID = "GeneZC/MiniChat-2-3B"


def prompt_template(llm_type: str):
    """Template selection by string equality, as ref:utils.py:50-57 (local checkpoint directories are
    accepted when their basename matches the hub id's)."""
    base = llm_type.rstrip("/").split("/")[-1]
    if llm_type == MINICHAT_ID or base == MINICHAT_ID.split("/")[-1]:
        return MINICHAT_PROMPT_PREFIX, MINICHAT_PROMPT_SUFFIX
    if llm_type == LLAMA_ID or base == LLAMA_ID.split("/")[-1]:
        return LLAMA_PROMPT_PREFIX, LLAMA_PROMPT_SUFFIX
    raise Exception("Unknown LLM type.")


def compute_num_audio_embeds(audio_samples, sr=16000):
    """ref:utils.py:13-24, reproduced bit for bit (it defines the crop; may be one short)."""
    num_embeds = (audio_samples - (sr * 0.01)) // (sr * 0.02)
    return int(num_embeds // 4 - 1)


def merge_prompt_response_tokens(prefix_input_ids, suffix_input_ids, inputs_embeds, response_input_ids, embed_tokens):
    """[prefix | x | suffix[1:] | response[1:]]  (ref:utils.py:27-46)."""
    return torch.cat([embed_tokens(prefix_input_ids), inputs_embeds, embed_tokens(suffix_input_ids)[:, 1:, :],
                      embed_tokens(response_input_ids)[:, 1:, :]], dim=1)


def merge_prompt_tokens(inputs_embeds, tokenizer, embed_tokens, llm_type, device):
    """[prefix | x | suffix[1:]]  (ref:utils.py:49-73)."""
    prefix, suffix = prompt_template(llm_type)
    prefix_input_ids = tokenizer(prefix, return_tensors="pt").input_ids.to(device)
    suffix_input_ids = tokenizer(suffix, return_tensors="pt").input_ids.to(device)
    return torch.cat([embed_tokens(prefix_input_ids), inputs_embeds, embed_tokens(suffix_input_ids)[:, 1:, :]], dim=1)


def construct_attention_mask(seq_lens):
    max_len = max(seq_lens)
    return torch.stack([F.pad(torch.ones(n), (max_len - n, 0)) for n in seq_lens]).long()


def batch_full_embed_sequence(all_audio_embeds, all_text_input_ids, all_response_input_ids, tokenizer, embed_tokens,
                              llm_type, device, process_text=False):
    """ref:utils.py:85-164: left-padded audio / text training sequences and their masks."""
    prefix, suffix = prompt_template(llm_type)
    prefix_ids = tokenizer(prefix, return_tensors="pt").input_ids.to(device)
    suffix_ids = tokenizer(suffix, return_tensors="pt").input_ids.to(device)
    audio_seqs, text_seqs = [], []
    for audio_embeds, text_ids, response_ids in zip(all_audio_embeds, all_text_input_ids, all_response_input_ids):
        audio_seqs.append(merge_prompt_response_tokens(prefix_ids, suffix_ids, audio_embeds.unsqueeze(0),
                                                       response_ids.unsqueeze(0).to(device), embed_tokens))
        if process_text:
            text_seqs.append(merge_prompt_response_tokens(prefix_ids, suffix_ids, embed_tokens(text_ids.unsqueeze(0).to(device)),
                                                          response_ids.unsqueeze(0).to(device), embed_tokens))

    def pad(seqs):
        lens = [s.shape[1] for s in seqs]
        m = max(lens)
        return torch.cat([F.pad(s, (0, 0, m - s.shape[1], 0)) for s in seqs]), construct_attention_mask(lens)

    a, am = pad(audio_seqs)
    t, tm = pad(text_seqs) if process_text else (None, None)
    return a, am, t, tm


def soft_cross_entropy(input, target, reduction="mean"):
    """ref:utils.py:167-178 — mean over positions of -sum_v softmax(target)_v * log_softmax(input)_v — on the HIP path
    (sl_soft_ce_loss, fp32 statistics).  Inputs are GPU tensors (..., V); returns a 0-d tensor (`reduction="mean"`) or the
    per-position values.  The KD step itself calls the same kernel with its gradient output (training.KDTrainer); this
    entry point is the reference's forward-only helper."""
    from . import _lib as L
    from . import ops
    L.require_gpu(input.contiguous(), "input")
    V = input.shape[-1]
    s = input.reshape(-1, V).float().contiguous()
    t = target.reshape(-1, V).float().contiguous()
    rows = s.shape[0]
    if reduction == "mean":
        loss = torch.zeros(1, device=s.device, dtype=torch.float32)
        ops.soft_ce_loss(s, t, 1.0 / rows, loss, None)
        return loss[0]
    out = torch.zeros(rows, device=s.device, dtype=torch.float32)
    for r in range(rows):
        ops.soft_ce_loss(s[r:r + 1], t[r:r + 1], 1.0, out[r:r + 1], None)
    return out.view(input.shape[:-1])
