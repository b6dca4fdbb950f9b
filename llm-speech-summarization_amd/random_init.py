"""Seeded random-init state dicts in the reference's checkpoint layouts.

There is no network for pretrained checkpoints, so benchmarks, parity tests and the golden-fixture
generator all build weights from these functions (same seed -> same tensors on every box: the
generator is a CPU `torch.Generator`).  Key names follow what the reference loads:
  * AudioEncoder flat state-dict, ref:inference.py:24-26 (`encoder.*` HF HubertModel + `embed_projection.*`)
  * HF LlamaForCausalLM state-dict, ref:inference.py:47-52.
Norm gains/biases are perturbed away from 1/0 so that parity tests can see them.
"""
from __future__ import annotations

import math
from typing import Dict

import torch


def _g(seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return g


def _randn(g, *shape, std=1.0):
    return torch.randn(*shape, generator=g, dtype=torch.float32) * std


def hubert_encoder_state_dict(cfg, llm_dim: int, seed: int = 0, downsample: str = "pool",
                              downsample_factor: int = 4, weight_norm_keys: str = "parametrizations",
                              lin_std: float = 0.02) -> Dict[str, torch.Tensor]:
    """cfg: any object with the HubertCfg fields (conv_dim, conv_kernel, hidden_size, ...)."""
    g = _g(seed)
    sd: Dict[str, torch.Tensor] = {}
    H = cfg.hidden_size
    sd["encoder.masked_spec_embed"] = torch.rand(H, generator=g)
    cin = 1
    for i, (c, k) in enumerate(zip(cfg.conv_dim, cfg.conv_kernel)):
        p = f"encoder.feature_extractor.conv_layers.{i}."
        sd[p + "conv.weight"] = _randn(g, c, cin, k, std=math.sqrt(2.0 / (cin * k)))  # kaiming-normal
        sd[p + "conv.bias"] = _randn(g, c, std=0.02)
        sd[p + "layer_norm.weight"] = 1.0 + _randn(g, c, std=0.1)
        sd[p + "layer_norm.bias"] = _randn(g, c, std=0.05)
        cin = c
    p = "encoder.feature_projection."
    sd[p + "layer_norm.weight"] = 1.0 + _randn(g, cin, std=0.1)
    sd[p + "layer_norm.bias"] = _randn(g, cin, std=0.05)
    sd[p + "projection.weight"] = _randn(g, H, cin, std=lin_std * 2)
    sd[p + "projection.bias"] = _randn(g, H, std=0.02)
    p = "encoder.encoder.pos_conv_embed.conv."
    kpos, groups = cfg.num_conv_pos_embeddings, cfg.num_conv_pos_embedding_groups
    v = _randn(g, H, H // groups, kpos, std=2 * math.sqrt(4.0 / (kpos * H)))
    gain = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt() * (1.0 + _randn(g, 1, 1, kpos, std=0.1))
    if weight_norm_keys == "parametrizations":
        sd[p + "parametrizations.weight.original0"] = gain
        sd[p + "parametrizations.weight.original1"] = v
    else:  # torch<2.1 spelling used by the released checkpoint
        sd[p + "weight_g"] = gain
        sd[p + "weight_v"] = v
    sd[p + "bias"] = _randn(g, H, std=0.02)
    for li in range(cfg.num_hidden_layers):
        p = f"encoder.encoder.layers.{li}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"attention.{n}.weight"] = _randn(g, H, H, std=lin_std)
            sd[p + f"attention.{n}.bias"] = _randn(g, H, std=0.02)
        sd[p + "layer_norm.weight"] = 1.0 + _randn(g, H, std=0.1)
        sd[p + "layer_norm.bias"] = _randn(g, H, std=0.05)
        sd[p + "feed_forward.intermediate_dense.weight"] = _randn(g, cfg.intermediate_size, H, std=lin_std)
        sd[p + "feed_forward.intermediate_dense.bias"] = _randn(g, cfg.intermediate_size, std=0.02)
        sd[p + "feed_forward.output_dense.weight"] = _randn(g, H, cfg.intermediate_size, std=lin_std)
        sd[p + "feed_forward.output_dense.bias"] = _randn(g, H, std=0.02)
        sd[p + "final_layer_norm.weight"] = 1.0 + _randn(g, H, std=0.1)
        sd[p + "final_layer_norm.bias"] = _randn(g, H, std=0.05)
    sd["encoder.encoder.layer_norm.weight"] = 1.0 + _randn(g, H, std=0.1)
    sd["encoder.encoder.layer_norm.bias"] = _randn(g, H, std=0.05)
    in_dim = H * downsample_factor if downsample == "stack" else H
    sd["embed_projection.weight"] = _randn(g, llm_dim, in_dim, std=lin_std)
    sd["embed_projection.bias"] = _randn(g, llm_dim, std=0.02)
    return sd


def llama_state_dict(cfg, seed: int = 0, std: float = 0.02, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """cfg: any object with the LlamaCfg fields.  Generated layer by layer in `dtype` to bound memory."""
    g = _g(seed)
    H, hd = cfg.hidden_size, cfg.head_dim
    nh, nkv, F_ = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.intermediate_size
    sd: Dict[str, torch.Tensor] = {}
    sd["model.embed_tokens.weight"] = _randn(g, cfg.vocab_size, H, std=std).to(dtype)
    for li in range(cfg.num_hidden_layers):
        p = f"model.layers.{li}."
        sd[p + "self_attn.q_proj.weight"] = _randn(g, nh * hd, H, std=std).to(dtype)
        sd[p + "self_attn.k_proj.weight"] = _randn(g, nkv * hd, H, std=std).to(dtype)
        sd[p + "self_attn.v_proj.weight"] = _randn(g, nkv * hd, H, std=std).to(dtype)
        sd[p + "self_attn.o_proj.weight"] = _randn(g, H, nh * hd, std=std).to(dtype)
        sd[p + "mlp.gate_proj.weight"] = _randn(g, F_, H, std=std).to(dtype)
        sd[p + "mlp.up_proj.weight"] = _randn(g, F_, H, std=std).to(dtype)
        sd[p + "mlp.down_proj.weight"] = _randn(g, H, F_, std=std).to(dtype)
        sd[p + "input_layernorm.weight"] = (1.0 + _randn(g, H, std=0.1)).to(dtype)
        sd[p + "post_attention_layernorm.weight"] = (1.0 + _randn(g, H, std=0.1)).to(dtype)
    sd["model.norm.weight"] = (1.0 + _randn(g, H, std=0.1)).to(dtype)
    if not cfg.tie_word_embeddings:
        sd["lm_head.weight"] = _randn(g, cfg.vocab_size, H, std=std).to(dtype)
    return sd


def synthetic_waveform(n_samples: int, seed: int = 1234, std: float = 0.1) -> torch.Tensor:
    """N(0, 0.1^2) clipped to [-1, 1], 16 kHz mono, NOT normalised."""
    return (_randn(_g(seed), n_samples, std=std)).clamp_(-1.0, 1.0)


def synthetic_ids(n: int, vocab: int, seed: int = 7, bos: int = 0) -> torch.Tensor:
    ids = torch.randint(0, vocab, (1, n), generator=_g(seed), dtype=torch.int64)
    ids[0, 0] = bos
    return ids


def whisper_encoder_state_dict(cfg, llm_dim: int, seed: int = 0, lin_std: float = 0.02) -> Dict[str, torch.Tensor]:
    """AudioEncoder state-dict for the Whisper base (ref:model/audio_encoder.py:10-13): `encoder.*` = HF WhisperEncoder."""
    g = _g(seed)
    H, M = cfg.d_model, cfg.num_mel_bins
    sd: Dict[str, torch.Tensor] = {}
    sd["encoder.conv1.weight"] = _randn(g, H, M, 3, std=math.sqrt(2.0 / (3 * M)))
    sd["encoder.conv1.bias"] = _randn(g, H, std=0.02)
    sd["encoder.conv2.weight"] = _randn(g, H, H, 3, std=math.sqrt(2.0 / (3 * H)))
    sd["encoder.conv2.bias"] = _randn(g, H, std=0.02)
    half = H // 2
    inv = torch.exp(-math.log(10000.0) / (half - 1) * torch.arange(half))
    st = torch.arange(cfg.max_source_positions).view(-1, 1) * inv.view(1, -1)
    sd["encoder.embed_positions.weight"] = torch.cat([st.sin(), st.cos()], dim=1)     # hf:...modeling_whisper.py:55-64
    for li in range(cfg.encoder_layers):
        p = f"encoder.layers.{li}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sd[p + f"self_attn.{n}.weight"] = _randn(g, H, H, std=lin_std)
            if n != "k_proj":
                sd[p + f"self_attn.{n}.bias"] = _randn(g, H, std=0.02)
        sd[p + "self_attn_layer_norm.weight"] = 1.0 + _randn(g, H, std=0.1)
        sd[p + "self_attn_layer_norm.bias"] = _randn(g, H, std=0.05)
        sd[p + "fc1.weight"] = _randn(g, cfg.encoder_ffn_dim, H, std=lin_std)
        sd[p + "fc1.bias"] = _randn(g, cfg.encoder_ffn_dim, std=0.02)
        sd[p + "fc2.weight"] = _randn(g, H, cfg.encoder_ffn_dim, std=lin_std)
        sd[p + "fc2.bias"] = _randn(g, H, std=0.02)
        sd[p + "final_layer_norm.weight"] = 1.0 + _randn(g, H, std=0.1)
        sd[p + "final_layer_norm.bias"] = _randn(g, H, std=0.05)
    sd["encoder.layer_norm.weight"] = 1.0 + _randn(g, H, std=0.1)
    sd["encoder.layer_norm.bias"] = _randn(g, H, std=0.05)
    sd["embed_projection.weight"] = _randn(g, llm_dim, H, std=lin_std)
    sd["embed_projection.bias"] = _randn(g, llm_dim, std=0.02)
    return sd
