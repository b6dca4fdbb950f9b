"""Oracle (test infrastructure): HuBERT encoder + reference AudioEncoder, fp32 CPU.

Restates, op for op:
  * ref:model/audio_encoder.py:56-88  (AudioEncoder.forward: encoder -> downsample -> projection)
  * hf:models/hubert/modeling_hubert.py:127-151,203-213 (layer-norm conv feature extractor)
  * hf:...hubert.py:216-231 (feature projection), :45-103 (weight-normed positional conv),
    :504-547 (stable-layer-norm encoder layer), :262-344 (attention), :347-368 (FFN), :562-623 (encoder)

Weights are consumed in the state-dict layout of the reference's `AudioEncoder`
(`encoder.*` = HF HubertModel, `embed_projection.*`), with either weight-norm key spelling
(`conv.weight_g/weight_v` from torch<2.1 checkpoints or `conv.parametrizations.weight.original0/1`).

Only the HuBERT variant the reference configs use is covered (`feat_extract_norm="layer"`,
`do_stable_layer_norm=True`, i.e. facebook/hubert-large-ls960-ft, ref:config/llama3_hubert.yaml:16).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F


@dataclass
class HubertCfg:
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def num_frames(self, n_samples: int) -> int:
        """hf:models/hubert/modeling_hubert.py:664-677 (conv length recursion)."""
        length = n_samples
        for k, s in zip(self.conv_kernel, self.conv_stride):
            length = (length - k) // s + 1
        return length


HUBERT_LARGE = HubertCfg()


def pos_conv_weight(sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """Fold weight-norm (dim=2): w = g * v / ||v|| with the norm over dims (0,1) per tap.

    hf:models/hubert/modeling_hubert.py:58-78; torch.nn.utils.parametrizations.weight_norm(dim=2).
    """
    if prefix + "conv.parametrizations.weight.original0" in sd:
        g = sd[prefix + "conv.parametrizations.weight.original0"]
        v = sd[prefix + "conv.parametrizations.weight.original1"]
    elif prefix + "conv.weight_g" in sd:
        g = sd[prefix + "conv.weight_g"]
        v = sd[prefix + "conv.weight_v"]
    else:
        return sd[prefix + "conv.weight"].float()
    g = g.float()
    v = v.float()
    norm = v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
    return v * (g / norm)


def hubert_forward(
    sd: Dict[str, torch.Tensor],
    cfg: HubertCfg,
    wave: torch.Tensor,
    prefix: str = "encoder.",
    taps: Optional[dict] = None,
    train: Optional[dict] = None,
) -> torch.Tensor:
    """HubertModel.forward(...).last_hidden_state, no attention mask; eval mode unless `train` is given.

    `train` (training-mode regularisers of hf:models/hubert/modeling_hubert.py with the random draws SUPPLIED by the
    caller, so a test can replay the masks of the kernel path): {"drop": callable(site, layer, tensor) -> tensor applying
    that site's dropout (sites "fp" feature-projection dropout, "pos" encoder dropout after the positional embedding,
    "attn_out" / "ffn_out" hidden dropouts of a layer, "act" intermediate dropout), "skip": set of LayerDrop-skipped
    layers, "spec_mask": (B, T) bool SpecAugment mask (rows replaced by `masked_spec_embed`), "attn_drop":
    callable(layer, probabilities (B, nh, T, T)) -> dropped probabilities}.

    wave: (B, N) float32 raw 16 kHz samples (the reference feeds un-normalised audio and no mask:
    ref:model/audio_encoder.py:57).  Returns (B, T, hidden).  `taps`, if given, is filled with the
    per-stage tensors used by the kernel parity tests.
    """
    p = prefix
    x = wave.float()[:, None, :]  # hf:...hubert.py:204
    for i, (k, s) in enumerate(zip(cfg.conv_kernel, cfg.conv_stride)):
        q = f"{p}feature_extractor.conv_layers.{i}."
        x = F.conv1d(x, sd[q + "conv.weight"].float(), sd[q + "conv.bias"].float(), stride=s)
        x = x.transpose(-2, -1)
        x = F.layer_norm(x, (x.shape[-1],), sd[q + "layer_norm.weight"].float(),
                         sd[q + "layer_norm.bias"].float(), 1e-5)  # nn.LayerNorm default eps
        x = x.transpose(-2, -1)
        x = F.gelu(x)
        if taps is not None:
            taps[f"conv{i}"] = x.transpose(1, 2).contiguous()  # (B, L, C) channel-last
    x = x.transpose(1, 2)  # (B, T, C)  hf:...hubert.py:923

    q = f"{p}feature_projection."
    x = F.layer_norm(x, (x.shape[-1],), sd[q + "layer_norm.weight"].float(),
                     sd[q + "layer_norm.bias"].float(), cfg.layer_norm_eps)
    x = F.linear(x, sd[q + "projection.weight"].float(), sd[q + "projection.bias"].float())
    drop = (lambda site, layer, v: v) if train is None else train["drop"]
    x = drop("fp", 0, x)                                   # HubertFeatureProjection.dropout
    if train is not None and train.get("spec_mask") is not None:   # HubertModel._mask_hidden_states
        m = train["spec_mask"].to(torch.bool)
        x = torch.where(m[..., None], sd[f"{p}masked_spec_embed"].float().expand_as(x), x)
    if taps is not None:
        taps["feature_projection"] = x

    # positional conv embedding  hf:...hubert.py:91-101, 584-585
    q = f"{p}encoder.pos_conv_embed."
    kpos = cfg.num_conv_pos_embeddings
    w = pos_conv_weight(sd, q)
    pos = F.conv1d(x.transpose(1, 2), w, sd[q + "conv.bias"].float(), padding=kpos // 2,
                   groups=cfg.num_conv_pos_embedding_groups)
    if kpos % 2 == 0:
        pos = pos[:, :, :-1]
    pos = F.gelu(pos).transpose(1, 2)
    x = x + pos
    x = drop("pos", 0, x)                                  # HubertEncoderStableLayerNorm: dropout(hidden + pos_conv_embed(hidden))
    if taps is not None:
        taps["pos_conv"] = x

    B, T, H = x.shape
    nh, hd = cfg.num_attention_heads, cfg.head_dim
    for li in range(cfg.num_hidden_layers):
        q = f"{p}encoder.layers.{li}."
        if train is not None and li in train.get("skip", ()):   # LayerDrop
            continue
        res = x
        h = F.layer_norm(x, (H,), sd[q + "layer_norm.weight"].float(), sd[q + "layer_norm.bias"].float(),
                         cfg.layer_norm_eps)
        a = q + "attention."
        qs = F.linear(h, sd[a + "q_proj.weight"].float(), sd[a + "q_proj.bias"].float()).view(B, T, nh, hd).transpose(1, 2)
        ks = F.linear(h, sd[a + "k_proj.weight"].float(), sd[a + "k_proj.bias"].float()).view(B, T, nh, hd).transpose(1, 2)
        vs = F.linear(h, sd[a + "v_proj.weight"].float(), sd[a + "v_proj.bias"].float()).view(B, T, nh, hd).transpose(1, 2)
        att = torch.matmul(qs, ks.transpose(2, 3)) * (hd ** -0.5)  # hf:...hubert.py:248
        att = F.softmax(att, dim=-1)
        if train is not None and train.get("attn_drop") is not None:   # HubertAttention: dropout on the probabilities
            att = train["attn_drop"](li, att)                           # (B, nh, T, T)
        o = torch.matmul(att, vs).transpose(1, 2).reshape(B, T, H)
        o = F.linear(o, sd[a + "out_proj.weight"].float(), sd[a + "out_proj.bias"].float())
        x = res + drop("attn_out", li, o)
        h = F.layer_norm(x, (H,), sd[q + "final_layer_norm.weight"].float(),
                         sd[q + "final_layer_norm.bias"].float(), cfg.layer_norm_eps)
        f = q + "feed_forward."
        h = F.gelu(F.linear(h, sd[f + "intermediate_dense.weight"].float(), sd[f + "intermediate_dense.bias"].float()))
        h = drop("act", li, h)
        h = F.linear(h, sd[f + "output_dense.weight"].float(), sd[f + "output_dense.bias"].float())
        x = x + drop("ffn_out", li, h)
        if taps is not None:
            taps[f"layer{li}"] = x
    q = f"{p}encoder."
    x = F.layer_norm(x, (H,), sd[q + "layer_norm.weight"].float(), sd[q + "layer_norm.bias"].float(),
                     cfg.layer_norm_eps)
    if taps is not None:
        taps["last_hidden_state"] = x
    return x


def downsample(
    encoder_out: torch.Tensor,
    method: str,
    *,
    kernel_size: int = 8,
    stride: int = 4,
    factor: int = 4,
    ctc_pool_ranges: Optional[Sequence[Sequence[Tuple[int, int]]]] = None,
    fix_stack_quirk: bool = False,
) -> torch.Tensor:
    """ref:model/audio_encoder.py:59-85.

    `stack` reproduces the reference exactly, including `[:, :-0, :]` yielding an EMPTY sequence when
    T % factor == 0, unless `fix_stack_quirk` is set (the build's documented fix).
    """
    if method == "pool":
        return F.avg_pool1d(encoder_out.transpose(1, 2), kernel_size=kernel_size, stride=stride).transpose(1, 2)
    if method == "stack":
        to_crop = encoder_out.shape[1] % factor
        if fix_stack_quirk and to_crop == 0:
            cropped = encoder_out
        else:
            cropped = encoder_out[:, :-to_crop, :] if to_crop else encoder_out[:, :0, :]
        return cropped.reshape(1, -1, factor * encoder_out.shape[2])
    if method == "ctc_pool":
        assert ctc_pool_ranges is not None, "Need to specify CTC pool ranges if using ctc_pool downsample method."
        pooled = [encoder_out[:, s:e, :].mean(dim=1) for s, e in ctc_pool_ranges[0]]
        return torch.stack(pooled, dim=1)
    raise Exception("Invalid downsampling method for audio encoder.")


def audio_encoder_forward(
    sd: Dict[str, torch.Tensor],
    cfg: HubertCfg,
    wave: torch.Tensor,
    method: str = "pool",
    taps: Optional[dict] = None,
    train: Optional[dict] = None,
    **ds_kwargs,
) -> torch.Tensor:
    """ref:model/audio_encoder.py:56-88 -> (B, P, llm_dim)."""
    enc = hubert_forward(sd, cfg, wave, prefix="encoder.", taps=taps, train=train)
    pooled = downsample(enc, method, **ds_kwargs)
    if taps is not None:
        taps["pooled"] = pooled
    out = F.linear(pooled, sd["embed_projection.weight"].float(), sd["embed_projection.bias"].float())
    if taps is not None:
        taps["audio_embeds"] = out
    return out


def compute_num_audio_embeds(audio_samples: int, sr: int = 16000) -> int:
    """ref:utils.py:13-24, bit for bit (float floor-division then int())."""
    num_embeds = (audio_samples - (sr * 0.01)) // (sr * 0.02)
    return int(num_embeds // 4 - 1)
