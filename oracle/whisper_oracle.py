"""Oracle (test infrastructure): Whisper log-mel front end + encoder + the reference's pool/projection, fp32 CPU.

Restates:
  * hf:models/whisper/feature_extraction_whisper.py:69-103 (mel bank parameters), :135-168 (torch feature path:
    pad/trim to chunk, Hann STFT hop 160, drop last frame, power, slaney mel, clamp/log10, max-8 floor, (x+4)/4)
    as invoked by ref:trainer.py:178-182 (collate_audio_batch_whisper)
  * hf:models/whisper/modeling_whisper.py:592-646 (WhisperEncoder.forward), :360-414 (encoder layer), attention with
    q scaled by head_dim**-0.5 and a bias-less k_proj
  * ref:model/audio_encoder.py:56-63,87 (pool + projection) and ref:trainer.py:280-291 (crop to compute_num_audio_embeds)
The mel filter bank is the published slaney definition (Auditory Toolbox / librosa), restated here independently.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F


@dataclass
class WhisperCfg:
    d_model: int = 1024
    encoder_layers: int = 24
    encoder_attention_heads: int = 16
    encoder_ffn_dim: int = 4096
    num_mel_bins: int = 80
    max_source_positions: int = 1500
    n_fft: int = 400
    hop_length: int = 160
    sampling_rate: int = 16000

    @property
    def n_frames(self) -> int:
        return 2 * self.max_source_positions


def mel_filters(n_freqs: int, n_mels: int, sr: int = 16000, fmin: float = 0.0, fmax: float = 8000.0) -> torch.Tensor:
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0
    hz2mel = lambda f: min_log_mel + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp
    pts = torch.linspace(hz2mel(fmin), hz2mel(fmax), n_mels + 2, dtype=torch.float64)
    hz = torch.where(pts >= min_log_mel, min_log_hz * torch.exp(logstep * (pts - min_log_mel)), f_sp * pts)
    freqs = torch.linspace(0, sr // 2, n_freqs, dtype=torch.float64)
    fb = torch.zeros(n_freqs, n_mels, dtype=torch.float64)
    for i in range(n_mels):
        lo, ce, hi = hz[i], hz[i + 1], hz[i + 2]
        fb[:, i] = torch.clamp(torch.minimum((freqs - lo) / (ce - lo), (hi - freqs) / (hi - ce)), min=0.0) * (2.0 / (hi - lo))
    return fb.float()


def log_mel(cfg: WhisperCfg, wave: torch.Tensor) -> torch.Tensor:
    """1-D waveform -> (n_mel, n_frames) float32, as WhisperFeatureExtractor(..., return_tensors="pt").input_features[0]."""
    n = cfg.n_frames * cfg.hop_length
    x = torch.zeros(n, dtype=torch.float32)
    m = min(n, wave.numel())
    x[:m] = wave.float()[:m]
    stft = torch.stft(x, cfg.n_fft, cfg.hop_length, window=torch.hann_window(cfg.n_fft), return_complex=True)
    mag = stft[..., :-1].abs() ** 2
    mel = mel_filters(cfg.n_fft // 2 + 1, cfg.num_mel_bins, cfg.sampling_rate).T @ mag
    ls = torch.clamp(mel, min=1e-10).log10()
    ls = torch.maximum(ls, ls.max() - 8.0)
    return (ls + 4.0) / 4.0


def whisper_encoder_forward(sd: Dict[str, torch.Tensor], cfg: WhisperCfg, feats: torch.Tensor, prefix: str = "encoder.") -> torch.Tensor:
    """feats (B, n_mel, n_frames) -> (B, max_source_positions, d_model)."""
    p = prefix
    x = F.gelu(F.conv1d(feats.float(), sd[p + "conv1.weight"].float(), sd[p + "conv1.bias"].float(), padding=1))
    x = F.gelu(F.conv1d(x, sd[p + "conv2.weight"].float(), sd[p + "conv2.bias"].float(), stride=2, padding=1))
    x = x.permute(0, 2, 1) + sd[p + "embed_positions.weight"].float()
    B, T, H = x.shape
    nh = cfg.encoder_attention_heads
    hd = H // nh
    for li in range(cfg.encoder_layers):
        q = f"{p}layers.{li}."
        a = q + "self_attn."
        res = x
        h = F.layer_norm(x, (H,), sd[q + "self_attn_layer_norm.weight"].float(), sd[q + "self_attn_layer_norm.bias"].float(), 1e-5)
        qs = (F.linear(h, sd[a + "q_proj.weight"].float(), sd[a + "q_proj.bias"].float()) * hd ** -0.5).view(B, T, nh, hd).transpose(1, 2)
        ks = F.linear(h, sd[a + "k_proj.weight"].float()).view(B, T, nh, hd).transpose(1, 2)
        vs = F.linear(h, sd[a + "v_proj.weight"].float(), sd[a + "v_proj.bias"].float()).view(B, T, nh, hd).transpose(1, 2)
        att = F.softmax(torch.matmul(qs, ks.transpose(2, 3)), dim=-1)
        o = torch.matmul(att, vs).transpose(1, 2).reshape(B, T, H)
        x = res + F.linear(o, sd[a + "out_proj.weight"].float(), sd[a + "out_proj.bias"].float())
        res = x
        h = F.layer_norm(x, (H,), sd[q + "final_layer_norm.weight"].float(), sd[q + "final_layer_norm.bias"].float(), 1e-5)
        h = F.gelu(F.linear(h, sd[q + "fc1.weight"].float(), sd[q + "fc1.bias"].float()))
        x = res + F.linear(h, sd[q + "fc2.weight"].float(), sd[q + "fc2.bias"].float())
    return F.layer_norm(x, (H,), sd[p + "layer_norm.weight"].float(), sd[p + "layer_norm.bias"].float(), 1e-5)


def audio_encoder_forward(sd, cfg: WhisperCfg, feats: torch.Tensor, kernel_size: int = 8, stride: int = 4) -> torch.Tensor:
    enc = whisper_encoder_forward(sd, cfg, feats)
    pooled = F.avg_pool1d(enc.transpose(1, 2), kernel_size=kernel_size, stride=stride).transpose(1, 2)
    return F.linear(pooled, sd["embed_projection.weight"].float(), sd["embed_projection.bias"].float())
