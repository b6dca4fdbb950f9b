"""AudioEncoder: host-side mirror of ref:model/audio_encoder.py:16-88 driving the HIP encoder.

Same constructor arguments, attributes (`downsample_method`, `downsample_factor`, `encoder_base`) and
`forward(input, ctc_pool_ranges=None) -> (B, P, llm_dim)` contract, same exceptions for bad configs.
The arithmetic (HuBERT feature extractor, transformer, pooling, projection) runs in libspeechllm via
`sl_hubert_forward`; this class only owns buffers and weight tables.  There is no PyTorch fallback.

Differences from the reference, all deliberate:
  * the encoder architecture comes from a local `config.json` or the built-in table of hub ids (no network);
  * `forward` also accepts a list of 1-D waveforms of different lengths (each encoded at its own length);
  * `stack` with T % factor == 0 keeps all frames instead of returning an empty sequence (Q5);
  * Whisper (`base: whisper`): `forward` takes the (B, n_mel, 3000) log-mel features like the reference's trainer path
    (ref:trainer.py:168-199) and `.feature_extractor(raw_audios, return_tensors="pt", sampling_rate=...)` computes them
    on the GPU (sl_whisper_logmel) with the HF feature extractor's call shape; the crop to
    `compute_num_audio_embeds` stays with the caller, as in ref:trainer.py:280-291.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional, Sequence, Union

import torch

from . import _lib as L
from . import ops
from .weights import (KNOWN_HUBERT, KNOWN_WHISPER, HubertArch, HubertDeviceWeights, WhisperArch, WhisperDeviceWeights,
                      init_embed_projection, normalize_encoder_state_dict, pretrained_encoder_state_dict, resolve_pretrained_dir)


def resolve_whisper_arch(type_str: str) -> WhisperArch:
    local = resolve_pretrained_dir(type_str)
    if local and os.path.exists(os.path.join(local, "config.json")):
        with open(os.path.join(local, "config.json")) as f:
            return WhisperArch.from_hf_config(json.load(f))
    if type_str in KNOWN_WHISPER:
        return KNOWN_WHISPER[type_str]
    raise L.SpeechLLMError(f"unknown audio encoder '{type_str}': give a local directory with config.json or one of {list(KNOWN_WHISPER)}")


class WhisperFeatures:
    """GPU counterpart of HF's WhisperFeatureExtractor call used at ref:trainer.py:178-182."""

    def __init__(self, owner: "AudioEncoder"):
        self._o = owner
        self.sampling_rate = 16000

    def __call__(self, raw_audios, return_tensors="pt", sampling_rate=16000):
        from types import SimpleNamespace
        o = self._o
        if sampling_rate != self.sampling_rate:
            raise ValueError(f"The model corresponding to this feature extractor was trained using a sampling rate of {self.sampling_rate}")
        if o.weights is None:
            raise L.SpeechLLMError("AudioEncoder weights are not on the GPU: the log-mel front end runs on the HIP path only")
        a, w, lib = o.arch, o.weights, L.lib()
        if torch.is_tensor(raw_audios) and raw_audios.dim() == 1 or not isinstance(raw_audios, (list, tuple)) and getattr(raw_audios, "ndim", 2) == 1:
            raw_audios = [raw_audios]
        nbytes = lib.sl_whisper_logmel_workspace_bytes(a.n_fft, a.hop_length, a.n_frames, a.num_mel_bins)
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=o.device)
        feats = torch.empty((len(raw_audios), a.n_frames, a.num_mel_bins), device=o.device, dtype=torch.float32)
        for i, wav in enumerate(raw_audios):
            x = torch.as_tensor(wav, dtype=torch.float32).reshape(-1).to(o.device).contiguous()
            L.check(lib.sl_whisper_logmel(x.data_ptr(), x.numel(), w.dft_basis.data_ptr(), w.mel_w.data_ptr(), feats[i].data_ptr(), a.n_fft,
                                          a.hop_length, a.n_frames, a.num_mel_bins, ws.data_ptr(), ws.numel(), L.SL_F32, L.stream_ptr()),
                    "sl_whisper_logmel")
        return SimpleNamespace(input_features=feats.transpose(1, 2))   # (B, n_mel, n_frames) view, as HF returns


def resolve_hubert_arch(type_str: str) -> HubertArch:
    local = resolve_pretrained_dir(type_str)
    if local and os.path.exists(os.path.join(local, "config.json")):
        with open(os.path.join(local, "config.json")) as f:
            return HubertArch.from_hf_config(json.load(f))
    if type_str in KNOWN_HUBERT:
        return KNOWN_HUBERT[type_str]
    raise L.SpeechLLMError(f"unknown audio encoder '{type_str}': give a local directory with config.json or one of {list(KNOWN_HUBERT)}")


class AudioEncoder:
    def __init__(self, config, device, dtype: torch.dtype = torch.bfloat16, arch: Optional[HubertArch] = None,
                 load_pretrained: bool = True):
        self.config = config
        self.device = torch.device(device)
        self.dtype = dtype
        base = self.config.model.audio_encoder.base
        if base == "hubert":
            self.encoder_base = "hubert"
            self.arch = arch or resolve_hubert_arch(self.config.model.audio_encoder.type)
        elif base == "whisper":
            self.encoder_base = "whisper"
            self.arch = arch or resolve_whisper_arch(self.config.model.audio_encoder.type)
            self.feature_extractor = WhisperFeatures(self)
        else:
            raise Exception("Unexpected encoder type in config.")
        self.downsample_method = self.config.model.audio_encoder.downsample_method
        self.downsample_factor = self.config.model.audio_encoder.downsample_factor
        if self.downsample_method not in ("pool", "stack", "ctc_pool"):
            raise Exception("Invalid downsampling method for audio encoder.")
        self.pool_kernel = self.pool_stride = 0
        if self.downsample_method == "pool":
            self.pool_kernel = self.config.model.audio_encoder.pooling.kernel_size
            self.pool_stride = self.config.model.audio_encoder.pooling.stride
        self.llm_dim = self.config.model.llm_embedding_channels
        self.weights: Optional[HubertDeviceWeights] = None
        self._state: Optional[Dict[str, torch.Tensor]] = None
        self._ws: Optional[torch.Tensor] = None
        self.training = False
        self.last_encode_ms = None
        self.pretrained_from: Optional[str] = None
        if load_pretrained and arch is None:
            self._cold_start()

    def _cold_start(self) -> None:
        """ref:model/audio_encoder.py:6-13,34-52: the reference's constructor leaves the module holding the PRETRAINED encoder
        (`AutoModel.from_pretrained(type)`; `.encoder` of it for Whisper) and a freshly initialised `embed_projection`.  Same
        here whenever the weights can be found without a network (a local HF directory, or the hub id's snapshot in the local HF
        cache); otherwise the object stays weightless until `load_state_dict` (the inference path loads a checkpoint next
        anyway, ref:inference.py:24-26) and `state_dict()` says what is missing."""
        local = resolve_pretrained_dir(self.config.model.audio_encoder.type)
        if local is None:
            return
        sd = pretrained_encoder_state_dict(local, self.encoder_base)
        if sd is None:
            return
        in_dim = self.arch.hidden_size * (self.downsample_factor if self.downsample_method == "stack" else 1)
        seed = getattr(self.config, "seed_everything", 0)
        sd.update(init_embed_projection(in_dim, self.llm_dim, 0 if seed is None else int(seed)))
        self.pretrained_from = local
        self.load_state_dict(sd)

    # -- nn.Module-like surface the reference's callers use ------------------------------------
    def load_state_dict(self, state_dict, strict: bool = True):
        sd = normalize_encoder_state_dict(state_dict)
        self._state = {k: v.detach().float() for k, v in sd.items()}  # kept where they live (CPU checkpoints, GPU master weights)
        if self.device.type == "cuda":
            self._upload()
        return self

    def refresh_weights(self, state_dict) -> "AudioEncoder":
        """load_state_dict for weights that only changed VALUE (an optimizer step): the device tensors are updated in place
        (weights.*DeviceWeights.refresh) instead of being rebuilt."""
        if self.weights is None:
            return self.load_state_dict(state_dict)
        sd = normalize_encoder_state_dict(state_dict)
        self._state = {k: v.detach().float() for k, v in sd.items()}
        self.weights.refresh(self._state)
        return self

    def state_dict(self):
        if self._state is None:
            raise L.SpeechLLMError(
                f"AudioEncoder has no weights yet: no pretrained checkpoint for '{self.config.model.audio_encoder.type}' was found "
                "(model.audio_encoder.type must be a local HF directory with config.json + model.safetensors / pytorch_model.bin, or "
                "a hub id already in the local HF cache — there is no network), and load_state_dict has not been called")
        return dict(self._state)

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        """nn.Module surface (ref:trainer.py:258 calls audio_encoder.train()).  The flag is recorded; the differentiable,
        regularised forward of the optimisation step lives in the training tape (training.EncoderTape / WhisperEncoderTape,
        driven by KDTrainer / Trainer), `forward()` itself stays the inference path and refuses to run while the flag is set."""
        self.training = bool(mode)
        return self

    def to(self, device):
        self.device = torch.device(device)
        if self._state is not None and self.device.type == "cuda":
            self._upload()
        return self

    def _upload(self):
        if self.encoder_base == "whisper":
            self.weights = WhisperDeviceWeights(self.arch, self._state, self.llm_dim, self.device, self.dtype,
                                                pool_kernel=max(self.pool_kernel, 1), pool_stride=max(self.pool_stride, 1),
                                                downsample=self.downsample_method)
            return
        self.weights = HubertDeviceWeights(self.arch, self._state, self.llm_dim, self.device, self.dtype,
                                           pool_kernel=max(self.pool_kernel, 1), pool_stride=max(self.pool_stride, 1),
                                           downsample=self.downsample_method)

    # -- forward -------------------------------------------------------------------------------
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def encode_packed(self, waves: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None,
                      out_row_offsets: Optional[Sequence[int]] = None, want_last_hidden: bool = False):
        """Encode utterances of arbitrary lengths.  Returns (out, P_list, last_hidden|None, T_list).

        `out` (rows, llm_dim) + `out_row_offsets` let the encoder write each utterance's embeddings
        directly at its place inside a prompt buffer (ref:utils.py:66-72 concatenation becomes a no-op).
        """
        if self.weights is None:
            raise L.SpeechLLMError("AudioEncoder weights are not on the GPU: call load_state_dict(...).to('cuda')")
        lib = L.lib()
        if getattr(self.weights, "fold_stale", False):       # a fused optimizer step moved the layer weights since the folded copies were made
            from .weights import build_layernorm_fold
            build_layernorm_fold(self.weights)
        m = self.weights.struct
        lens = [int(w.numel()) for w in waves]
        offs = [0]
        for n in lens:
            offs.append(offs[-1] + n)
        flat = torch.cat([w.reshape(-1).to(device=self.device, dtype=torch.float32) for w in waves]).contiguous()
        n_utt = len(waves)
        T = [self.arch.num_frames(n) for n in lens]
        pool = self.downsample_method == "pool"
        P = [((t - self.pool_kernel) // self.pool_stride + 1) if pool else 0 for t in T]
        last_hidden = None
        if want_last_hidden or not pool:
            last_hidden = torch.empty((sum(T), self.arch.hidden_size), device=self.device, dtype=self.dtype)
        if pool and out is None:
            out = torch.empty((sum(P), self.llm_dim), device=self.device, dtype=self.dtype)
        if pool and out_row_offsets is None:
            out_row_offsets, acc_rows = [], 0
            for p_ in P:
                out_row_offsets.append(acc_rows)
                acc_rows += p_
        # Utterances are independent, so a large batch goes through the library in groups of at most `max_samples_per_call`
        # samples (default 256 x 10 s): the conv stack's workspace grows with the audio in flight (33 KB per sample-ms: ~20 GB
        # per 2 560 audio-seconds), and 1 024 x 10 s in one call would hold ~75 GB of it for no gain — the GEMMs of a 256-utterance
        # group already have 128 k rows.
        limit = int(getattr(self, "max_samples_per_call", 256 * 160000))
        t_row = 0
        start = 0
        while start < n_utt:
            end = start + 1
            while end < n_utt and offs[end + 1] - offs[start] <= limit:
                end += 1
            n_c = end - start
            offs_c = (C.c_int64 * (n_c + 1))(*[o - offs[start] for o in offs[start:end + 1]])
            nbytes = lib.sl_hubert_workspace_bytes(C.byref(m), offs_c, n_c)
            if nbytes == 0:
                raise L.SpeechLLMError("sl_hubert_workspace_bytes: " + lib.sl_last_error().decode())
            ws = self._workspace(nbytes)
            rows_c = (C.c_int64 * n_c)(*[int(r) for r in out_row_offsets[start:end]]) if pool else None
            lh_ptr = (last_hidden.data_ptr() + t_row * last_hidden.stride(0) * last_hidden.element_size()) if last_hidden is not None else None
            L.check(lib.sl_hubert_forward(C.byref(m), flat.data_ptr() + offs[start] * 4, offs_c, n_c, L.ptr(out), (out.stride(0) if out is not None else 0),
                                          rows_c, lh_ptr, ws.data_ptr(), ws.numel(), L.stream_ptr()), "sl_hubert_forward")
            t_row += sum(T[start:end])
            start = end
        return out, P, last_hidden, T

    def _downsample_host(self, hidden: torch.Tensor, ctc_pool_ranges) -> torch.Tensor:
        """stack / ctc_pool on one utterance's (T, H) frames, composed from HIP ops (ref:model/audio_encoder.py:65-87)."""
        w, b = self.weights.proj_w, self.weights.proj_b
        if self.downsample_method == "stack":
            f = self.downsample_factor
            keep = (hidden.shape[0] // f) * f  # Q5 fix: crop only the remainder
            stacked = hidden[:keep].reshape(keep // f, f * hidden.shape[1]).contiguous()
            return ops.gemm(stacked, w, bias=b)
        assert ctc_pool_ranges is not None, "Need to specify CTC pool ranges if using ctc_pool downsample method."
        ranges = torch.tensor([list(r) for r in ctc_pool_ranges[0]], dtype=torch.int32, device=self.device)
        pooled = ops.avgpool_rows(hidden, ranges=ranges)
        return ops.gemm(pooled, w, bias=b)

    def forward(self, input: Union[torch.Tensor, List[torch.Tensor]], ctc_pool_ranges=None) -> torch.Tensor:
        if self.weights is None:
            raise L.SpeechLLMError("AudioEncoder weights are not on the GPU: call load_state_dict(...).to('cuda') — "
                                   "the encoder runs on the HIP path only")
        if getattr(self, "training", False):
            raise L.SpeechLLMError("AudioEncoder.forward is the inference path; in train() mode run the step through training.KDTrainer "
                                   "(or Trainer), whose tape applies the dropouts / LayerDrop / SpecAugment and the backward — or call eval()")
        if self.encoder_base == "whisper":
            return self._forward_whisper(input)
        if torch.is_tensor(input):
            if input.dim() == 1:
                input = input[None]
            waves = [input[i] for i in range(input.shape[0])]
        else:
            waves = list(input)
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        start.record()
        if self.downsample_method == "pool":
            out, P, _, _ = self.encode_packed(waves)
            end.record()
            self._events = (start, end)
            if len(set(P)) != 1:
                raise L.SpeechLLMError("forward() returns a dense (B,P,C) tensor: use encode_packed for ragged batches")
            return out.view(len(waves), P[0], self.llm_dim)
        outs = []
        _, _, hidden, T = self.encode_packed(waves, want_last_hidden=True)
        t0 = 0
        for t in T:
            outs.append(self._downsample_host(hidden[t0:t0 + t], ctc_pool_ranges))
            t0 += t
        end.record()
        self._events = (start, end)
        if len(outs) != 1:
            # the reference's stack / ctc_pool paths assume batch size 1 (ref:model/audio_encoder.py:68,77)
            raise L.SpeechLLMError("stack / ctc_pool downsampling assumes batch size 1, as the reference does")
        return outs[0][None]

    def _forward_whisper(self, input_features: torch.Tensor) -> torch.Tensor:
        """(B, n_mel, 2*max_source_positions) log-mel -> (B, P, llm_dim); WhisperEncoder raises on any other length
        (hf:models/whisper/modeling_whisper.py:612-616) and so does this."""
        a, lib = self.arch, L.lib()
        if self.downsample_method != "pool":
            raise L.SpeechLLMError("the Whisper path is built for the `pool` downsample (all shipped configs use it)")
        if input_features.dim() != 3 or input_features.shape[1] != a.num_mel_bins or input_features.shape[2] != a.n_frames:
            raise ValueError(f"Whisper expects the mel input features to be of length {a.n_frames}, but found {tuple(input_features.shape)}. "
                             f"Make sure to pad the input mel features to {a.n_frames}.")
        B = input_features.shape[0]
        if getattr(self.weights, "fold_stale", False):
            from .weights import build_layernorm_fold
            build_layernorm_fold(self.weights)
        mel = input_features.to(self.device).transpose(1, 2).to(self.dtype).contiguous()       # channel-last rows for the implicit-GEMM convs
        m = self.weights.struct
        ws = self._workspace(lib.sl_whisper_workspace_bytes(C.byref(m), B))
        P = (a.max_source_positions - self.pool_kernel) // self.pool_stride + 1
        out = torch.empty((B * P, self.llm_dim), device=self.device, dtype=self.dtype)
        L.check(lib.sl_whisper_forward(C.byref(m), mel.data_ptr(), B, out.data_ptr(), out.stride(0), None, None, ws.data_ptr(), ws.numel(),
                                       L.stream_ptr()), "sl_whisper_forward")
        return out.view(B, P, self.llm_dim)

    __call__ = forward
