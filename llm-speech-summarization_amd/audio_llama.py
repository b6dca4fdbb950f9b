"""AudioLlamaForCausalLM: host-side mirror of ref:model/audio_llama.py:18-113 driving the HIP Llama path.

Surface kept from the reference / HF base class it inherits: `from_pretrained(path, use_cache=True,
torch_dtype=...)`, `.model.embed_tokens(ids)`, `.forward(inputs_embeds=, attention_mask=, labels=,
output_hidden_states=)` -> object with `.logits/.hidden_states/.loss`, `.generate(input_ids=None,
inputs_embeds=, max_new_tokens=)` -> LongTensor(B, n_new) of NEW tokens only (the prompt is embeddings:
hf:generation/utils.py:736-744), `.eval()`, `.to(device)`, `.parameters()`.

RMSNorm, RoPE, GQA attention, SwiGLU, lm_head, argmax, EOS handling and the KV cache all run in
libspeechllm (sl_llama_prefill / sl_generate); generation is greedy by construction
(SURVEY.md §9 Q3).  No PyTorch fallback exists.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L
from . import ops
from .weights import KNOWN_LLAMA, LlamaArch, LlamaDeviceWeights


class _EmbedTokens:
    """`llm.model.embed_tokens(ids)` (ref:utils.py:63-64, ref:inference.py:85,121) as a HIP row gather."""

    def __init__(self, owner: "AudioLlamaForCausalLM"):
        self._o = owner

    def __call__(self, ids: torch.Tensor) -> torch.Tensor:
        w = self._o._dev()
        shape = tuple(ids.shape)
        out = ops.embed_gather(w.embed, ids)
        return out.view(*shape, w.arch.hidden_size)

    @property
    def weight(self) -> torch.Tensor:
        return self._o._dev().embed


class AudioLlamaForCausalLM:
    def __init__(self, arch: LlamaArch, state_dict: Dict[str, torch.Tensor], torch_dtype: torch.dtype = torch.bfloat16,
                 device=None, max_ctx: int = 2048, max_batch: int = 16, pack_decode: bool = True):
        if torch_dtype == torch.float16:
            # the reference runs fp16 autocast (ref:inference.py:50); gfx950 MFMA path here is bf16 / exact fp32
            torch_dtype = torch.bfloat16
        self.arch = arch
        self.dtype = torch_dtype
        self.config = SimpleNamespace(vocab_size=arch.vocab_size, hidden_size=arch.hidden_size,
                                      num_hidden_layers=arch.num_hidden_layers, eos_token_id=list(arch.eos_token_ids),
                                      pad_token_id=arch.pad_token_id, use_return_dict=True)
        # Greedy by default: BASELINE.json's north_star specifies greedy decode (the reference never passes
        # do_sample, so a hub generation_config.json decides; from_pretrained keeps that file's sampling parameters here and its
        # do_sample flag under `hub_do_sample`, but sampling is only used when a caller sets do_sample=True).
        self.generation_config = SimpleNamespace(eos_token_id=list(arch.eos_token_ids), pad_token_id=arch.pad_token_id,
                                                 do_sample=False, temperature=1.0, top_k=50, top_p=1.0, hub_do_sample=None)
        self.sample_seed = 0
        self._sd = state_dict
        self.device = torch.device("cpu")
        self.max_ctx, self.max_batch, self.pack_decode = max_ctx, max_batch, pack_decode
        self._w: Optional[LlamaDeviceWeights] = None
        self._kv = None
        self._ws: Optional[torch.Tensor] = None
        self.model = SimpleNamespace(embed_tokens=_EmbedTokens(self))
        self.last_timings_ms = None
        self.last_generate_stats = None
        if device is not None:
            self.to(device)

    # -- construction ---------------------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, name_or_path: str, use_cache: bool = True, torch_dtype: torch.dtype = torch.bfloat16, **kw):
        """Local directory with config.json + *.safetensors (HF layout).  Hub ids cannot be fetched
        (no network): a known id without a local directory raises with instructions."""
        if not os.path.isdir(name_or_path):
            raise L.SpeechLLMError(
                f"'{name_or_path}' is not a local directory. Download the checkpoint (config.json + *.safetensors) and pass "
                "its path; hub downloads are not available to this build.")
        with open(os.path.join(name_or_path, "config.json")) as f:
            arch = LlamaArch.from_hf_config(json.load(f))
        from safetensors.torch import load_file
        sd: Dict[str, torch.Tensor] = {}
        for fn in sorted(os.listdir(name_or_path)):
            if fn.endswith(".safetensors"):
                sd.update(load_file(os.path.join(name_or_path, fn)))
        if not sd:
            raise L.SpeechLLMError(f"no *.safetensors files under {name_or_path}")
        obj = cls(arch, sd, torch_dtype=torch_dtype, **kw)
        gc_path = os.path.join(name_or_path, "generation_config.json")
        if os.path.exists(gc_path):
            with open(gc_path) as f:
                gc = json.load(f)
            g = obj.generation_config
            g.temperature, g.top_k, g.top_p = float(gc.get("temperature", 1.0)), int(gc.get("top_k", 50)), float(gc.get("top_p", 1.0))
            g.hub_do_sample = bool(gc.get("do_sample", False))
            if gc.get("eos_token_id") is not None:
                e = gc["eos_token_id"]
                g.eos_token_id = list(e) if isinstance(e, (list, tuple)) else [e]
            if gc.get("pad_token_id") is not None:
                g.pad_token_id = gc["pad_token_id"]
        return obj

    def eval(self):
        return self

    def parameters(self):
        return iter(())  # frozen LLM: no trainable parameters are exposed (ref:trainer.py:63-64)

    def to(self, device):
        self.device = torch.device(device)
        if self.device.type == "cuda" and self._w is None:
            self._w = LlamaDeviceWeights(self.arch, self._sd, self.device, self.dtype, rope_len=max(self.max_ctx, 64))
            if self.pack_decode:
                self._w.build_decode_weights()
            self._sd = None  # device copy is the only copy from here on (6.4 GB bf16 for Llama-3.2-3B)
        return self

    def _dev(self) -> LlamaDeviceWeights:
        if self._w is None:
            raise L.SpeechLLMError("LLM weights are not on the GPU: call .to('cuda') — the hot path is HIP-only")
        return self._w

    # -- buffers --------------------------------------------------------------------------------
    def _kv_cache(self, slots: int, shared_prefix: int = 0):
        a = self.arch
        if self._kv is None or self._kv[0].shape[1] < slots or self._kv[0].shape[3] != self.max_ctx:
            self._kv = None        # release the smaller cache BEFORE the larger one is allocated (2 048 slots x 448 positions of Llama-3.2-3B: 105 GB)
            shape = (a.num_hidden_layers, slots, a.num_key_value_heads, self.max_ctx, a.head_dim)
            k = torch.zeros(shape, device=self.device, dtype=self.dtype)
            v = torch.zeros(shape, device=self.device, dtype=self.dtype)
            self._kv = (k, v)
        k, v = self._kv
        kv = L.KVCache()
        kv.k_cache, kv.v_cache, kv.slots, kv.max_ctx = k.data_ptr(), v.data_ptr(), k.shape[1], self.max_ctx
        kv.shared_prefix = int(shared_prefix)
        return kv

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    @staticmethod
    def _pack(inputs_embeds, attention_mask=None):
        """(B,S,h) [+ left-padding mask] or list of (S_i,h)  ->  packed (sum S_i, h), lengths."""
        if torch.is_tensor(inputs_embeds):
            B, S, _ = inputs_embeds.shape
            if attention_mask is None:
                return inputs_embeds.reshape(B * S, -1).contiguous().clone(), [S] * B
            lens = attention_mask.sum(dim=1).tolist()
            rows = [inputs_embeds[b, S - int(n):] for b, n in enumerate(lens)]  # left padding (ref:utils.py:76-82)
            return torch.cat(rows, 0).contiguous(), [int(n) for n in lens]
        return torch.cat(list(inputs_embeds), 0).contiguous(), [int(x.shape[0]) for x in inputs_embeds]

    # -- forward (prefill with all-position logits; training / validation callers) ----------------
    def forward(self, input_ids=None, attention_mask=None, inputs_embeds=None, labels=None, output_hidden_states=False,
                **unused):
        w = self._dev()
        a = self.arch
        if inputs_embeds is None:
            inputs_embeds = self.model.embed_tokens(input_ids)
        B, S = inputs_embeds.shape[0], inputs_embeds.shape[1]
        x, lens = self._pack(inputs_embeds.to(self.dtype), attention_mask)
        n_tok = x.shape[0]
        if max(lens) > self.max_ctx:
            raise L.SpeechLLMError(f"sequence of {max(lens)} tokens exceeds max_ctx={self.max_ctx}")
        lib = L.lib()
        kv = self._kv_cache(B)
        cu = [0]
        for n in lens:
            cu.append(cu[-1] + n)
        cu_c = (C.c_int32 * (B + 1))(*cu)
        ws = self._workspace(lib.sl_llama_workspace_bytes(C.byref(w.struct), n_tok, B))
        last_logits = torch.empty((B, a.vocab_size), device=self.device, dtype=torch.float32)
        ctx = torch.empty(B, device=self.device, dtype=torch.int32)
        taps = torch.empty((a.num_hidden_layers + 1, n_tok, a.hidden_size), device=self.device, dtype=self.dtype)
        L.check(lib.sl_llama_prefill(C.byref(w.struct), C.byref(kv), x.data_ptr(), cu_c, B, last_logits.data_ptr(), ctx.data_ptr(),
                                     taps.data_ptr(), ws.data_ptr(), ws.numel(), L.stream_ptr()), "sl_llama_prefill")
        # all-position logits from the post-norm hidden state (ref:model/audio_llama.py:67 with num_logits_to_keep=0)
        logits_packed = ops.gemm(taps[-1], w.lm_head, out_f32=True)

        def unpack(t):  # packed (n_tok, C) -> left-padded (B, S, C)
            out = t.new_zeros((B, S, t.shape[-1]))
            for b, n in enumerate(lens):
                out[b, S - n:] = t[cu[b]:cu[b + 1]]
            return out

        logits = unpack(logits_packed)
        hidden_states = tuple(unpack(taps[i]) for i in range(a.num_hidden_layers + 1)) if output_hidden_states else None
        loss = None
        if labels is not None:
            # per-sample response-only next-token CE, batch mean (ref:model/audio_llama.py:72-101): logits[-n:-1] vs labels[1:]
            acc = torch.zeros(1, device=self.device, dtype=torch.float32)
            for b, lab in enumerate(labels):
                lab = lab.reshape(-1).to(self.device)
                n = int(lab.shape[0])
                rows = logits_packed[cu[b + 1] - n: cu[b + 1] - 1]
                ops.ce_loss(rows, lab[1:].to(torch.int32).contiguous(), 1.0 / ((n - 1) * B), acc, None, dtype=self.dtype)
            loss = acc[0]
        return SimpleNamespace(loss=loss, logits=logits, hidden_states=hidden_states, past_key_values=None, attentions=None)

    __call__ = forward

    # -- generation -----------------------------------------------------------------------------
    def generate(self, input_ids=None, inputs_embeds=None, max_new_tokens: int = 256, attention_mask=None, use_eos: bool = True,
                 do_sample: Optional[bool] = None, temperature: Optional[float] = None, top_k: Optional[int] = None,
                 top_p: Optional[float] = None, seed: Optional[int] = None, **unused) -> torch.Tensor:
        w = self._dev()
        a = self.arch
        if inputs_embeds is None:
            if input_ids is None:
                raise L.SpeechLLMError("generate needs inputs_embeds or input_ids")
            inputs_embeds = self.model.embed_tokens(input_ids)
        x, lens = self._pack(inputs_embeds.to(self.dtype) if torch.is_tensor(inputs_embeds) else [e.to(self.dtype) for e in inputs_embeds],
                             attention_mask)
        g = self.generation_config
        sample = None
        if (g.do_sample if do_sample is None else do_sample):
            # hf:generation/utils.py logits warpers: temperature, top-k (HF default 50), top-p, then one draw per row
            if seed is None:
                seed, self.sample_seed = self.sample_seed, self.sample_seed + 1
            sample = dict(temperature=float(g.temperature if temperature is None else temperature), top_k=int(g.top_k if top_k is None else top_k),
                          top_p=float(g.top_p if top_p is None else top_p), seed=int(seed))
        ids, n_cols = self.generate_packed(x, lens, max_new_tokens, use_eos=use_eos, sample=sample)
        return ids[:, :n_cols].to(torch.int64)

    def generate_packed(self, x: torch.Tensor, lens: Sequence[int], max_new_tokens: int, use_eos: bool = True, sample: Optional[dict] = None,
                        shared_prefix: int = 0, row_limits: Optional[Sequence[int]] = None, compact: bool = True, check_every: int = 4):
        """x: packed prompt embeddings (sum S_i, h) on the GPU (overwritten).  Returns (int32 (B, max_new) host tensor, n_cols).
        shared_prefix = P: the caller's promise that the first P rows of every sequence are the same rows (one prompt template in
        front of the audio, ref:inference.py:95-113) — the batched decode attention then reads those P cache positions from slot 0
        (sl_kv_cache.shared_prefix); ids and logits are bit for bit those of P = 0.
        row_limits: one token budget per sequence (a per-request max_new_tokens; the row then finishes like a row that emitted EOS).
        compact: with EOS / budgets on, the batch is compacted as its rows finish (sl_generate_opts.compact) — the decode step then
        costs what the LIVE rows cost; per-sequence results are those of the uncompacted batch (`last_generate_stats` has the counts)."""
        w = self._dev()
        a = self.arch
        lib = L.lib()
        B = len(lens)
        if B > L.MAX_DECODE_BATCH:
            raise L.SpeechLLMError(f"{B} sequences in one generate call; the library takes {L.MAX_DECODE_BATCH} (split the batch: sequences are independent)")
        if max(lens) + max_new_tokens > self.max_ctx:
            raise L.SpeechLLMError(f"prompt ({max(lens)}) + max_new_tokens ({max_new_tokens}) exceeds max_ctx={self.max_ctx}")
        if not 0 <= shared_prefix <= min(lens):
            raise L.SpeechLLMError(f"shared_prefix={shared_prefix} outside [0, shortest prompt={min(lens)}]")
        kv = self._kv_cache(B, shared_prefix)
        cu = [0]
        for n in lens:
            cu.append(cu[-1] + int(n))
        cu_c = (C.c_int32 * (B + 1))(*cu)
        gen = self.generation_config
        eos = list(gen.eos_token_id) if isinstance(gen.eos_token_id, (list, tuple)) else ([] if gen.eos_token_id is None else [gen.eos_token_id])
        use_eos = bool(use_eos and len(eos) > 0)
        pad = gen.pad_token_id if gen.pad_token_id is not None else (eos[0] if eos else 0)
        eos_c = (C.c_int32 * max(1, len(eos)))(*eos)
        out = (C.c_int32 * (B * max_new_tokens))()
        o = L.GenerateOpts()
        # the ids are handed over as configured; with use_eos = 0 the library ignores them (rows then finish on their budgets only)
        o.eos_ids_host, o.n_eos, o.pad_id, o.use_eos = eos_c, len(eos), int(pad), int(use_eos)
        o.max_new_tokens, o.check_every, o.compact = int(max_new_tokens), int(check_every), int(bool(compact))
        lim_c = None
        if row_limits is not None:
            if len(row_limits) != B:
                raise L.SpeechLLMError(f"row_limits has {len(row_limits)} entries for {B} sequences")
            lim_c = (C.c_int32 * B)(*[int(v) for v in row_limits])
            o.row_limits_host = lim_c
        if sample is not None:
            o.sample, o.temperature, o.top_k, o.top_p = 1, float(sample["temperature"]), int(sample["top_k"]), float(sample["top_p"])
            o.seed = int(sample["seed"]) & 0xFFFFFFFFFFFFFFFF
        st = L.GenerateStats()
        ws = self._workspace(lib.sl_generate_workspace_bytes(C.byref(w.struct), x.shape[0], B, max_new_tokens))
        L.check(lib.sl_generate(C.byref(w.struct), C.byref(kv), x.data_ptr(), cu_c, B, C.byref(o), out, C.byref(st), ws.data_ptr(), ws.numel(),
                                L.stream_ptr()), "sl_generate")
        self.last_timings_ms = (st.prefill_ms, st.decode_ms)
        self.last_generate_stats = {"rows": B, "n_steps": int(st.n_steps), "decode_launches": int(st.decode_launches), "compactions": int(st.compactions),
                                    "final_rows": int(st.final_rows), "row_steps": int(st.row_steps)}
        ids = torch.frombuffer(out, dtype=torch.int32).clone().view(B, max_new_tokens)
        return ids, int(st.n_steps)
