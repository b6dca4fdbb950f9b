"""LLMSpeechTextInference: drop-in mirror of ref:inference.py:18-137 on the HIP path.

Same constructor signature, attributes (`config, device, audio_encoder, llm_type, llm_tokenizer,
prompt_prefix, prompt_suffix, llm`) and methods (`generate_llm_response`, `generate_text_response`,
`generate_audio_response`) with the reference's order of operations:
    audio -> AudioEncoder -> [text[1:] | audio] -> [prefix | . | suffix[1:]] -> greedy generate -> decode.
Optional keyword arguments (`tokenizer`, `llm`, `dtype`) let callers inject a tokenizer object or
already-built models where hub downloads are impossible; when omitted the reference's loading path
(local checkpoint directory named by `config.model.llm_type`) is followed.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch

from . import _lib as L
from .audio_encoder import AudioEncoder
from .audio_llama import AudioLlamaForCausalLM
from .utils import (LLAMA_PROMPT_PREFIX, LLAMA_PROMPT_SUFFIX, MINICHAT_PROMPT_PREFIX, MINICHAT_PROMPT_SUFFIX,
                    merge_prompt_tokens)


class LLMSpeechTextInference():
    def __init__(self, config, audio_encoder_checkpoint, device, *, tokenizer=None, llm: Optional[AudioLlamaForCausalLM] = None,
                 audio_encoder: Optional[AudioEncoder] = None, dtype: torch.dtype = torch.bfloat16):
        self.config = config
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.SpeechLLMError("LLMSpeechTextInference runs on the MI355X HIP path only (device must be cuda:N)")
        # libspeechllm launches on the CURRENT HIP device and stream: a `-g N` run must not execute cuda:N's pointers on GPU 0
        torch.cuda.set_device(self.device)

        # Audio encoder (ref:inference.py:24-28): flat state-dict checkpoint, strict load.
        if audio_encoder is None:
            checkpoint = audio_encoder_checkpoint
            if isinstance(checkpoint, str):
                checkpoint = torch.load(checkpoint, map_location="cpu")
            # load_pretrained=False: the checkpoint below overwrites every tensor the reference's constructor would have fetched
            audio_encoder = AudioEncoder(self.config, self.device, dtype=dtype, load_pretrained=False)
            audio_encoder.load_state_dict(checkpoint)
        self.audio_encoder = audio_encoder.eval().to(self.device)

        # Tokenizer (ref:inference.py:31-37).
        self.llm_type = self.config.model.llm_type
        if tokenizer is None:
            from transformers import AutoTokenizer  # host-side text processing only
            tokenizer = AutoTokenizer.from_pretrained(self.llm_type, use_fast=False, padding_side="left")
            tokenizer.pad_token = tokenizer.eos_token
        self.llm_tokenizer = tokenizer

        # ref:inference.py:39-44 tests the substring "llama" in the hub id; a local checkpoint directory is judged by its own
        # name, not by whatever its parent directories are called
        if "llama" in os.path.basename(os.path.normpath(self.llm_type)).lower() or (not os.path.isdir(self.llm_type) and "llama" in self.llm_type.lower()):
            self.prompt_prefix, self.prompt_suffix = LLAMA_PROMPT_PREFIX, LLAMA_PROMPT_SUFFIX
        else:
            self.prompt_prefix, self.prompt_suffix = MINICHAT_PROMPT_PREFIX, MINICHAT_PROMPT_SUFFIX

        # Frozen LLM (ref:inference.py:46-52).
        if llm is None:
            llm = AudioLlamaForCausalLM.from_pretrained(self.llm_type, use_cache=True, torch_dtype=dtype)
        self.llm = llm.eval().to(self.device)

    def generate_llm_response(self, inputs_embeds, max_new_tokens=256) -> List[str]:
        generate_ids = self.llm.generate(input_ids=None, inputs_embeds=inputs_embeds, max_new_tokens=max_new_tokens)
        self.last_generate_ids = generate_ids
        return self.llm_tokenizer.batch_decode(generate_ids, skip_special_tokens=True, clean_up_tokenization_spaces=True)

    def generate_text_response(self, input_text, max_new_tokens=256) -> str:
        # note the spaces around the user text (ref:inference.py:78)
        full_text_prompt = f"{self.prompt_prefix} {input_text}{self.prompt_suffix} "
        prompt_input_ids = self.llm_tokenizer(full_text_prompt, return_tensors='pt').input_ids.to(self.device)
        prompt_embeds = self.llm.model.embed_tokens(prompt_input_ids)
        return self.generate_llm_response(inputs_embeds=prompt_embeds, max_new_tokens=max_new_tokens)[0]

    def generate_audio_response(self, audio, additional_text_prompt="", max_new_tokens=256) -> str:
        audio_tensor = torch.as_tensor(audio, dtype=torch.float32).unsqueeze(0).to(self.device)
        if self.audio_encoder.downsample_method == "ctc_pool":
            # the reference calls an undefined self.get_ctc_pool_ranges here
            raise AttributeError("'LLMSpeechTextInference' object has no attribute 'get_ctc_pool_ranges'")
        if self.audio_encoder.encoder_base == "whisper":
            audio_embeds = self._whisper_audio_embeds(audio)
        else:
            audio_embeds = self.audio_encoder(audio_tensor, ctc_pool_ranges=None)

        if len(additional_text_prompt) > 0:  # text prompt goes before the audio, BOS dropped (ref:inference.py:113-122)
            additional_text_input_ids = self.llm_tokenizer(additional_text_prompt, return_tensors='pt').input_ids[:, 1:].to(self.device)
            text_embeds = self.llm.model.embed_tokens(additional_text_input_ids)
            combined_embeds = torch.cat([text_embeds, audio_embeds], dim=1)
        else:
            combined_embeds = audio_embeds

        prompt_emb_sequence = merge_prompt_tokens(inputs_embeds=combined_embeds, tokenizer=self.llm_tokenizer,
                                                  embed_tokens=self.llm.model.embed_tokens, llm_type=self.llm_type,
                                                  device=self.device)
        return self.generate_llm_response(prompt_emb_sequence, max_new_tokens)[0]

    def _whisper_audio_embeds(self, audio) -> torch.Tensor:
        """Whisper base (ref:config/llama3_whisper.yaml).  ref:inference.py:97-107 hands the raw waveform to the Whisper encoder,
        which cannot run; the working order of operations is the trainer's (ref:trainer.py:168-199, 278-291)
        and is what this follows: log-mel of the utterance padded / cut to the 30 s window (sl_whisper_logmel) -> encoder ->
        pool -> projector (sl_whisper_forward) -> crop to compute_num_audio_embeds(n_samples) rows."""
        from .utils import compute_num_audio_embeds
        wave = torch.as_tensor(audio, dtype=torch.float32).reshape(-1)
        sr = int(getattr(getattr(self.config, "audio", None), "sampling_rate", 16000) or 16000)
        feats = self.audio_encoder.feature_extractor([wave], return_tensors="pt", sampling_rate=sr).input_features
        padded = self.audio_encoder(feats)                                    # (1, P_window, llm_dim)
        keep = max(0, min(int(padded.shape[1]), compute_num_audio_embeds(int(wave.numel()), sr=sr)))
        return padded[:, :keep]

    def generate_audio_responses(self, audios, additional_text_prompts=None, max_new_tokens=256) -> List[str]:
        """Batched form of generate_audio_response (an extension: the reference answers one utterance per call).
        Every utterance is encoded and prefilled at its own length in ONE ragged batch — the encoder writes its embeddings
        straight into the packed prompt buffer `[prefix | text[1:] | audio | suffix[1:]]` — and decoded together, so that the
        weights are streamed once per step for the whole batch; rows that stop early leave the batch (sl_generate's compaction).
        Both encoder bases take this path (Whisper: log-mel of all windows -> one encoder pass -> each utterance's rows cropped to
        compute_num_audio_embeds, the trainer's order of operations ref:trainer.py:168-199,278-291); more utterances than one
        generate call takes (SL_MAX_DECODE_BATCH) are answered in consecutive chunks.  Results equal per-utterance calls (tests)."""
        n = len(audios)
        texts = list(additional_text_prompts) if additional_text_prompts is not None else [""] * n
        if len(texts) != n:
            raise ValueError(f"{len(texts)} text prompts for {n} utterances")
        if self.audio_encoder.downsample_method != "pool":       # stack / ctc_pool: the one-utterance path (ctc_pool raises as the reference does)
            return [self.generate_audio_response(a, texts[i], max_new_tokens) for i, a in enumerate(audios)]
        out: List[str] = []
        ids_all = []
        for lo_ in range(0, n, L.MAX_DECODE_BATCH):
            ids = self._generate_chunk(audios[lo_:lo_ + L.MAX_DECODE_BATCH], texts[lo_:lo_ + L.MAX_DECODE_BATCH], max_new_tokens)
            ids_all.append(ids)
            out += self.llm_tokenizer.batch_decode(ids, skip_special_tokens=True, clean_up_tokenization_spaces=True)
        # one (n, widest chunk) id matrix, pad-filled like a single HF call over the whole list would leave it
        width = max(int(i.shape[1]) for i in ids_all) if ids_all else 0
        pad = self.llm.generation_config.pad_token_id
        if pad is None:
            e = self.llm.generation_config.eos_token_id
            pad = (e[0] if isinstance(e, (list, tuple)) else e) if e else 0
        self.last_generate_ids = torch.cat([torch.nn.functional.pad(i, (0, width - int(i.shape[1])), value=int(pad)) for i in ids_all]) if ids_all else None
        return out

    def _generate_chunk(self, audios, texts, max_new_tokens) -> torch.Tensor:
        """One generate call's worth of utterances -> LongTensor (n, n_cols) of new tokens."""
        from .utils import compute_num_audio_embeds, prompt_template
        n = len(audios)
        emb = self.llm.model.embed_tokens
        prefix, suffix = prompt_template(self.llm_type)
        pre_e = emb(self.llm_tokenizer(prefix, return_tensors="pt").input_ids.to(self.device))[0]
        suf_e = emb(self.llm_tokenizer(suffix, return_tensors="pt").input_ids.to(self.device))[0, 1:]
        txt_e = [emb(self.llm_tokenizer(t_, return_tensors="pt").input_ids[:, 1:].to(self.device))[0] if len(t_) > 0 else None for t_ in texts]
        waves = [torch.as_tensor(a, dtype=torch.float32).reshape(-1) for a in audios]
        enc = self.audio_encoder
        whisper = enc.encoder_base == "whisper"
        if whisper:
            sr = int(getattr(getattr(self.config, "audio", None), "sampling_rate", 16000) or 16000)
            feats = enc.feature_extractor(waves, return_tensors="pt", sampling_rate=sr).input_features
            padded = enc(feats)                                                  # (n, P_window, llm_dim): every 30 s window in one pass
            Ps = [max(0, min(int(padded.shape[1]), compute_num_audio_embeds(int(w.numel()), sr=sr))) for w in waves]
        else:
            Ps = [(enc.arch.num_frames(int(w.numel())) - enc.pool_kernel) // enc.pool_stride + 1 for w in waves]
        lens, heads = [], []
        for i in range(n):
            n_head = pre_e.shape[0] + (txt_e[i].shape[0] if txt_e[i] is not None else 0)
            heads.append(n_head)
            lens.append(n_head + Ps[i] + suf_e.shape[0])
        starts = [0]
        for ln in lens:
            starts.append(starts[-1] + ln)
        x = torch.empty((starts[-1], pre_e.shape[1]), device=self.device, dtype=self.llm.dtype)
        for i in range(n):
            o = starts[i]
            x[o:o + pre_e.shape[0]] = pre_e
            if txt_e[i] is not None:
                x[o + pre_e.shape[0]:o + heads[i]] = txt_e[i]
            if whisper:
                x[o + heads[i]:o + heads[i] + Ps[i]] = padded[i, :Ps[i]]
            x[o + heads[i] + Ps[i]:starts[i + 1]] = suf_e
        if not whisper:
            enc.encode_packed(waves, out=x, out_row_offsets=[starts[i] + heads[i] for i in range(n)])
        # every prompt opens with the same template rows (and the same instruction text when the caller gave one text for all):
        # the batched decode attention reads those cache positions once for the batch (sl_kv_cache.shared_prefix)
        shared = int(pre_e.shape[0])
        if txt_e[0] is not None and all(t_ == texts[0] for t_ in texts):
            shared += int(txt_e[0].shape[0])
        ids, n_cols = self.llm.generate_packed(x, lens, max_new_tokens, use_eos=True, shared_prefix=shared)
        return ids[:, :n_cols].to(torch.int64)
