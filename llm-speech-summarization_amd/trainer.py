"""Trainer: host-side mirror of ref:trainer.py:23-545 around the HIP KD step, lifted to data parallel.

Kept from the reference: constructor `Trainer(args, config, device)` (args: run_name, checkpoint_path, gpu_idx),
`train()`, `validate(epoch)`, `load_checkpoint(path)`, the two collate functions, the dataset schema
(`audio{array}`, `text`, `text_input_ids`, `response_input_ids` (nested [0]), `pool_ranges_4`; ref:trainer.py:134-199),
the checkpoint format `{"audio_encoder","optimizer","lr_scheduler","epoch","step"}` under
`checkpoints/<run>/epoch_{e}_step_{s}.pt` (ref:trainer.py:516-528) and the hyper-parameters read from the yaml.
New: one process per GPU; every accumulation window of `grad_accum_interval` samples of a seeded per-epoch shuffle is dealt
to the ranks (`window[rank::world]`), each rank processes its share as one packed micro-batch (`KDTrainer.micro_batch`);
validation is sharded over the ranks the same way (NLL sums all-reduced); rank 0 logs (JSON lines instead of TensorBoard),
generates the sample responses and writes the checkpoint, and every rank waits for it at a barrier.
Datasets may be passed in as python sequences (tests, synthetic runs) instead of `datasets.load_from_disk` paths.
"""
from __future__ import annotations

import json
import math
import os
import random
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L
from .audio_encoder import AudioEncoder
from .audio_llama import AudioLlamaForCausalLM
from .training import KDTrainer, TrainRegularizers
from .utils import compute_num_audio_embeds, merge_prompt_tokens, prompt_template
from .weights import rename_weight_norm_keys


class Trainer():
    def __init__(self, args, config, device, *, tokenizer=None, llm: Optional[AudioLlamaForCausalLM] = None,
                 audio_encoder: Optional[AudioEncoder] = None, train_dataset: Optional[Sequence[dict]] = None,
                 val_dataset: Optional[Sequence[dict]] = None, dtype: torch.dtype = torch.bfloat16) -> None:
        self.args, self.config = args, config
        self.run_name = args.run_name
        self.device = torch.device(device)
        if self.device.type == "cuda":
            # libspeechllm launches on the CURRENT HIP device and stream: make the trainer's device current before anything is
            # allocated (the reference's `-g N` CLI on a multi-GPU node would otherwise run cuda:N's pointers on GPU 0)
            torch.cuda.set_device(self.device)
        import torch.distributed as dist
        self.dist = dist if dist.is_initialized() else None
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        seed = int(config.seed_everything) + self.rank      # the reference seeds only the CUDA RNG
        torch.manual_seed(seed); random.seed(seed); np.random.seed(seed % (2 ** 32))     # numpy: SpecAugment spans (hf _compute_mask_indices)
        self.checkpoint_save_dir = os.path.join(config.log.checkpoint_dir, self.run_name)
        self.log_dir = os.path.join(config.log.log_dir, self.run_name)
        if self.rank == 0:
            os.makedirs(self.checkpoint_save_dir, exist_ok=True)
            os.makedirs(self.log_dir, exist_ok=True)
        self.encoder_base = config.model.audio_encoder.base
        # ref:trainer.py:44-46 + ref:model/audio_encoder.py:6-13,34-52: a run without `-p` starts from the PRETRAINED encoder named by
        # config.model.audio_encoder.type and a freshly initialised embed_projection (drawn from config.seed_everything alone, so
        # every rank starts from the same weights although the ranks' own RNG streams differ)
        self.audio_encoder = audio_encoder or AudioEncoder(config, self.device, dtype=dtype)
        self.llm_type = config.model.llm_type
        if tokenizer is None:
            from transformers import AutoTokenizer
            tokenizer = AutoTokenizer.from_pretrained(self.llm_type, use_fast=False, padding_side="left")
            tokenizer.pad_token = tokenizer.eos_token
        self.tokenizer = tokenizer
        self.llm = (llm or AudioLlamaForCausalLM.from_pretrained(self.llm_type, use_cache=True, torch_dtype=dtype)).eval().to(self.device)
        self.train_dataset, self.val_dataset = train_dataset, val_dataset
        if self.train_dataset is None:
            self.get_dataloaders()
        self.step, self.start_epoch = 0, 0
        # samples per optimizer step over all ranks: grad_accum_interval (reference-equivalent strong scaling) or, with the new key
        # train.per_rank_accum = k, k x world (weak scaling: every rank keeps k samples per step) — training.effective_accum
        from .training import effective_accum
        self.grad_accum_interval = effective_accum(config.train, self.world)[0]
        self.num_epochs = int(config.train.epochs)
        prefix, suffix = prompt_template(self.llm_type)
        self.prefix_ids = self.tokenizer(prefix, return_tensors="pt").input_ids
        self.suffix_ids = self.tokenizer(suffix, return_tensors="pt").input_ids
        total_iters = self.num_epochs * len(self.train_dataset) // self.grad_accum_interval     # ref:trainer.py:106-110 (global steps)
        # ref:trainer.py:258 puts the encoder in train() mode: HF's dropouts, LayerDrop and SpecAugment are active during the
        # optimisation step (validate() runs the inference path, i.e. eval mode, ref:trainer.py:402)
        reg = None if getattr(args, "no_regularizers", False) else self._regularizers(seed)
        self.kd = KDTrainer(config, self.audio_encoder.to(self.device), self.llm, self.prefix_ids, self.suffix_ids,
                            total_optimizer_steps=max(1, total_iters), regularizers=reg)
        self.optimizer, self.lr_scheduler = self.kd.optimizer, self.kd.scheduler
        if getattr(self.args, "checkpoint_path", None):
            self.load_checkpoint(self.args.checkpoint_path)
        self._sync_masters()

    def _regularizers(self, seed: int) -> TrainRegularizers:
        """The training-mode regularisers HF's module would apply in train() mode (ref:trainer.py:258): read from the encoder's own
        config.json when the encoder came from a local checkpoint directory, else the published values of the two encoders the
        shipped configs name (hubert-large-ls960-ft; whisper-medium, which has every dropout at 0 and no SpecAugment)."""
        from .weights import resolve_pretrained_dir
        local = resolve_pretrained_dir(self.config.model.audio_encoder.type)
        d = None
        if local and os.path.exists(os.path.join(local, "config.json")):
            with open(os.path.join(local, "config.json")) as f:
                d = json.load(f)
        if self.encoder_base == "whisper":
            return TrainRegularizers.from_whisper_config(d or {}, seed=seed)
        return TrainRegularizers.from_hf_config(d, seed=seed) if d is not None else TrainRegularizers(seed=seed)

    def _sync_masters(self) -> None:
        """Data parallel: every rank applies the same all-reduced gradient, so every rank must START from the same weights.  The
        cold start and the checkpoint are rank-independent by construction; the broadcast from rank 0 makes it unconditional
        (318 M fp32 once per run)."""
        if self.dist is None or self.world == 1:
            return
        cpu_group = self.dist.get_backend() != "nccl"
        for k in self.kd.param_names:
            t = self.kd.master[k]
            buf = t.cpu() if cpu_group else t
            self.dist.broadcast(buf, src=0)
            if cpu_group:
                t.copy_(buf)
        self.audio_encoder.refresh_weights(self.kd.master)

    # -- checkpoints (ref:trainer.py:116-132, 516-528) ---------------------------------------------
    def load_checkpoint(self, checkpoint_path):
        """Reads checkpoints in the reference's format — written by this build or by ref:trainer.py:516-528: the encoder
        state-dict under either weight-norm spelling, the optimizer state in the reference's two-param-group layout."""
        ck = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
        enc_sd = rename_weight_norm_keys(ck["audio_encoder"], self.kd.master)
        missing = [k for k in self.kd.master if k not in enc_sd]
        if missing:
            raise L.SpeechLLMError(f"checkpoint lacks encoder parameters {missing[:4]}{'...' if len(missing) > 4 else ''}")
        for k, dst in self.kd.master.items():
            dst.copy_(enc_sd[k].to(dst.device, torch.float32))
        self.audio_encoder.refresh_weights(self.kd.master)
        self.kd.load_optimizer_state_dict(ck["optimizer"])
        self.lr_scheduler.load_state_dict(ck["lr_scheduler"])
        self.start_epoch, self.step = ck["epoch"], ck["step"]
        extra = ck.get("kd_state")           # this build's addition: regulariser counters + host RNGs, so a resumed run draws on
        if extra is not None:                # (the reference restarts its masks from scratch: SURVEY §9 Q9)
            self.kd.micro_batches = int(extra["micro_batches"])
            self.kd.micro_total = int(extra.get("micro_total", 0))
            per_rank = extra.get("rng", {}).get(self.rank)
            if per_rank is not None:
                np.random.set_state(per_rank["numpy"]); random.setstate(per_rank["python"]); torch.set_rng_state(per_rank["torch"])
        if self.rank == 0:
            print(f"Loaded checkpoint from {checkpoint_path}.\n")

    def _rng_states(self) -> dict:
        mine = {"numpy": np.random.get_state(), "python": random.getstate(), "torch": torch.get_rng_state()}
        if self.dist is None:
            return {0: mine}
        gathered = [None] * self.world
        self.dist.all_gather_object(gathered, mine)
        return dict(enumerate(gathered))

    def save_checkpoint(self, epoch, rng: Optional[dict] = None) -> str:
        save_path = os.path.join(self.checkpoint_save_dir, f"epoch_{epoch}_step_{self.step}.pt")
        torch.save({"audio_encoder": {k: v.detach().cpu() for k, v in self.kd.master.items()},
                    "optimizer": self.kd.optimizer_state_dict(), "lr_scheduler": self.lr_scheduler.state_dict(),
                    "epoch": epoch, "step": self.step,
                    "kd_state": {"micro_batches": self.kd.micro_batches, "micro_total": self.kd.micro_total, "rng": rng or {}}}, save_path)
        return save_path

    # -- data (ref:trainer.py:134-248) --------------------------------------------------------------
    def collate_audio_batch_hubert(self, data):
        raw_audios = [torch.as_tensor(x['audio']['array']) for x in data]
        audio_len_samples = [len(a) for a in raw_audios]
        max_len = max(audio_len_samples)
        padded = torch.stack([torch.nn.functional.pad(a, (0, max_len - len(a))) for a in raw_audios], dim=0).float()
        text_input_ids = [torch.as_tensor(x['text_input_ids'])[1:] for x in data]               # strip BOS (ref:trainer.py:155)
        response_input_ids = [torch.as_tensor(x['response_input_ids'][0])[1:] for x in data]    # nested [0] + strip BOS (:156)
        return (raw_audios, padded, audio_len_samples, [x['text'] for x in data], text_input_ids, response_input_ids,
                [x.get('pool_ranges_4') for x in data])

    def collate_audio_batch_whisper(self, data):
        raw_audios = [torch.as_tensor(x['audio']['array']).numpy() for x in data]
        feats = self.audio_encoder.feature_extractor(raw_audios, return_tensors="pt", sampling_rate=self.config.audio.sampling_rate).input_features
        text_input_ids = [torch.as_tensor(x['text_input_ids'])[1:] for x in data]
        response_input_ids = [torch.as_tensor(x['response_input_ids'][0])[1:] for x in data]
        return (raw_audios, feats, [len(a) for a in raw_audios], [x['text'] for x in data], text_input_ids, response_input_ids,
                [x.get('pool_ranges_4') for x in data])

    def get_dataloaders(self):
        from datasets import concatenate_datasets, load_from_disk

        def load(names):
            parts = []
            for name in names:
                ds = load_from_disk(os.path.join(self.config.data.base_path, name))
                ds.set_format(type='torch')
                parts.append(ds)
            return concatenate_datasets(parts)

        self.train_dataset, self.val_dataset = load(self.config.data.train_set), load(self.config.data.val_set)

    def _epoch_windows(self, epoch: int) -> List[List[int]]:
        """The epoch's accumulation windows (seeded shuffle shared by all ranks, cut every grad_accum_interval samples; the last
        one may be partial — the reference steps on the last batch too, ref:trainer.py:377)."""
        g = torch.Generator().manual_seed(int(self.config.seed_everything) + epoch)
        perm = torch.randperm(len(self.train_dataset), generator=g).tolist()
        a = self.grad_accum_interval
        return [perm[i:i + a] for i in range(0, len(perm), a)]

    def _epoch_indices(self, epoch: int) -> List[int]:
        """This rank's samples of the epoch, in processing order: every window is dealt round-robin to the ranks."""
        return [i for w in self._epoch_windows(epoch) for i in w[self.rank::self.world]]

    # -- training loop (ref:trainer.py:250-398) ------------------------------------------------------
    def log(self, **kv):
        if self.rank == 0:
            line = json.dumps(kv)
            print(line, flush=True)
            with open(os.path.join(self.log_dir, "metrics.jsonl"), "a") as f:
                f.write(line + "\n")

    @staticmethod
    def _crossed(prev: int, now: int, interval: int) -> bool:
        """True when the micro-step counter passed a multiple of `interval` in (prev, now]: the reference tests
        `step % interval == 0` after every micro-step (ref:trainer.py:389-395); here the counter advances a window at a time."""
        return interval > 0 and prev // interval != now // interval

    def train(self):
        for epoch in range(self.start_epoch, self.start_epoch + self.num_epochs):       # resume quirk kept
            for window in self._epoch_windows(epoch):
                mine = window[self.rank::self.world]
                tail = len(window) < self.grad_accum_interval
                losses = []
                if mine:
                    batch = [self.train_dataset[i] for i in mine]
                    if self.encoder_base == "whisper":      # ref:trainer.py:168-199: the tape computes the log-mel itself from the raw audio
                        raw, _, _, _, text_ids, resp_ids, _ = self.collate_audio_batch_whisper(batch)
                    else:
                        raw, _, _, _, text_ids, resp_ids, _ = self.collate_audio_batch_hubert(batch)
                    losses = self.kd.micro_batch(raw, text_ids, resp_ids, close_window=True if tail else None)
                else:
                    self.kd.close_window()                   # tail window with fewer samples than ranks: join the exchange
                prev, self.step = self.step, self.step + len(window)              # global micro-steps, like the reference's counter
                if losses and self._crossed(prev, self.step, int(self.config.log.log_interval)):
                    mean = {k: sum(l[k] for l in losses) / len(losses) for k in losses[0]}
                    self.log(step=self.step, epoch=epoch, lr=self.lr_scheduler.get_last_lr()[0], **{f"train/{k}": v for k, v in mean.items()})
                if self._crossed(prev, self.step, int(self.config.log.validation_interval)):
                    self.validate(epoch)
            self.validate(epoch)

    def close(self) -> None:
        """End of the run: every rank tears down the gradient exchange's communicator together (train.py calls it before the
        process group goes away)."""
        self.kd.close()

    def abort(self) -> None:
        """Failure path (train.py, on any exception / KeyboardInterrupt): local tear-down only — the peers may be inside a collective
        this rank will never join, so nothing here waits for them; the caller re-raises and the process exits non-zero."""
        self.kd.abort()

    # -- validation (ref:trainer.py:400-528) ---------------------------------------------------------
    def _val_sample(self, sample_idx):
        """One validation sample -> (audio_embeds (1,P,H), text ids, response ids, text)."""
        if self.encoder_base == "whisper":   # ref:trainer.py:417-419 feeds the padded 30 s window and does NOT crop here
            _, feats, _, texts, text_ids, resp_ids, _ = self.collate_audio_batch_whisper([self.val_dataset[sample_idx]])
            audio_embeds = self.audio_encoder(feats.to(self.device))
        else:
            raw, _, _, texts, text_ids, resp_ids, _ = self.collate_audio_batch_hubert([self.val_dataset[sample_idx]])
            audio_embeds = self.audio_encoder(raw[0][None].to(self.device))
        return audio_embeds, text_ids[0].to(self.device), resp_ids[0].to(self.device), texts[0]

    @torch.no_grad()
    def validate(self, epoch):
        """ref:trainer.py:400-528.  Every rank evaluates `val[rank::world]`; the NLL sums are all-reduced, so each rank holds
        the same perplexities as a single-process run over the whole set; rank 0 generates the sample responses and saves the
        checkpoint while the others wait at the closing barrier (bounded work: num_generate_samples generations)."""
        emb = self.llm.model.embed_tokens
        pre, suf = emb(self.prefix_ids.to(self.device)), emb(self.suffix_ids.to(self.device))[:, 1:]
        sums = torch.zeros(3, dtype=torch.float64)
        for sample_idx in range(self.rank, len(self.val_dataset), self.world):
            audio_embeds, text_ids, resp, _ = self._val_sample(sample_idx)
            r = emb(resp[None])[:, 1:]
            a_seq = torch.cat([pre, audio_embeds, suf, r], dim=1)
            t_seq = torch.cat([pre, emb(text_ids[None]), suf, r], dim=1)
            sums[0] += float(self.llm(inputs_embeds=a_seq, labels=[resp]).loss)
            sums[1] += float(self.llm(inputs_embeds=t_seq, labels=[resp]).loss)
            sums[2] += 1
        rng = self._rng_states()
        if self.dist is not None:
            red = sums.to(self.device) if self.dist.get_backend() == "nccl" else sums
            self.dist.all_reduce(red, op=self.dist.ReduceOp.SUM)
            sums = red.cpu()
        n = max(1.0, float(sums[2]))
        out = dict(step=self.step, epoch=epoch, **{"validation/audio_perplexity": math.exp(float(sums[0]) / n),
                                                   "validation/text_perplexity": math.exp(float(sums[1]) / n)})
        if self.rank == 0:
            samples = []
            for sample_idx in range(min(int(self.config.log.num_generate_samples), len(self.val_dataset))):
                audio_embeds, text_ids, _, text = self._val_sample(sample_idx)
                a_prompt = merge_prompt_tokens(audio_embeds, self.tokenizer, emb, self.llm_type, self.device)
                t_prompt = merge_prompt_tokens(emb(text_ids[None]), self.tokenizer, emb, self.llm_type, self.device)
                n_in = audio_embeds.shape[1]                                                # same budget for both prompts
                samples.append(dict(text=text, audio_response=self.generate_llm_response(a_prompt, n_in)[0],
                                    text_response=self.generate_llm_response(t_prompt, n_in)[0]))
            self.log(**out, samples=samples[:2])
            path = self.save_checkpoint(epoch, rng)
            print(f"Saved checkpoint for epoch {epoch} to {path}.\n")
        if self.dist is not None:
            self.dist.barrier()
        return out

    def generate_llm_response(self, inputs_embeds, len_inputs=60):
        generate_ids = self.llm.generate(input_ids=None, inputs_embeds=inputs_embeds, max_new_tokens=2 * len_inputs)
        return self.tokenizer.batch_decode(generate_ids, skip_special_tokens=True, clean_up_tokenization_spaces=True)
