"""Cold start of the audio encoder as the reference does it (ref:model/audio_encoder.py:6-13,34-52, ref:trainer.py:44-46):
`AudioEncoder(config, device)` finds the PRETRAINED encoder inside an HF checkpoint directory (head-model files: `hubert.` prefix
+ `lm_head` for HuBERT, `model.encoder.` + decoder for Whisper; safetensors or pytorch_model.bin) and draws `embed_projection`
like nn.Linear does; `train.py` then runs with no `-p` and nothing injected.  Fixtures `coldstart_*.npz` come from the reference
AudioEncoder built with `from_pretrained(<dir>)` (oracle/gen_golden.py coldstart); `whisper_wide.npz` is the Whisper-medium-width
case (BASELINE configs[3])."""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

from conftest import GOLDEN, REPO, golden, pkg, rel_err, t
from hf_dirs import write_hubert_ctc_dir, write_kd_dataset, write_whisper_gen_dir
from oracle import hubert_oracle as ho
from oracle import whisper_oracle as wo
from oracle.golden_cfgs import TINY_HUBERT, TINY_LLAMA, TINY_WHISPER, WIDE_WHISPER

ri = pkg("random_init")
cfgm = pkg("config")
weights = pkg("weights")

DEV = "cuda:0"


def enc_conf(base, type_str, llm_dim, seed=1234):
    return cfgm.from_dict(dict(seed_everything=seed, audio=dict(sampling_rate=16000),
                               model=dict(audio_encoder=dict(base=base, type=str(type_str), downsample_method="pool", downsample_factor=4,
                                                             pooling=dict(kernel_size=8, stride=4)),
                                          llm_embedding_channels=llm_dim, llm_type="meta-llama/Llama-3.2-3B-Instruct")))


# ------------------------------------------------------------------------------------------------------------------
# CPU: the loader's host logic and the oracle against the new fixtures
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fmt,wn", [("safetensors", "legacy"), ("bin", "parametrizations")])
def test_pretrained_hubert_loader_strips_ctc_prefix_and_head(tmp_path, fmt, wn):
    sd = ri.hubert_encoder_state_dict(TINY_HUBERT, 256, seed=81, weight_norm_keys=wn)
    write_hubert_ctc_dir(tmp_path / "h", TINY_HUBERT, sd, fmt=fmt)
    got = weights.pretrained_encoder_state_dict(str(tmp_path / "h"), "hubert")
    want = {k: v for k, v in sd.items() if k.startswith("encoder.")}
    assert set(got) == set(want)                                # no lm_head, no embed_projection
    assert all(torch.equal(got[k], want[k]) for k in want)
    # a bare HubertModel directory (no prefix) reads the same
    from safetensors.torch import save_file
    os.makedirs(tmp_path / "bare")
    save_file({k[len("encoder."):]: v.contiguous() for k, v in want.items()}, str(tmp_path / "bare" / "model.safetensors"))
    bare = weights.pretrained_encoder_state_dict(str(tmp_path / "bare"), "hubert")
    assert set(bare) == set(want) and all(torch.equal(bare[k], want[k]) for k in want)
    # a directory with a config but no weight files is "no checkpoint", not an error
    os.makedirs(tmp_path / "empty")
    assert weights.pretrained_encoder_state_dict(str(tmp_path / "empty"), "hubert") is None


def test_pretrained_whisper_loader_takes_the_encoder_of_a_generation_checkpoint(tmp_path):
    sd = ri.whisper_encoder_state_dict(TINY_WHISPER, 256, seed=82)
    write_whisper_gen_dir(tmp_path / "w", TINY_WHISPER, sd)
    got = weights.pretrained_encoder_state_dict(str(tmp_path / "w"), "whisper")
    want = {k: v for k, v in sd.items() if k.startswith("encoder.")}
    assert set(got) == set(want) and all(torch.equal(got[k], want[k]) for k in want)
    with pytest.raises(pkg("_lib").SpeechLLMError):
        weights.pretrained_encoder_state_dict(str(tmp_path / "w"), "hubert")      # wrong family: says so instead of loading nothing


def test_embed_projection_init_is_nn_linear_default_and_rank_independent():
    a = weights.init_embed_projection(1024, 3072, seed=1234)
    b = weights.init_embed_projection(1024, 3072, seed=1234)
    c = weights.init_embed_projection(1024, 3072, seed=1235)
    w, bias = a["embed_projection.weight"], a["embed_projection.bias"]
    assert w.shape == (3072, 1024) and bias.shape == (3072,)
    assert torch.equal(w, b["embed_projection.weight"]) and not torch.equal(w, c["embed_projection.weight"])
    bound = 1 / math.sqrt(1024)
    assert float(w.abs().max()) <= bound and float(bias.abs().max()) <= bound
    # U(-b, b): std = b / sqrt(3); nn.Linear(1024, 3072) itself, for the record of what "default" means
    assert abs(float(w.std()) - bound / math.sqrt(3)) < 0.01 * bound
    lin = torch.nn.Linear(1024, 3072)
    assert abs(float(lin.weight.std()) - float(w.std())) < 0.01 * bound and float(lin.weight.abs().max()) <= bound
    # the global RNG is neither read nor advanced (ranks seed it differently: seed_everything + rank)
    torch.manual_seed(7)
    s0 = torch.get_rng_state()
    weights.init_embed_projection(64, 64, seed=1)
    assert torch.equal(torch.get_rng_state(), s0)


def test_hub_id_resolves_through_the_local_hf_cache(tmp_path, monkeypatch):
    snap = tmp_path / "hub" / "models--facebook--hubert-large-ls960-ft" / "snapshots" / "abc123"
    os.makedirs(snap)
    json.dump({"model_type": "hubert"}, open(snap / "config.json", "w"))
    monkeypatch.setenv("HF_HUB_CACHE", str(tmp_path / "hub"))
    assert weights.resolve_pretrained_dir("facebook/hubert-large-ls960-ft") == str(snap)
    assert weights.resolve_pretrained_dir("facebook/not-there") is None
    assert weights.resolve_pretrained_dir(str(tmp_path)) == str(tmp_path)


def test_oracle_vs_reference_cold_start_fixtures():
    g = golden("coldstart_hubert")
    sd = ri.hubert_encoder_state_dict(TINY_HUBERT, 256, seed=int(g["weight_seed"]))
    sd["embed_projection.weight"], sd["embed_projection.bias"] = t(g["proj_w"]), t(g["proj_b"])
    wave = ri.synthetic_waveform(int(g["n_samples"]), seed=int(g["wave_seed"]))
    assert rel_err(ho.audio_encoder_forward(sd, TINY_HUBERT, wave[None]), t(g["audio_embeds"])) < 2e-5
    b = 1 / math.sqrt(TINY_HUBERT.hidden_size)       # what the reference's nn.Linear drew obeys the bound the product's init uses
    assert float(t(g["proj_w"]).abs().max()) <= b and float(t(g["proj_b"]).abs().max()) <= b
    g = golden("coldstart_whisper")
    sd = ri.whisper_encoder_state_dict(TINY_WHISPER, 256, seed=int(g["weight_seed"]))
    sd["embed_projection.weight"], sd["embed_projection.bias"] = t(g["proj_w"]), t(g["proj_b"])
    feats = wo.log_mel(TINY_WHISPER, ri.synthetic_waveform(int(g["n_samples"]), seed=int(g["wave_seed"])))[None]
    assert float((feats - t(g["input_features"])).abs().max()) < 1e-5
    assert rel_err(wo.audio_encoder_forward(sd, TINY_WHISPER, feats), t(g["audio_embeds"])) < 2e-5


def test_oracle_vs_reference_whisper_medium_width_fixture():
    g = golden("whisper_wide")
    sd = ri.whisper_encoder_state_dict(WIDE_WHISPER, 3072, seed=int(g["weight_seed"]))
    feats = wo.log_mel(WIDE_WHISPER, ri.synthetic_waveform(int(g["n_samples"]), seed=int(g["wave_seed"])))[None]
    assert float((feats - t(g["input_features"])).abs().max()) < 1e-5
    out = wo.audio_encoder_forward(sd, WIDE_WHISPER, feats)
    assert out.shape[1] == int(g["P"])
    assert rel_err(out[:, ::4], t(g["audio_embeds_every4"])) < 2e-5
    assert rel_err(wo.whisper_encoder_forward(sd, WIDE_WHISPER, feats)[:, ::24], t(g["last_hidden_rows"])) < 2e-5


# ------------------------------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("fmt,wn", [("safetensors", "legacy"), ("bin", "parametrizations")])
def test_cold_start_hubert_from_ctc_checkpoint_dir_vs_reference_fixture(tmp_path, fmt, wn):
    """`AudioEncoder(config, device)` and nothing else: encoder tensors out of HubertForCTC files, embed_projection drawn from
    config.seed_everything.  Hidden states in front of the projection equal the reference's (which loaded the same directory
    through AutoModel.from_pretrained); with the reference's own projection draw put in, the embeddings do too."""
    enc_mod = pkg("audio_encoder")
    g = golden("coldstart_hubert")
    sd = ri.hubert_encoder_state_dict(TINY_HUBERT, 256, seed=int(g["weight_seed"]), weight_norm_keys=wn)
    write_hubert_ctc_dir(tmp_path / "hubert-tiny-ft", TINY_HUBERT, sd, fmt=fmt)
    conf = enc_conf("hubert", tmp_path / "hubert-tiny-ft", 256)
    enc = enc_mod.AudioEncoder(conf, DEV, dtype=torch.float32).eval()
    assert enc.pretrained_from == str(tmp_path / "hubert-tiny-ft") and enc.weights is not None
    state = enc.state_dict()
    assert all(torch.equal(state[k].cpu(), v) for k, v in sd.items() if k.startswith("encoder."))
    same_seed = weights.init_embed_projection(TINY_HUBERT.hidden_size, 256, 1234)
    assert torch.equal(state["embed_projection.weight"].cpu(), same_seed["embed_projection.weight"])
    wave = ri.synthetic_waveform(int(g["n_samples"]), seed=int(g["wave_seed"]))
    _, _, hidden, _ = enc.encode_packed([wave.to(DEV)], want_last_hidden=True)
    assert rel_err(hidden.float().cpu(), t(g["last_hidden_state"])[0]) < 1e-4
    state["embed_projection.weight"], state["embed_projection.bias"] = t(g["proj_w"]), t(g["proj_b"])
    enc.load_state_dict(state)
    assert rel_err(enc(wave[None].to(DEV)).float().cpu(), t(g["audio_embeds"])) < 1e-4


@pytest.mark.gpu
def test_cold_start_whisper_from_generation_checkpoint_dir_vs_reference_fixture(tmp_path):
    enc_mod = pkg("audio_encoder")
    g = golden("coldstart_whisper")
    sd = ri.whisper_encoder_state_dict(TINY_WHISPER, 256, seed=int(g["weight_seed"]))
    write_whisper_gen_dir(tmp_path / "whisper-tiny", TINY_WHISPER, sd)
    enc = enc_mod.AudioEncoder(enc_conf("whisper", tmp_path / "whisper-tiny", 256), DEV, dtype=torch.float32).eval()
    assert enc.pretrained_from is not None and enc.arch.d_model == TINY_WHISPER.d_model
    state = enc.state_dict()
    state["embed_projection.weight"], state["embed_projection.bias"] = t(g["proj_w"]), t(g["proj_b"])
    enc.load_state_dict(state)
    wave = ri.synthetic_waveform(int(g["n_samples"]), seed=int(g["wave_seed"])).numpy()
    feats = enc.feature_extractor([wave], return_tensors="pt", sampling_rate=16000).input_features
    assert float((feats.cpu() - t(g["input_features"])).abs().max()) < 2e-4
    assert rel_err(enc(feats).float().cpu(), t(g["audio_embeds"])) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2)])
def test_whisper_medium_width_two_layers_vs_reference_fixture(dtype, tol):
    """BASELINE configs[3] at Whisper-medium width: conv1 80 -> 1024, the 1 500-row position table, 16 heads, FFN 4096, llm_dim
    3072 — log-mel on the GPU, encoder, pool, projector against the reference AudioEncoder + HF feature extractor."""
    enc_mod = pkg("audio_encoder")
    g = golden("whisper_wide")
    WC = WIDE_WHISPER
    arch = weights.WhisperArch(WC.d_model, WC.encoder_layers, WC.encoder_attention_heads, WC.encoder_ffn_dim, WC.num_mel_bins, WC.max_source_positions)
    enc = enc_mod.AudioEncoder(enc_conf("whisper", "synthetic", 3072), DEV, dtype=dtype, arch=arch)
    sd = ri.whisper_encoder_state_dict(WC, 3072, seed=int(g["weight_seed"]))
    if dtype == torch.bfloat16:          # same bf16-rounded weights on both sides; the fixture is fp32 weights, so widen by the rounding
        tol = 4e-2
    enc.load_state_dict(sd).eval().to(DEV)
    wave = ri.synthetic_waveform(int(g["n_samples"]), seed=int(g["wave_seed"])).numpy()
    feats = enc.feature_extractor([wave], return_tensors="pt", sampling_rate=16000).input_features
    assert feats.shape == t(g["input_features"]).shape
    assert float((feats.cpu() - t(g["input_features"])).abs().max()) < 2e-4
    out = enc(t(g["input_features"]).to(DEV))                        # the reference's own features: isolates the encoder
    assert out.shape[1] == int(g["P"])
    assert rel_err(out[:, ::4].float().cpu(), t(g["audio_embeds_every4"])) < tol
    out2 = enc(feats)                                                 # and end to end from the waveform
    assert rel_err(out2[:, ::4].float().cpu(), t(g["audio_embeds_every4"])) < max(tol, 5e-4)


@pytest.mark.gpu
def test_train_py_cold_start_runs_without_checkpoint_or_injection(tmp_path):
    """`python train.py -c cfg.yaml -n run` — no `-p`, nothing injected (ref:train.py:9-27, ref:trainer.py:23-114): pretrained tiny
    HuBERT from a local CTC checkpoint directory, fresh embed_projection, AutoTokenizer + sharded safetensors LLM from a local
    directory, `datasets` directories on disk.  One epoch = one full and one partial accumulation window, validation, checkpoint."""
    import dataclasses
    import shutil

    import yaml
    from test_models_gpu import _write_hf_llama_dir
    rec = json.load(open(os.path.join(GOLDEN, "tokenizers", "tokenizer_ids.json")))["llama3"]
    llm_dir = tmp_path / "Llama-3.2-3B-Instruct"
    shutil.copytree(os.path.join(GOLDEN, "tokenizers", "Llama-3.2-3B-Instruct"), llm_dir)
    cfg = dataclasses.replace(TINY_LLAMA, vocab_size=512, eos_token_ids=(rec["eos_token_id"],), pad_token_id=None)
    _write_hf_llama_dir(llm_dir, cfg, ri.llama_state_dict(cfg, seed=77), rec["bos_token_id"], rec["eos_token_id"])
    enc_sd = ri.hubert_encoder_state_dict(TINY_HUBERT, cfg.hidden_size, seed=78, weight_norm_keys="legacy")
    write_hubert_ctc_dir(tmp_path / "hubert-tiny-ft", TINY_HUBERT, enc_sd, layerdrop=0.0, hidden_dropout=0.1, activation_dropout=0.1,
                         attention_dropout=0.1, feat_proj_dropout=0.0, apply_spec_augment=True, mask_time_prob=0.05, mask_time_length=4,
                         mask_time_min_masks=2)
    os.makedirs(tmp_path / "data")
    write_kd_dataset(str(tmp_path / "data" / "train.hf"), 6, 400)
    write_kd_dataset(str(tmp_path / "data" / "val.hf"), 2, 400, first_len=20000, step=1500, seed=6)
    conf = dict(seed_everything=1234, data=dict(base_path=str(tmp_path / "data"), train_set=["train.hf"], val_set=["val.hf"]),
                model=dict(audio_encoder=dict(base="hubert", type=str(tmp_path / "hubert-tiny-ft"), downsample_method="pool", downsample_factor=4,
                                              pooling=dict(kernel_size=8, stride=4)),
                           llm_type=str(llm_dir), llm_embedding_channels=cfg.hidden_size),
                audio=dict(sampling_rate=16000),
                train=dict(num_gpus=1, num_workers=0, optimizer=dict(lr="5e-5", beta1=0.9, beta2=0.999), batch_size=1, grad_accum_interval=4, epochs=1,
                           use_ld_loss=True, use_fd_loss=True, ntp_loss_weight=0.5, ld_loss_weight=0.5, fd_loss_weight=1.0,
                           fd_loss_connector_layers=[0, 1, 3]),
                log=dict(checkpoint_dir=str(tmp_path / "checkpoints"), log_dir=str(tmp_path / "logs"), log_interval=2, validation_interval=30000,
                         num_generate_samples=1),
                runtime=dict(dtype="fp32", max_ctx=256, max_batch=4))
    with open(tmp_path / "cfg.yaml", "w") as f:
        yaml.safe_dump(conf, f)
    r = subprocess.run([sys.executable, os.path.join(REPO, "train.py"), "-c", str(tmp_path / "cfg.yaml"), "-g", "0", "-n", "cold"],
                       cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [json.loads(l) for l in open(tmp_path / "logs" / "cold" / "metrics.jsonl")]
    assert any("train/total" in l for l in lines) and any("validation/audio_perplexity" in l for l in lines)
    assert all(math.isfinite(v) for l in lines for k, v in l.items() if isinstance(v, float))
    ck = torch.load(tmp_path / "checkpoints" / "cold" / "epoch_0_step_6.pt", map_location="cpu", weights_only=False)
    assert ck["step"] == 6 and ck["lr_scheduler"]["last_epoch"] == 2              # 4 + 2 samples: two optimizer steps
    trained = ck["audio_encoder"]
    k = "encoder.encoder.layers.0.attention.q_proj.weight"
    assert trained[k].shape == enc_sd[k].shape and not torch.equal(trained[k], enc_sd[k])
    assert float((trained[k] - enc_sd[k]).abs().max()) < 1e-3                     # two AdamW steps at lr 5e-5 away from the PRETRAINED weights
    drawn = weights.init_embed_projection(TINY_HUBERT.hidden_size, cfg.hidden_size, 1234)["embed_projection.weight"]
    assert float((trained["embed_projection.weight"] - drawn).abs().max()) < 1e-3  # ... and from the seeded projection draw
