"""CPU: the oracle (oracle/*.py) reproduces what the REFERENCE classes produced in the build container.

Fixtures: tests/golden/*.npz, written by oracle/gen_golden.py from ref:model/audio_encoder.py,
ref:model/audio_llama.py and ref:utils.py.  Tolerances: fp32 hidden states <= 2e-5 relative L2,
greedy ids identical.
"""
import pytest
import torch

from conftest import golden, pkg, rel_err, t
from oracle import hubert_oracle as ho
from oracle import kd_oracle as ko
from oracle import llama_oracle as lo
from oracle.golden_cfgs import TINY_HUBERT, TINY_LLAMA, TINY_MHA, WIDE_HUBERT, WIDE_LLAMA

ri = pkg("random_init")
TOL = 2e-5


@pytest.mark.parametrize("n", [16000, 32000])
def test_encoder_tiny_pool_all_stages(n):
    g = golden(f"enc_tiny_pool_{n}")
    sd = ri.hubert_encoder_state_dict(TINY_HUBERT, 256, seed=int(g["weight_seed"]))
    wave = ri.synthetic_waveform(n, seed=int(g["wave_seed"]))[None]
    taps = {}
    out = ho.audio_encoder_forward(sd, TINY_HUBERT, wave, "pool", taps=taps)
    for k in [k for k in g if k in taps and k != "audio_embeds"]:
        assert taps[k].shape == t(g[k]).shape, k
        assert rel_err(taps[k], t(g[k])) < TOL, k
    assert rel_err(out, t(g["audio_embeds"])) < TOL
    assert out.shape[1] == (TINY_HUBERT.num_frames(n) - 8) // 4 + 1


def test_encoder_weight_norm_key_spellings_agree():
    a = ri.hubert_encoder_state_dict(TINY_HUBERT, 256, seed=3, weight_norm_keys="parametrizations")
    b = ri.hubert_encoder_state_dict(TINY_HUBERT, 256, seed=3, weight_norm_keys="legacy")
    wave = ri.synthetic_waveform(8000, seed=1)[None]
    assert torch.equal(ho.audio_encoder_forward(a, TINY_HUBERT, wave), ho.audio_encoder_forward(b, TINY_HUBERT, wave))


def test_encoder_stack_and_quirk():
    g = golden("enc_tiny_stack_16000")
    sd = ri.hubert_encoder_state_dict(TINY_HUBERT, 256, seed=12, downsample="stack")
    wave = ri.synthetic_waveform(16000, seed=int(g["wave_seed"]))[None]
    out = ho.audio_encoder_forward(sd, TINY_HUBERT, wave, "stack")
    assert rel_err(out, t(g["audio_embeds"])) < TOL
    # T % 4 == 0: the reference returns an EMPTY sequence and the oracle reproduces that
    g = golden("enc_tiny_stack_16720")
    assert int(g["T"]) % 4 == 0 and g["audio_embeds"].shape[1] == 0
    wave = ri.synthetic_waveform(16720, seed=int(g["wave_seed"]))[None]
    assert ho.audio_encoder_forward(sd, TINY_HUBERT, wave, "stack").shape[1] == 0
    assert ho.audio_encoder_forward(sd, TINY_HUBERT, wave, "stack", fix_stack_quirk=True).shape[1] == int(g["T"]) // 4


def test_encoder_ctc_pool():
    g = golden("enc_tiny_ctcpool_16000")
    sd = ri.hubert_encoder_state_dict(TINY_HUBERT, 256, seed=13, downsample="ctc_pool")
    wave = ri.synthetic_waveform(16000, seed=int(g["wave_seed"]))[None]
    out = ho.audio_encoder_forward(sd, TINY_HUBERT, wave, "ctc_pool",
                                   ctc_pool_ranges=[[tuple(r) for r in g["ranges"].tolist()]])
    assert rel_err(out, t(g["audio_embeds"])) < TOL


def test_encoder_batch2():
    g = golden("enc_tiny_pool_batch2")
    sd = ri.hubert_encoder_state_dict(TINY_HUBERT, 256, seed=11)
    wave = torch.stack([ri.synthetic_waveform(24000, seed=int(s)) for s in g["wave_seeds"]])
    assert rel_err(ho.audio_encoder_forward(sd, TINY_HUBERT, wave), t(g["audio_embeds"])) < TOL


def test_encoder_wide():
    g = golden("enc_wide_pool_32000")
    sd = ri.hubert_encoder_state_dict(WIDE_HUBERT, 3072, seed=int(g["weight_seed"]))
    wave = ri.synthetic_waveform(32000, seed=int(g["wave_seed"]))[None]
    taps = {}
    out = ho.audio_encoder_forward(sd, WIDE_HUBERT, wave, taps=taps)
    for k in ("conv6", "feature_projection", "layer0", "last_hidden_state"):
        assert rel_err(taps[k], t(g[k])) < TOL, k
    assert rel_err(out, t(g["audio_embeds"])) < TOL


def test_num_audio_embeds_known_answers():
    g = golden("num_audio_embeds")
    for n, e in zip(g["n_samples"].tolist(), g["expected"].tolist()):
        assert ho.compute_num_audio_embeds(n) == e
    assert ho.compute_num_audio_embeds(160000) == 123 and ho.compute_num_audio_embeds(480000) == 373


@pytest.mark.parametrize("name,cfg", [("tiny_gqa", TINY_LLAMA), ("tiny_mha", TINY_MHA)])
def test_llama_tiny_forward_loss_generate(name, cfg):
    g = golden(f"llama_{name}")
    sd = ri.llama_state_dict(cfg, seed=int(g["weight_seed"]))
    gen = torch.Generator().manual_seed(int(g["embeds_seed"]))
    S = int(g["S"])
    x = torch.randn(1, S, cfg.hidden_size, generator=gen) * 0.05
    labels = [torch.randint(0, cfg.vocab_size, (6,), generator=gen)]
    assert torch.equal(labels[0], t(g["labels"]))
    out = lo.llama_forward(sd, cfg, x, attention_mask=torch.ones(1, S, dtype=torch.long), output_hidden_states=True)
    assert rel_err(out["logits"], t(g["logits"])) < TOL
    hs = torch.stack(out["hidden_states"])
    assert hs.shape == t(g["hidden_states"]).shape
    assert rel_err(hs, t(g["hidden_states"])) < TOL
    assert abs(float(lo.response_only_loss(out["logits"], labels)) - float(g["loss"])) < 1e-5
    assert torch.equal(lo.greedy_generate(sd, cfg, x, 32, use_eos=False), t(g["ids_noeos"]))
    assert torch.equal(lo.greedy_generate(sd, cfg, x, 32, use_eos=True), t(g["ids_eos"]))


@pytest.mark.parametrize("name,cfg", [("tiny_gqa", TINY_LLAMA), ("tiny_mha", TINY_MHA)])
def test_llama_left_padded_batch(name, cfg):
    g = golden(f"llama_{name}_padbatch")
    sd = ri.llama_state_dict(cfg, seed=int(g["weight_seed"]))
    out = lo.llama_forward(sd, cfg, t(g["x"]), attention_mask=t(g["mask"]), output_hidden_states=True)
    m = t(g["mask"]).bool()
    assert rel_err(out["logits"][m], t(g["logits"])[m]) < TOL          # padded positions are don't-care
    assert rel_err(out["last_hidden"][m], t(g["last_hidden"])[m]) < TOL


def test_llama_wide():
    g = golden("llama_wide")
    cfg = WIDE_LLAMA
    sd = ri.llama_state_dict(cfg, seed=int(g["weight_seed"]))
    gen = torch.Generator().manual_seed(int(g["embeds_seed"]))
    x = torch.randn(1, int(g["S"]), cfg.hidden_size, generator=gen) * 0.02
    out = lo.llama_forward(sd, cfg, x, output_hidden_states=True)
    assert rel_err(torch.stack(out["hidden_states"]), t(g["hidden_states"])) < TOL
    assert rel_err(out["logits"][:, -1], t(g["last_logits"])) < TOL
    ids, margins = lo.greedy_generate(sd, cfg, x, 12, use_eos=False, return_margins=True)
    assert torch.equal(ids, t(g["ids_noeos"]))
    assert rel_err(margins, t(g["margins"])) < 1e-3
    assert float(t(g["margins"]).min()) > 1e-4  # fixture ids are margin-qualified for fp32 comparisons


def test_pipeline_order_of_operations_and_kd_losses():
    g = golden("pipeline_tiny")
    cfg = TINY_LLAMA
    enc_sd = ri.hubert_encoder_state_dict(TINY_HUBERT, cfg.hidden_size, seed=int(g["enc_seed"]))
    sd = ri.llama_state_dict(cfg, seed=int(g["llm_seed"]))
    wave = ri.synthetic_waveform(int(g["n_samples"]), seed=int(g["wave_seed"]))[None]
    prefix, suffix, textp = t(g["prefix_ids"]), t(g["suffix_ids"]), t(g["text_prompt_ids"])
    audio = ho.audio_encoder_forward(enc_sd, TINY_HUBERT, wave)
    ids = ko.generate_audio_response_ids(sd, cfg, audio, prefix, suffix, None, max_new_tokens=40)
    assert torch.equal(ids, t(g["ids_audio"]))
    ids = ko.generate_audio_response_ids(sd, cfg, audio, prefix, suffix, textp, max_new_tokens=40)
    assert torch.equal(ids, t(g["ids_text_audio"]))
    P = audio.shape[1]
    assert int(g["prompt_len_audio"]) == prefix.shape[1] + P + suffix.shape[1] - 1
    assert int(g["prompt_len_text_audio"]) == prefix.shape[1] + (textp.shape[1] - 1) + P + suffix.shape[1] - 1
    # KD micro-step losses (dropout/layerdrop/spec-augment off)
    audio = audio.detach().requires_grad_()
    losses = ko.kd_losses(sd, cfg, audio, t(g["text_ids"]), t(g["response_ids"]), prefix, suffix,
                          connector_layers=tuple(g["connector_layers"].tolist()))
    for k in ("ntp", "ld", "fd", "total"):
        assert abs(float(losses[k]) - float(g[k])) < 2e-5 * max(1.0, abs(float(g[k]))), k
    n = int(t(g["response_ids"]).shape[0])
    assert int(g["kd_seq_len_audio"]) == prefix.shape[1] + P + suffix.shape[1] - 1 + n - 1
    # dgrad-only path through the frozen LLM: d(ntp)/d(audio_embeds)
    audio2 = audio.detach().clone().requires_grad_()
    seq = ko.merge_prompt_response_tokens(sd, prefix, suffix, audio2, t(g["response_ids"])[None])
    ntp = lo.response_only_loss(lo.llama_forward(sd, cfg, seq)["logits"], [t(g["response_ids"])])
    (d,) = torch.autograd.grad(ntp, audio2)
    assert rel_err(d, t(g["d_ntp_d_audio_embeds"])) < 1e-4


def test_whisper_logmel_and_encoder_vs_reference_fixture():
    from oracle import whisper_oracle as wo
    from oracle.golden_cfgs import TINY_WHISPER
    g = golden("whisper_tiny")
    sd = ri.whisper_encoder_state_dict(TINY_WHISPER, 256, seed=int(g["weight_seed"]))
    feats = torch.stack([wo.log_mel(TINY_WHISPER, ri.synthetic_waveform(int(n), seed=int(s))) for n, s in zip(g["n_samples"], g["wave_seeds"])])
    assert float((feats - t(g["input_features"])).abs().max()) < 1e-5        # feature extractor (HF torch path)
    assert rel_err(wo.audio_encoder_forward(sd, TINY_WHISPER, feats), t(g["audio_embeds"])) < TOL
