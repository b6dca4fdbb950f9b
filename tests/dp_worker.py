"""Worker of tests/test_dp_gpu.py (not a test itself): one rank of a tiny data-parallel KD run through `Trainer`.

    RANK=r WORLD_SIZE=w MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_worker.py <out.pt> <n_rows> <accum> <workdir>

Every rank builds the same seeded tiny HuBERT + tiny Llama (fp32, regularisers off), trains one epoch over the same synthetic
dataset and dumps what the equivalence test compares: the gradients handed to AdamW at every optimizer step (after the
all-reduce), the fp32 master weights at the end, the validation perplexities, this rank's sample order.  All ranks share
cuda:0 and exchange through gloo (RCCL refuses two ranks on one device); the code path — GradArena, BucketedAllReduce's
bucket sequence, close_window, sharded validation — is the one RCCL runs with on a multi-GPU node.
"""
import os
import sys
from types import SimpleNamespace

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    out_path, n_rows, accum, workdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(0)
    backend = os.environ.get("DP_BACKEND", "gloo")
    if world > 1 or backend == "nccl":
        import torch.distributed as dist
        if backend == "nccl":     # RCCL, one rank (it refuses two per device): the collectives are identity sums, the call path is the real one
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda:0"))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import golden, pkg, t
    from oracle.golden_cfgs import TINY_HUBERT, TINY_LLAMA
    from test_models_gpu import StubTokenizer, make_encoder, make_llama
    ri, cfgm, utils, trainer_mod, dist_mod = pkg("random_init"), pkg("config"), pkg("utils"), pkg("trainer"), pkg("dist")
    g = golden("pipeline_tiny")
    gen = torch.Generator().manual_seed(5)
    V = TINY_LLAMA.vocab_size

    def row(n, nt, nr):
        return {"audio": {"array": ri.synthetic_waveform(n, seed=n)}, "text": f"utt{n}",
                "text_input_ids": torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, V, (nt,), generator=gen)]),
                "response_input_ids": torch.cat([torch.zeros(1, dtype=torch.long), torch.randint(1, V, (nr,), generator=gen)])[None],
                "pool_ranges_4": []}

    train_ds = [row(16000 + 700 * i, 5 + i % 3, 6 + i % 4) for i in range(n_rows)]
    val_ds = [row(20000 + 1500 * i, 6 - i % 2, 7 + i % 3) for i in range(5)]
    conf = cfgm.from_dict(dict(seed_everything=1234, audio=dict(sampling_rate=16000),
                               model=dict(audio_encoder=dict(base="hubert", type="synthetic", downsample_method="pool", downsample_factor=4,
                                                             pooling=dict(kernel_size=8, stride=4)),
                                          llm_embedding_channels=TINY_LLAMA.hidden_size, llm_type=utils.LLAMA_ID),
                               train=dict(optimizer=dict(lr=5e-5, beta1=0.9, beta2=0.999), batch_size=1, grad_accum_interval=accum, epochs=1, **({"per_rank_accum": int(os.environ["DP_PER_RANK_ACCUM"])} if os.environ.get("DP_PER_RANK_ACCUM") else {}),
                                          use_ld_loss=True, use_fd_loss=True, ntp_loss_weight=0.5, ld_loss_weight=0.5, fd_loss_weight=1.0,
                                          fd_loss_connector_layers=[0, 1, 3]),
                               log=dict(checkpoint_dir=os.path.join(workdir, f"ckpt_w{world}"), log_dir=os.path.join(workdir, f"logs_w{world}"),
                                        log_interval=4, validation_interval=100000, num_generate_samples=1)))
    # DP_PERTURB=1: every rank but 0 starts from OTHER encoder weights — Trainer must bring all ranks to rank 0's before the first step
    enc_seed = 51 + (rank if os.environ.get("DP_PERTURB") == "1" else 0)
    enc, _ = make_encoder(TINY_HUBERT, TINY_LLAMA.hidden_size, enc_seed, torch.float32)
    llm, _ = make_llama(TINY_LLAMA, 52, torch.float32)
    tok = StubTokenizer({utils.LLAMA_PROMPT_PREFIX: t(g["prefix_ids"]), utils.LLAMA_PROMPT_SUFFIX: t(g["suffix_ids"])})
    args = SimpleNamespace(run_name="dp", checkpoint_path=None, gpu_idx=0, no_regularizers=True)
    tr = trainer_mod.Trainer(args, conf, "cuda:0", tokenizer=tok, llm=llm, audio_encoder=enc, train_dataset=train_ds, val_dataset=val_ds,
                             dtype=torch.float32)
    if backend == "nccl" and world == 1:
        tr.kd.reducer = dist_mod.BucketedAllReduce(tr.kd.enc_tape.arena, single_rank=True)
    if tr.kd.reducer is not None:
        tr.kd.reducer.min_bytes = 64 << 10          # several buckets even at the tiny model's 1.3 MB of gradients
        tr.kd.reducer.time_exchange = True          # what bench.py's kd_step.comm reports: asserted by the test
    master0 = {k: v.cpu().clone() for k, v in tr.kd.master.items()}
    dev_w0 = [w.cpu().clone() for w in tr.audio_encoder.weights._keep]
    tr.kd.keep_last_grads = True
    steps, buckets = [], []
    inner = tr.kd.optimizer_step

    def recording_step():
        inner()
        steps.append({k: v.cpu() for k, v in tr.kd.last_grads.items()})
        if tr.kd.reducer is not None:
            buckets.append(list(tr.kd.reducer.last_buckets))

    tr.kd.optimizer_step = recording_step
    val = {}
    inner_validate = tr.validate

    def recording_validate(epoch):
        val.update(inner_validate(epoch))
        return val

    tr.validate = recording_validate
    tr.train()
    comm = tr.kd.reducer.comm_info() if tr.kd.reducer is not None else None
    tr.close()                                      # collective tear-down of the exchange, before the process group goes
    torch.save({"rank": rank, "comm": comm, "world": world, "grads": steps, "buckets": buckets,
                "master": {k: v.cpu() for k, v in tr.kd.master.items()}, "master0": master0, "dev_w0": dev_w0, "val": val, "indices": tr._epoch_indices(0),
                "windows": tr._epoch_windows(0), "step": tr.step, "optimizer_steps": tr.kd.optimizer_steps,
                "lr": tr.lr_scheduler.get_last_lr()[0]}, out_path)
    if world > 1 or backend == "nccl":
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
