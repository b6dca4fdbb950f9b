#!/usr/bin/env python3
"""Drop-in entry point with the reference's CLI (ref:train.py:9-27):

    python train.py -c config/llama3_hubert.yaml -g 0 -n run_name [-p checkpoint.pt]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py -c ... -n run   # data parallel

One process per GPU; under torch.distributed.run the process group is RCCL (backend "nccl") and -g is ignored in
favour of LOCAL_RANK.  The implementation is `llm-speech-summarization_amd/trainer.py`.
"""
import argparse
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('-c', '--config', type=str, required=True, help="yaml file for configuration")
    parser.add_argument('-g', '--gpu_idx', type=int, default=0, help="index of home GPU device")
    parser.add_argument('-n', '--run_name', type=str, required=True, help="name of the run")
    parser.add_argument('-p', '--checkpoint_path', type=str, default=None, help="path of checkpoint to resume from")
    args = parser.parse_args()
    import torch
    import datetime
    distributed = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
    if distributed:
        args.gpu_idx = int(os.environ["LOCAL_RANK"])
    # libspeechllm launches on the current HIP device / stream: make -g N current before anything is allocated
    torch.cuda.set_device(args.gpu_idx)
    if distributed:
        # validation (sharded over the ranks) ends at a barrier behind rank 0's sample generations + checkpoint write:
        # give collectives a generous watchdog
        torch.distributed.init_process_group("nccl", device_id=torch.device(f"cuda:{args.gpu_idx}"), timeout=datetime.timedelta(hours=2))
    device = torch.device(f"cuda:{args.gpu_idx}")
    config = importlib.import_module("llm-speech-summarization_amd.config").load_config(args.config)
    dtype = torch.float32 if str(config.get("runtime", {}).get("dtype", "bf16")) == "fp32" else torch.bfloat16
    Trainer = importlib.import_module("llm-speech-summarization_amd.trainer").Trainer
    trainer = Trainer(args, config, device, dtype=dtype)
    try:
        trainer.train()
    except BaseException:
        # A rank that fails alone must EXIT, not wait: its peers may be inside an RCCL all-reduce it will never join, and both the
        # collective close() (synchronize + ncclCommDestroy) and destroy_process_group() would block on them — torchrun would then never
        # see the failure.  Local abort of the communicator, the exception re-raised: the process leaves non-zero and the launcher ends the job.
        import traceback
        traceback.print_exc()
        trainer.abort()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)                # not `raise`: interpreter shutdown would still run the process group's destructor, which may wait on the peers
    trainer.close()                # success path only: collective tear-down of the gradient exchange, while the process group still exists
    if distributed:
        torch.distributed.destroy_process_group()
