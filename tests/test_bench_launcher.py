"""CPU: `python bench.py --gpus N` run as ONE process starts N ranks itself (child processes under torch.distributed.run), forwards
rank 0's JSON line as its only stdout line and hands back the children's exit code.  SL_BENCH_DRY=1 keeps the ranks off the GPU:
gloo rendezvous + the barrier / max-over-ranks timing bracket, nothing else."""
import json
import os
import subprocess
import sys

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def run(extra_env, *argv):
    env = dict(os.environ, SL_BENCH_DRY="1", **extra_env)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, BENCH, *argv], capture_output=True, text=True, env=env, timeout=300, cwd="/tmp")


def test_gpus_2_launches_two_ranks_and_forwards_one_line():
    r = run({}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                   # the ranks' stray stdout went to stderr
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1
    assert "a stray stdout line from a rank" in r.stderr
    # the self-describing fields survive the launcher: what every rank did, and what carried the gradient exchange
    assert [p_["rank"] for p_ in rec["per_rank"]] == [0, 1] and all(p_["utterances"] == 3 and p_["tokens"] == 3 * 256 for p_ in rec["per_rank"])
    assert rec["collective_backend"] == "gloo"
    c = rec["kd_step"]["comm"]
    assert c["backend"] == "torch" and c["group_world"] == 2 and c["group_backend"] == "gloo" and not c["fell_back"] and c["sum_ok"]
    assert c["buckets"] == 2 and sum(c["bucket_bytes"]) == 4 * (1024 + 64) and c["rccl_nranks"] is None


def test_a_hung_kd_collective_still_prints_the_line_and_exits_non_zero():
    """a rank whose KD leg never returns (a stuck collective) prints its line — with kd_step.error — and then leaves
    with bench.EXIT_KD_HUNG through the same run_bounded / leave_process pair the GPU run uses; the launcher hands the status back."""
    r = run({"SL_BENCH_DRY_KD_HUNG": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "0")
    assert r.returncode != 0, r.stdout
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert "did not complete" in rec["kd_step"]["error"] and rec["n_gpus"] == 2
    # the exchange's self-description is still in the record: per-bucket durations and the algorithmic bytes a rank puts on the wire
    c = rec["kd_step"]["comm"]
    assert c["algo_bytes_on_wire"] == {"per_rank_sent_total": 4 * 1088, "ring_per_link": 4 * 1088, "direct_per_link": 4 * 1088}    # 2 (N-1)/N of the payload at N = 2
    assert isinstance(c["bucket_ms"], list)


def test_wire_bytes_of_the_gradient_exchange_at_eight_ranks():
    """kd_step.comm.algo_bytes_on_wire: 2 (N-1)/N of the arena per rank; a ring pushes all of it through one link direction, the direct
    reduce-scatter + all-gather spreads it over the N-1 links (14.6 ms vs 2.1 ms for 1.274 GB at 8 ranks and ~153 GB/s per link)."""
    import importlib
    sys.path.insert(0, REPO)
    dm = importlib.import_module("llm-speech-summarization_amd.dist")
    w = dm.BucketedAllReduce.wire_bytes(1274350080, 8)
    assert w["per_rank_sent_total"] == 2 * 7 * 1274350080 // 8 and w["ring_per_link"] == w["per_rank_sent_total"] and w["direct_per_link"] == w["per_rank_sent_total"] // 7
    assert abs(w["ring_per_link"] / 153e9 * 1e3 - 14.6) < 0.1 and abs(w["direct_per_link"] / 153e9 * 1e3 - 2.08) < 0.05
    assert dm.BucketedAllReduce.wire_bytes(1000, 1) == {"per_rank_sent_total": 0, "ring_per_link": 0, "direct_per_link": 0}


def test_wrong_n_gpus_in_the_line_is_an_error():
    r = run({"SL_BENCH_DRY_REPORT_GPUS": "1"}, "--gpus", "2", "--steps", "2")
    assert r.returncode == 4 and "n_gpus=1" in r.stderr


def test_a_failing_rank_fails_the_launch():
    r = run({"SL_BENCH_DRY_FAIL_RANK": "1"}, "--gpus", "2", "--steps", "2")
    assert r.returncode != 0


def test_world_size_mismatch_is_refused_before_any_gpu_work():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"], capture_output=True, text=True, env=env, timeout=120, cwd="/tmp")
    assert r.returncode == 4 and "WORLD_SIZE=2" in r.stderr


def test_launched_by_torchrun_directly_as_the_driver_does():
    """The driver's own form: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ..."""
    env = dict(os.environ, SL_BENCH_DRY="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29731", BENCH, "--gpus", "2", "--steps", "2", "--warmup", "0"], capture_output=True, text=True,
                       env=env, timeout=300, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["n_gpus"] == 2
