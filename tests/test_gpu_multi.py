"""Multi-GPU path on hardware (RCCL): runs with 2 ranks when the box has >= 2 GPUs, with 1 rank (same program, nccl backend,
no exchange partner) otherwise -- the driver's 8-GPU node runs the real thing, the 1-GPU dev box still executes every line."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _world():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return 2 if torch.cuda.device_count() >= 2 else 1


def test_data_parallel_step_one_allreduce_over_rccl():
    world = _world()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker_gpu.py")],
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "DIST_OK world=%d" % world in r.stdout


def test_two_real_ranks_sharing_one_gpu_over_gloo():
    """A 1-GPU box cannot give RCCL two ranks -- but two PROCESSES can share cuda:0 and exchange through gloo (which moves CUDA
    tensors).  That runs the whole 2-rank path on HIP-produced gradients: sharding by cloud, per-shard parity against a single
    process, the flat store's all-reduce, and the whole autoencoder's GradArena with exactly ONE counted collective per step whose
    result is the mean of the two ranks' gradients.  Only the transport differs from the 8-GPU run."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(_env(), DPF_TEST_BACKEND="gloo", DPF_TEST_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker_gpu.py")],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "DIST_OK world=2 backend=gloo" in r.stdout


def test_bench_two_ranks_sharing_one_gpu_reports_one_collective():
    """bench.py's training leg with --gpus 2 on ONE GPU (DPF_BENCH_BACKEND=gloo, DPF_BENCH_SHARE_GPU=1): the launcher starts two
    ranks, every rank trains on its own clouds, the line reports the COUNTED collectives per step (1) and the all-reduce's size."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    env = dict(_env(), DPF_BENCH_BACKEND="gloo", DPF_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--leg", "train", "--model", "autoencoder", "--layers", "6",
                        "--batch", "4", "--points", "512", "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    t = line["train_step"]
    assert line["n_gpus"] == 2 and line["value"] > 0 and t["replay_equals_eager"] is True
    assert t["collectives_per_step"] == 1 and t["collectives_counted"] is True
    assert t["allreduce"]["backend"] == "gloo" and t["allreduce"]["elements"] * 4 == t["flat_gradient_bytes"]


def test_bench_gpus_flag_on_hardware():
    """`python bench.py --gpus N` starts its own ranks; the train leg reports the collective when N > 1."""
    world = _world()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--leg", "train", "--layers", "6",
                        "--batch", "4", "--points", "512", "--steps", "3", "--warmup", "1"],
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == world and line["value"] > 0
    if world > 1:
        ar = line["train_step"]["allreduce"]
        assert ar["backend"] == "nccl" and ar["us"] > 0 and line["train_step"]["collectives_per_step"] == 1
