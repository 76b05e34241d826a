"""Data-parallel training of a denoiser around the D3PM core (BASELINE config 5: DDP of the protein denoiser with
the fused loss kernels), through `tools/train_ddp.py`: one rank in-process, and two ranks on the one visible GPU
with the gloo backend (a rehearsal of the N-rank code path: RCCL needs one GPU per rank)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
TOOL = os.path.join(ROOT, "tools", "train_ddp.py")


def _load_tool():
    import importlib.util
    spec = importlib.util.spec_from_file_location("train_ddp_under_test", TOOL)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_training_harness_single_rank():
    """Three optimiser steps of a 2-layer denoiser with the two-launch training step: finite losses (their size
    depends on the timesteps drawn, so no trend is asserted over so few steps), the path a small share of the step."""
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    res = _load_tool().main(["--steps", "3", "--warmup", "2", "--layers", "2", "--d-model", "128",
                             "--batch-per-gpu", "64", "--watchdog", "100"])
    assert res["n_gpus"] == 1 and res["global_batch"] == 64
    assert 0.0 < res["mean_loss_bits_per_dim"] < 10.0 and 0.0 < res["first_step_loss_bits_per_dim"] < 10.0
    assert 0.0 < res["path_share"] < 0.5 and res["allreduce_share"] == 0.0


@pytest.mark.timeout(300)
def test_training_harness_two_ranks_gloo():
    """Two ranks, per-rank batch mean + averaged gradients: both ranks end with identical weights."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), TOOL, "--steps", "3", "--warmup",
                          "2", "--layers", "2", "--d-model", "128", "--batch-per-gpu", "64", "--backend", "gloo",
                          "--watchdog", "60"], env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["backend"] == "gloo" and out["global_batch"] == 128
    assert out["weight_checksum_spread_over_ranks"] == 0.0
    assert 0.0 < out["mean_loss_bits_per_dim"] < 10.0


def test_rccl_refuses_two_ranks_per_gpu():
    """`--backend nccl` with more ranks than GPUs stops with a message instead of stalling in the communicator."""
    import torch
    n = torch.cuda.device_count() + 1
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    res = subprocess.run([sys.executable, TOOL, "--steps", "1"], env=env, capture_output=True, text=True, timeout=200)
    assert res.returncode != 0 and "one GPU per rank" in res.stderr
