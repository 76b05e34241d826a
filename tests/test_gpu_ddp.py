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


def test_training_harness_single_rank_image_style():
    """Three optimiser steps of a 2-layer denoiser with the d3pm-style two-launch training step: finite losses
    (their size depends on the timesteps drawn, so no trend is asserted over so few steps), the path a small share
    of the step."""
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    res = _load_tool().main(["--style", "image", "--steps", "3", "--warmup", "2", "--layers", "2", "--d-model", "128",
                             "--batch-per-gpu", "64", "--watchdog", "100"])
    assert res["n_gpus"] == 1 and res["global_batch"] == 64 and res["style"] == "image"
    assert 0.0 < res["mean_loss"] < 10.0 and 0.0 < res["first_step_loss"] < 10.0
    assert 0.0 < res["path_share"] < 0.5 and res["allreduce_share"] == 0.0


@pytest.mark.timeout(300)
def test_training_harness_protein_style_step_against_oracle(tmp_path):
    """BASELINE configuration 5 on one rank, as the reference trains (notebook cell-12): lean Collater -> denoiser
    [B, L] -> [B, L, 22] -> (D3PMLVBLoss + 0.001 CrossEntropyLoss) * n_tokens, 256 sequences of 263 positions, K = 22.
    The last step's (x0, x_t, t, logits) are captured and its loss and d loss / d logits compared with the CPU
    oracle's restatement of that loop."""
    import numpy as np
    import torch
    from oracle import protein_style as pr
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    dump = str(tmp_path / "step.pt")
    res = _load_tool().main(["--style", "protein", "--steps", "2", "--warmup", "1", "--layers", "2", "--d-model", "128",
                             "--batch-per-gpu", "256", "--seq-len", "263", "--watchdog", "100", "--dump", dump])
    assert res["n_gpus"] == 1 and res["global_batch"] == 256 and res["K"] == 22 and res["style"] == "protein"
    assert 0.0 < res["mean_loss"] < 20.0 and 0.0 < res["path_share"] < 0.5
    st = torch.load(dump)
    x0, x_t, t, lam = st["x0"], st["x_t"], st["t"], float(st["lam"])
    assert x0.shape == (256, 263) and st["logits"].shape == (256, 263, 22) and int(t.min()) >= 1 and int(t.max()) <= 499
    assert float((x0 != x_t).double().mean()) > 0.05       # the batch was corrupted
    qb, qt = pr.uniform_tables(22, 500, "sohl-dickstein")
    lg = st["logits"].float().clone().requires_grad_(True)
    loss, _, _ = pr.driver_loss(pr.one_hot(x_t, 22), None, lg, x0, pr.one_hot(x0, 22), t, qt, qb, lam, K=22, tmax=500)
    loss.backward()
    np.testing.assert_allclose(float(st["loss"]), float(loss.detach()), rtol=1e-5)
    ref = lg.grad.numpy()
    np.testing.assert_allclose(st["dlogits"].numpy(), ref, rtol=1e-5, atol=1e-6 * np.abs(ref).max())


@pytest.mark.timeout(300)
@pytest.mark.parametrize("style", ["protein", "image"])
def test_training_harness_two_ranks_gloo(style):
    """Two ranks, per-rank batch mean + averaged gradients: both ranks end with identical weights."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), TOOL, "--style", style, "--steps", "3",
                          "--warmup", "2", "--layers", "2", "--d-model", "128", "--batch-per-gpu", "64", "--backend",
                          "gloo", "--watchdog", "60"], env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-3000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["backend"] == "gloo" and out["global_batch"] == 128 and out["style"] == style
    assert out["weight_checksum_spread_over_ranks"] == 0.0
    assert 0.0 < out["mean_loss"] < 20.0


def test_watchdog_is_cancelled_when_the_harness_fails():
    """An exception inside the harness must not leave its watchdog armed in the calling process."""
    import faulthandler
    import time
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        os.environ.pop(k, None)
    with pytest.raises(BaseException):
        _load_tool().main(["--style", "protein", "--steps", "1", "--warmup", "1", "--layers", "0", "--d-model", "-8",
                           "--watchdog", "1.5"])
    time.sleep(2.5)                                        # an armed watchdog would have ended this process by now


def test_rccl_refuses_two_ranks_per_gpu():
    """`--backend nccl` with more ranks than GPUs stops with a message instead of stalling in the communicator."""
    import torch
    n = torch.cuda.device_count() + 1
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    res = subprocess.run([sys.executable, TOOL, "--steps", "1"], env=env, capture_output=True, text=True, timeout=200)
    assert res.returncode != 0 and "one GPU per rank" in res.stderr
