"""N > 1 path on CPU: the sharding logic that `bench.py --gpus N` and sharded
reverse sampling use, exercised with two gloo ranks, and the N-rank launch of
`bench.py` itself (no GPU and no HIP library needed: the helpers are loaded from
their source file, not through the package, which would load the library).

The hot path has no data-path collective: ranks own disjoint batch / sequence
ranges and disjoint Philox streams; the only communication is the timing
reduction (max over ranks) and an optional final gather of token ids."""
import importlib.util
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

PKG = os.path.join(ROOT, "exploring-discrete-denoising-diffusion-models-for-protein-design_amd")


def _host_helpers():
    """The package's sharding module, loaded by path (importing the package would load libd3pm_hip.so)."""
    spec = importlib.util.spec_from_file_location("d3pm_sharding_under_test", os.path.join(PKG, "sharding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import protein_style as pr
    sh = _host_helpers()
    n_total, L, T, K = 10, 12, 8, 22
    lo, hi = sh.shard_range(n_total, rank, world)
    qb, qt = pr.uniform_tables(K, T, "sohl-dickstein")
    # every rank draws its own shard from a stream keyed by (seed, global sequence index): the result
    # does not depend on the number of ranks
    xs = []
    for i in range(lo, hi):
        g = torch.Generator().manual_seed(1000 + i)
        x = torch.randint(0, K, (1, L), generator=g)
        for t in range(T - 1, 0, -1):
            logits = torch.randn(1, L, K, generator=g)
            e = torch.empty(1, L, K, dtype=torch.float64).exponential_(1, generator=g)
            x = pr.reverse_step(logits, x, t, qt, qb, e)
        xs.append(x)
    local = torch.cat(xs) if xs else torch.zeros((0, L), dtype=torch.int64)
    # timing reduction as in bench.py: max over ranks
    el = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    # final gather of uint8 token ids (ragged shards: padded to the largest), the driver's own routine
    full = sh.gather_token_shards(local)
    if rank == 0:
        torch.save({"tokens": full, "max_time": float(el.item())}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    sh = _host_helpers()
    for n in (0, 1, 7, 10000):
        for w in (1, 2, 3, 8):
            spans = [sh.shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_equal_sub_batches():
    """10 000 sequences over 1, 2, 4, 8 ranks in sub-batches of at most 1250: every rank replays equal
    sub-batches of 1250 (no short tail at the launch floor)."""
    sh = _host_helpers()
    for world in (1, 2, 4, 8):
        for r in range(world):
            lo, hi = sh.shard_range(10000, r, world)
            sizes = sh.equal_sub_batches(hi - lo, 1250)
            assert sizes == [1250] * (8 // world)
    assert sh.equal_sub_batches(0, 1250) == []
    assert sh.equal_sub_batches(1251, 1250) == [626, 625]
    for n in (1, 7, 1024, 3333):
        sizes = sh.equal_sub_batches(n, 1024)
        assert sum(sizes) == n and max(sizes) <= 1024 and max(sizes) - min(sizes) <= 1


def test_bench_spawn_plan():
    """`bench.py --gpus N` without a torchrun environment plans N children with distinct ranks / local ranks,
    one rendezvous address."""
    sys.path.insert(0, ROOT)
    import bench
    plan = bench.spawn_plan(8, 29555, base_env={"PATH": "/usr/bin"})
    assert len(plan) == 8
    assert [e["RANK"] for e in plan] == [str(i) for i in range(8)]
    assert [e["LOCAL_RANK"] for e in plan] == [str(i) for i in range(8)]
    assert {e["WORLD_SIZE"] for e in plan} == {"8"}
    assert {(e["MASTER_ADDR"], e["MASTER_PORT"]) for e in plan} == {("127.0.0.1", "29555")}
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and e["PATH"] == "/usr/bin" for e in plan)


@pytest.mark.timeout(120)
def test_bench_gpus_n_starts_n_ranks(tmp_path):
    """The launch path of `python bench.py --gpus 3`, rehearsed without a GPU: the parent starts three fresh
    children before any GPU call, each sees its own rank environment, rank 0's line is relayed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["D3PM_BENCH_SPAWN_ECHO"] = str(tmp_path / "echo")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "5"], env=env,
                         capture_output=True, text=True, timeout=100)
    assert res.returncode == 0, res.stderr
    assert json.loads(res.stdout.strip().splitlines()[-1]) == {"n_gpus": 3, "echo": True}
    recs = [json.load(open(f"{tmp_path}/echo.{r}")) for r in range(3)]
    assert [r["LOCAL_RANK"] for r in recs] == ["0", "1", "2"] and [r["RANK"] for r in recs] == ["0", "1", "2"]
    assert {r["WORLD_SIZE"] for r in recs} == {"3"} and len({r["MASTER_PORT"] for r in recs}) == 1


def test_bench_refuses_mismatched_world(tmp_path):
    """Under a launcher whose world size differs from --gpus the script stops instead of reporting the wrong N."""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    env.pop("D3PM_BENCH_SPAWN_ECHO", None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env,
                         capture_output=True, text=True, timeout=100)
    assert res.returncode != 0 and "WORLD_SIZE=2" in res.stderr


@pytest.mark.timeout(180)
def test_two_rank_sharded_sampling_matches_single_rank(tmp_path):
    outs = {}
    for world in (1, 2):
        out = str(tmp_path / f"w{world}.pt")
        mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        outs[world] = torch.load(out, weights_only=True)
    assert torch.equal(outs[1]["tokens"], outs[2]["tokens"])
    assert outs[1]["tokens"].shape == (10, 12)
    assert abs(outs[2]["max_time"] - 0.2) < 1e-12 and abs(outs[1]["max_time"] - 0.1) < 1e-12
