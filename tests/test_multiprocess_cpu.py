"""N > 1 path on CPU: the sharding logic that `bench.py --gpus N` and sharded
reverse sampling use, exercised with two gloo ranks (no GPU needed).

The hot path has no data-path collective: ranks own disjoint batch / sequence
ranges and disjoint Philox streams; the only communication is the timing
reduction (max over ranks) and an optional final gather of token ids."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _host_helpers():
    """shard_range / gather_token_shards of the package's sharded sampling driver."""
    sys.path.insert(0, ROOT)
    from d3pm_amd.protein_generation.generate import shard_range, gather_token_shards
    return shard_range, gather_token_shards


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
    shard_range, gather_token_shards = _host_helpers()
    n_total, L, T, K = 10, 12, 8, 22
    lo, hi = shard_range(n_total, rank, world)
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
    full = gather_token_shards(local)
    if rank == 0:
        torch.save({"tokens": full, "max_time": float(el.item())}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_everything():
    shard_range, _ = _host_helpers()
    for n in (0, 1, 7, 10000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


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
