"""Reverse sampling of the protein-style D3PM with the signatures of
``protein_generation/generate.py`` (:8-44, :118-130).  The per-step posterior
marginal and categorical draw of the whole batch is one kernel launch; the
reference loops over sequences in Python and calls ``torch.multinomial`` per
sequence."""
from __future__ import annotations

import torch

from .. import _lib
from .._ops import ops, i64
from ..tables import protein_tables
from ..sharding import shard_range, equal_sub_batches, gather_token_shards  # noqa: F401  (re-exported)


def reverse_step(tables, predictions, x_t, t, noise=None, last_class_div=1.0, exclude_last_prefix=0, rng=None):
    """x_{t-1} for a batch (generate.py:28-41): predictions [B,L,n_tokens]
    float32 logits, x_t [B,L] int64, t [B] int64; ``noise`` optionally the
    Exp(1) draws [B,L,K] float64 of the reference's multinomial.  The last two
    arguments are the MSA sampler's gap handling (generate.py:84-88): the last
    class is divided by ``last_class_div`` and cannot be drawn in the first
    ``exclude_last_prefix`` positions of each batch element.  ``rng=(seed, offset,
    counter)`` names the Philox stream explicitly, ``counter`` a one-element int64
    device tensor added to ``offset`` on the device (HIP-graph capture)."""
    from ..d3pm.diffusion_discrete import _philox_ticket
    _lib.require_device(predictions, x_t)
    pr = predictions.contiguous()
    if pr.dtype != torch.float32:
        pr = pr.float()
    x_t = x_t.to(torch.int64).contiguous()
    t = t.to(device=x_t.device, dtype=torch.int64).contiguous()
    B = x_t.shape[0]
    npos = x_t.numel() // max(B, 1)
    seed = off = 0
    counter = None
    if noise is not None:
        noise = noise.to(device=x_t.device, dtype=torch.float64).contiguous()
        assert noise.shape == x_t.shape + (tables.K,), noise.shape
    elif rng is not None:
        seed, off, counter = int(rng[0]) & 0xFFFFFFFFFFFFFFFF, int(rng[1]), rng[2]
    else:
        seed, off = _philox_ticket(x_t.device)
    return ops.lvb_reverse_step(*tables.op_args(), pr, x_t, t, noise, i64(seed), i64(off), counter,
                                float(last_class_div), int(exclude_last_prefix))


def generate_d3pm(model, tokenizer, Q, Q_bar, timesteps, seq_len, device, batch_size=3, x_init=None, noise_fn=None):
    """Sample `batch_size` sequences by T-1 reverse steps from uniform noise
    (generate.py:8-44): returns (tokens [B,L] int64, list of strings).
    ``x_init`` / ``noise_fn(step) -> [B,L,K]`` replay a recorded run."""
    model.eval()
    device = torch.device(device)
    if x_init is None:
        x_init = torch.randint(0, tokenizer.K, (batch_size, seq_len))
    sample = x_init.to(torch.long).to(device)
    tables = protein_tables(Q, Q_bar, device=device)
    steps = torch.linspace(timesteps - 1, 1, int((timesteps - 1) / 1), dtype=int).tolist()
    with torch.no_grad():
        for i, t in enumerate(steps):
            tt = torch.full((sample.shape[0],), t, dtype=torch.long, device=device)
            predictions = model(sample, tt)
            sample = reverse_step(tables, predictions, sample, tt, None if noise_fn is None else noise_fn(i))
    return sample, [tokenizer.untokenize(s) for s in sample.cpu()]


def sample_and_save(out_path, num_seqs, model, tokenizer, Q, Q_bar, timesteps, seq_len, device, batch_size=3):
    """Generate `num_seqs` sequences and write them as FASTA
    (generate.py:118-130; same file name and record format).  Sequences are
    generated `batch_size` at a time instead of one by one."""
    strings = []
    while len(strings) < num_seqs:
        n = min(max(batch_size, 1), num_seqs - len(strings))
        _, s = generate_d3pm(model, tokenizer, Q, Q_bar, timesteps, seq_len, device, batch_size=n)
        strings.extend(s)
    with open(out_path + "generated_samples_string.fasta", "w") as f:
        for i, s in enumerate(strings):
            f.write(">SEQUENCE_" + str(i) + "\n" + str(s) + "\n")
    return strings


def sample_and_save_sharded(out_path, num_seqs, model, tokenizer, Q, Q_bar, timesteps, seq_len, device, batch_size=1024,
                            seed=0, graph=True):
    """``sample_and_save`` (generate.py:118-130) over all ranks of the default
    process group: every rank generates its contiguous share of the `num_seqs`
    sequences in equal sub-batches of at most `batch_size` from its own Philox stream
    (seed + rank), with no communication until the final gather; rank 0 writes the
    FASTA file in the reference's format.  ``graph``: replay the T-1 step loop of a
    sub-batch as one HIP graph (``model`` must be capturable).  Returns the strings
    on rank 0, None elsewhere."""
    import torch.distributed as dist
    from ..graphs import GraphedReverseSampler
    dist_on = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist_on else (0, 1)
    lo, hi = shard_range(num_seqs, rank, world)
    device = torch.device(device)
    tables = protein_tables(Q, Q_bar, device=device)
    model.eval()
    out = torch.empty((hi - lo, seq_len), dtype=torch.long, device=device)
    # equal sub-batches (sizes differ by at most one; one graph of the larger size serves both, a surplus row is
    # dropped): a short last sub-batch would run at the launch floor of its kernels
    sub = max(equal_sub_batches(hi - lo, batch_size), default=1)
    done = 0
    if graph and hi > lo:
        smp = GraphedReverseSampler(tables, model, sub, seq_len, timesteps, device, seed=seed + rank)
        while done < hi - lo:                      # the last sub-batch may overshoot: surplus rows are dropped
            n = min(sub, hi - lo - done)
            out[done:done + n] = smp.sample()[:n]
            done += n
    else:
        steps = list(range(timesteps - 1, 0, -1))
        counter = torch.zeros(1, dtype=torch.int64, device=device)
        with torch.no_grad():
            while done < hi - lo:
                n = min(sub, hi - lo - done)
                x = torch.randint(0, tokenizer.K, (n, seq_len), device=device)
                for i, t in enumerate(steps):
                    tt = torch.full((n,), t, dtype=torch.long, device=device)
                    x = reverse_step(tables, model(x, tt), x, tt, rng=(seed + rank, 4 * i, counter))
                counter.add_(4 * len(steps))
                out[done:done + n] = x
                done += n
    tokens = gather_token_shards(out)
    if rank != 0:
        return None
    strings = [tokenizer.untokenize(s) for s in tokens.cpu()]
    with open(out_path + "generated_samples_string.fasta", "w") as f:
        for i, s in enumerate(strings):
            f.write(">SEQUENCE_" + str(i) + "\n" + str(s) + "\n")
    return strings


def generate_msa_d3pm(model, batch_size, n_sequences, seq_length, device, timesteps=500, penalty_value=0, Q_bar=None,
                      Q=None, tokenizer=None, start_query=True, query_sequence=None, x_init=None, noise_fn=None):
    """Sample `batch_size` MSAs of `n_sequences` x `seq_length` tokens by T-1 reverse
    steps (generate.py:48-99): returns (tokens [B,D,L] int64, [[string]] per MSA).

    Per step the whole batch is one kernel launch over the D*L flattened positions:
    the gap class (the last one) is divided by 1 + penalty_value and excluded from
    the query row (the first L positions); with ``start_query`` the query row is
    pinned to ``query_sequence`` (tokens [L]; the reference draws it from its
    training MSA, generate.py:102-115, which is a data file this package does not
    read).  ``x_init`` / ``noise_fn(step) -> [B, D*L, K]`` replay a recorded run;
    in the noise of a replay the query-row rows hold the K-1 draws of the
    reference's second multinomial call in their first K-1 columns."""
    from .utils import Tokenizer
    tokenizer = tokenizer or Tokenizer()
    model.eval()
    device = torch.device(device)
    if x_init is None:
        x_init = torch.randint(0, tokenizer.K, (batch_size, n_sequences, seq_length))
    sample = x_init.to(torch.long).to(device).clone()
    if start_query:
        if query_sequence is None:
            raise ValueError("start_query=True needs query_sequence (tokens of the query row)")
        sample[:, 0] = torch.as_tensor(query_sequence, dtype=torch.long, device=device)
    tables = protein_tables(Q, Q_bar, device=device)
    # the divisor as the reference forms it: a float32 tensor of ones plus the penalty
    div = float((torch.ones(1) + penalty_value).item())
    steps = torch.linspace(timesteps - 1, 1, int((timesteps - 1) / 1), dtype=int).tolist()
    with torch.no_grad():
        for i, t in enumerate(steps):
            tt = torch.full((sample.shape[0],), t, dtype=torch.long, device=device)
            predictions = model(sample, tt)
            nxt = reverse_step(tables, predictions.reshape(sample.shape[0], n_sequences * seq_length, -1),
                               sample.reshape(sample.shape[0], -1), tt, None if noise_fn is None else noise_fn(i),
                               last_class_div=div, exclude_last_prefix=seq_length).reshape(sample.shape)
            if start_query:
                nxt[:, 0] = sample[:, 0]
            sample = nxt
    flat = sample.reshape(sample.shape[0], -1).cpu()
    return sample, [[tokenizer.untokenize(s)] for s in flat]


def sample_and_save_msa(out_path, num_seqs, model, tokenizer, Q, Q_bar, timesteps, seq_len, device, batch_size,
                        penalty_value, start_query=False, query_sequence=None):
    """Generate `batch_size` MSAs of `num_seqs` sequences and append them to
    ``generated_msas.a3m`` in the reference's record format (generate.py:133-150)."""
    sample, string = generate_msa_d3pm(model, batch_size, num_seqs, seq_len, device, timesteps, penalty_value, Q_bar, Q,
                                       tokenizer, start_query=start_query, query_sequence=query_sequence)
    with open(out_path + "generated_msas.a3m", "a") as f:
        for msa in string:
            for seq in range(num_seqs):
                seq_string = str(msa[0][seq * seq_len:(seq + 1) * seq_len]).replace("!", "")
                f.write((">MSA_0" if seq == 0 else ">tr ") + "\n" + seq_string + "\n")
    return sample, string
