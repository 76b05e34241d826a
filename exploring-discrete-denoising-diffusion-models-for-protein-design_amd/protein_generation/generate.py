"""Reverse sampling of the protein-style D3PM with the signatures of
``protein_generation/generate.py`` (:8-44, :118-130).  The per-step posterior
marginal and categorical draw of the whole batch is one kernel launch; the
reference loops over sequences in Python and calls ``torch.multinomial`` per
sequence."""
from __future__ import annotations

import torch

from .. import _lib
from .._lib import lib, check, ptr, stream_ptr
from ..tables import protein_tables


def reverse_step(tables, predictions, x_t, t, noise=None):
    """x_{t-1} for a batch (generate.py:28-41): predictions [B,L,n_tokens]
    float32 logits, x_t [B,L] int64, t [B] int64; ``noise`` optionally the
    Exp(1) draws [B,L,K] float64 of the reference's multinomial."""
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
    if noise is None:
        seed, off = _philox_ticket(x_t.device)
    else:
        noise = noise.to(device=x_t.device, dtype=torch.float64).contiguous()
        assert noise.shape == x_t.shape + (tables.K,), noise.shape
    out = torch.empty_like(x_t)
    with torch.cuda.device(x_t.device):
        check(lib.d3pm_lvb_reverse_step(tables.desc(), ptr(pr), pr.shape[-1], ptr(x_t), ptr(t), B, npos, ptr(noise),
                                        seed, off, ptr(out), stream_ptr()))
    return out


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
