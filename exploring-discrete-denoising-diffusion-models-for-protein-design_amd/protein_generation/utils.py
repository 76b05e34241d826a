"""Schedules, tables, tokenizer and forward corruption of the protein-style
process, with the names and return conventions of
``protein_generation/utils.py`` (:7-102).  Tables are built and multiplied on
the device; forward corruption is one batched kernel launch instead of a
per-sequence ``torch.multinomial``."""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib
from .._ops import ops, i64
from ..tables import ProteinTables
from .constants import PROTEIN_ALPHABET, MSA_AAS, ALL_AAS, GAP


def _default_device():
    if not torch.cuda.is_available():
        raise RuntimeError("the protein-style kernels need a HIP device; there is no CPU path in this package")
    return torch.device("cuda", torch.cuda.current_device())


def cumprod_matrix(a):
    """Running products a[0], a[0]a[1], ... (utils.py:7-18).  `a` is a list of
    [K,K] matrices or a [T,K,K] tensor; returns a list of T device tensors."""
    stack = a if torch.is_tensor(a) else torch.stack(list(a))
    dev = stack.device if stack.is_cuda else _default_device()
    stack = stack.to(device=dev, dtype=torch.float64)
    return list(ops.cumprod(stack).unbind(0))


def _beta_schedule(num_steps, start=1e-5, end=0.999, schedule="linear"):
    """float32 beta schedules (utils.py:28-41); host-side, T scalars."""
    if schedule == "linear":
        return torch.linspace(start, end, num_steps)
    if schedule == "cosine":
        return torch.cos(torch.linspace(np.pi / 2, 0, num_steps)) * (end - start) + start
    if schedule == "sohl-dickstein":
        return 1 / (num_steps - torch.linspace(0, num_steps - 1, num_steps) + 1)
    raise NotImplementedError(schedule)


def sample_transition_matrix(x_0, Q_bar, noise=None):
    """x_t ~ Categorical(x_0 @ Q_bar) for one sequence (utils.py:44-52):
    returns (sample [L] int64, probabilities [L,K] float64).  ``x_0`` is the
    one-hot [L,K] float64 the reference passes, ``Q_bar`` one [K,K] matrix.
    ``noise`` optionally supplies the Exp(1) draw [L,K] of torch.multinomial."""
    dev = x_0.device if x_0.is_cuda else _default_device()
    x0 = x_0.to(dev).argmax(dim=-1).unsqueeze(0)                       # [1, L]
    Qb = Q_bar.to(device=dev, dtype=torch.float64).unsqueeze(0)        # T = 1
    tb = ProteinTables(Qb, Qb)
    s, p = corrupt_batch(tb, x0, torch.zeros(1, dtype=torch.int64, device=dev),
                         None if noise is None else noise.unsqueeze(0), want_probs=True)
    return s[0], p[0]


def corrupt_batch(tables: ProteinTables, x0, t, noise=None, want_probs=False):
    """Batched forward corruption: x0 [B,L] int64, t [B] int64 ->
    (x_t [B,L] int64, q(x_t|x0) [B,L,K] float64 or None)."""
    from ..d3pm.diffusion_discrete import _philox_ticket
    x0 = x0.to(torch.int64).contiguous()
    t = t.to(device=x0.device, dtype=torch.int64).contiguous()
    _lib.require_device(x0)
    B = x0.shape[0]
    npos = x0.numel() // max(B, 1)
    seed = off = 0
    if noise is None:
        seed, off = _philox_ticket(x0.device)
    else:
        noise = noise.to(device=x0.device, dtype=torch.float64).contiguous()
        assert noise.shape == x0.shape + (tables.K,), noise.shape
    x_t, probs = ops.corrupt(*tables.op_args(), x0, t, noise, i64(seed), i64(off), bool(want_probs))
    return x_t, (probs if want_probs else None)


class Tokenizer(object):
    """Character <-> index maps and the uniform transition schedule
    (utils.py:55-102)."""

    def __init__(self, protein_alphabet=PROTEIN_ALPHABET, all_aas=MSA_AAS, gap=GAP):
        self.alphabet = list("".join(protein_alphabet))
        self.all_aas = list("".join(all_aas))
        self.gap = gap
        self.a_to_i = {u: i for i, u in enumerate(self.alphabet)}
        self.i_to_a = np.array(self.alphabet)
        self.K = len(self.all_aas)

    def q_uniform_schedule(self, num_steps=500, schedule="sohl-dickstein", device=None):
        """(Q_prod, Q_t), both [T,K,K] float64 on the device (utils.py:65-83).

        The two float32 scalars per step of the reference (the off-diagonal
        ones/K*beta_t and 1 - its column sum) are evaluated with the same
        tensor expression on the host so that their float32 rounding is the
        reference's; the matrices and their running products are built on the
        device."""
        dev = torch.device(device) if device is not None else _default_device()
        betas = _beta_schedule(num_steps=num_steps, schedule=schedule)
        K, T = self.K, len(betas)
        col = torch.ones((K, K)) / K * betas.view(T, 1, 1)           # [T,K,K] float32
        off = col[:, 0, 0].contiguous()
        norm = torch.stack([1 - col[i].sum(axis=0)[0] for i in range(T)])
        Q_t, Q_prod = ops.protein_uniform_tables(off.to(dev), norm.contiguous().to(dev), K)
        return Q_prod, Q_t

    def tokenize(self, seq):
        return np.array([self.a_to_i[a] for a in seq])

    def untokenize(self, x):
        if torch.is_tensor(x):
            x = x.detach().cpu().tolist()
        return "".join(self.i_to_a[int(i)] for i in x)

    def one_hot(self, tokenized):
        return torch.nn.functional.one_hot(tokenized, num_classes=self.K).to(torch.double)

    def undo_one_hot(self, x_onehot):
        return [np.where(r == 1)[0] for r in x_onehot]


class CollaterV2(object):
    """Collater with a caller-chosen timestep (utils.py:240-293), used by the
    mutant scoring: __call__(data, t, noise=None) with data = [sequence, ...]
    returns (src, src_one_hot, timesteps, tokenized, one_hot, Q, Q_bar, q_x), or
    (src, timesteps, tokenized) when ``lean``.

    t == num_steps corrupts with Q_bar[t-1].  The reference decrements its loop
    variable after recording the first sequence's timestep (utils.py:277-280),
    so it reports [t, t-1, t-1, ...]; ``reference_timesteps=True`` (the default)
    keeps that, False reports t for every sequence."""

    def __init__(self, tokenizer=None, num_steps=500, Q=None, Q_bar=None, device=None, lean=False,
                 reference_timesteps=True):
        self.tokenizer = tokenizer or Tokenizer()
        self.num_steps = num_steps
        self.K = self.tokenizer.K
        self.Q = Q
        self.Q_bar = Q_bar
        self.device = torch.device(device) if device is not None else None
        self.lean = lean
        self.reference_timesteps = reference_timesteps

    def __call__(self, data, t, noise=None):
        from ..tables import protein_tables
        dev = self.device if self.device is not None else (self.Q_bar.device if self.Q_bar.is_cuda else _default_device())
        tok = np.stack([self.tokenizer.tokenize(s) for s in data]).astype(np.int64)
        tokenized = torch.from_numpy(tok).to(dev)
        B = tokenized.shape[0]
        tc = t - 1 if t == self.num_steps else t
        reported = [t] + [tc if self.reference_timesteps else t] * (B - 1)
        timesteps = torch.tensor(reported, dtype=torch.long, device=dev)
        tables = protein_tables(self.Q, self.Q_bar, device=dev)
        src, q_x = corrupt_batch(tables, tokenized, torch.full((B,), tc, dtype=torch.long, device=dev), noise,
                                 want_probs=not self.lean)
        if self.lean:
            return src, timesteps, tokenized
        return (src, self.tokenizer.one_hot(src), timesteps, tokenized, self.tokenizer.one_hot(tokenized), self.Q,
                self.Q_bar, q_x)


def parse_mutants(input_path, focus_seq, focus_seq_name, focus_seq_trimmed):
    """Sequences and descriptors of the valid mutants listed in a CSV
    (utils.py:107-150): first line a header, first field of every other line
    'A24C' or 'A24C:D30E' (wild type, 1-based position in the full protein,
    mutant).  Upper-case letters of ``focus_seq`` are the modelled columns; the
    name ends in '/<start>-<end>'.  Returns (sequences, descriptors) with the
    wild type first."""
    start_idx = int(focus_seq_name.split("/")[-1].split("-")[0])
    table = {}
    focus_idx = 0
    for i, letter in enumerate(focus_seq):
        if letter == letter.upper():
            for mut in ALL_AAS:
                if letter != mut:
                    table[letter + str(start_idx + i) + mut] = focus_idx
            focus_idx += 1
    sequences = ["".join(focus_seq_trimmed)]
    descriptors = ["wt"]
    with open(input_path, "r") as f:
        for i, line in enumerate(f):
            if i == 0:
                continue
            mutants = line.rstrip().split(",")[0].split(":")
            if all(m in table for m in mutants):
                seq = list(focus_seq_trimmed)
                for m in mutants:
                    seq[table[m]] = m[-1]
                sequences.append("".join(seq))
                descriptors.append(":".join(mutants))
    return np.array(sequences), descriptors


def custom_mutant_matrix(input_path, model, device, focus_seq, focus_seq_name, focus_seq_trimmed, tokenizer, loss_func1,
                         loss_func2, _lambda, timesteps, Q, Q_bar, N_pred_iterations=10, minibatch_size=2000,
                         filename_prefix="", offset=0, noise_fn=None):
    """ELBO difference of every listed mutant against the wild type
    (utils.py:105-236): per repetition and mini-batch three evaluations of the
    per-sequence LVB loss (t = 1; t ~ randint(2, T) weighted by T-1; t = T with
    zero logits), averaged over ``N_pred_iterations`` repetitions.  Returns
    (descriptors, delta_elbos), or writes ``<prefix>_samples-<N>_elbo_predictions.csv``.

    ``loss_func1`` is a per-sequence loss (``D3PMLVBLoss(evaluation=True)``);
    ``loss_func2``, ``_lambda`` and ``offset`` are accepted and unused, as in
    the reference.  The shuffle and the random t come from numpy's global RNG in
    the reference's order; ``noise_fn(iteration, batch_start, h) -> [B,L,K]``
    replays the corruption draws.  Tokens are corrupted and scored on the
    device; nothing but the strings is prepared on the host."""
    sequences, descriptors = parse_mutants(input_path, focus_seq, focus_seq_name, focus_seq_trimmed)
    n = sequences.shape[0]
    prediction_matrix = np.zeros((n, N_pred_iterations))
    batch_order = np.arange(n)
    collater = CollaterV2(tokenizer=tokenizer, num_steps=timesteps, Q=Q, Q_bar=Q_bar, device=device)
    steps = timesteps
    model.eval()
    with torch.no_grad():
        for i in range(N_pred_iterations):
            np.random.shuffle(batch_order)
            for j in range(0, n, minibatch_size):
                batch_index = batch_order[j:j + minibatch_size]
                batch = sequences[batch_index]
                losses = torch.zeros(batch_index.shape[0], device=device)
                for h in range(3):
                    t = 1 if h == 0 else (np.random.randint(2, steps) if h == 1 else steps)
                    src, src_onehot, ts, tgt, tgt_onehot, Qc, Qbc, q = collater(
                        batch, t, None if noise_fn is None else noise_fn(i, j, h))
                    outputs = torch.zeros_like(src_onehot) if t == steps else model(src, ts)
                    lvb_loss = loss_func1(src_onehot, q, outputs, tgt, tgt_onehot, ts, Qc, Qbc).to(torch.float32)
                    if h == 1:
                        lvb_loss = lvb_loss * (steps - 1)
                    losses += lvb_loss
                prediction_matrix[batch_index, i] = losses.cpu().numpy()
    mean_elbos = np.mean(prediction_matrix, axis=1).flatten().tolist()
    wt_elbo = mean_elbos.pop(0)
    descriptors = descriptors[1:]
    delta_elbos = -(np.asarray(mean_elbos) - wt_elbo)        # the losses are negative ELBOs
    if filename_prefix == "":
        return descriptors, delta_elbos
    with open(filename_prefix + "_samples-" + str(N_pred_iterations) + "_elbo_predictions.csv", "w") as f:
        for d, v in zip(descriptors, delta_elbos):
            f.write(d + ";" + str(v) + "\n")
