"""Oracle (test infrastructure only) for the protein-style (EvoDiff-like) D3PM.

Functional restatement, in eager PyTorch-CPU ops with the reference's dtypes
(float32 betas, float64 after the softmax) and its per-sequence Python loops,
of ``protein_generation/utils.py`` (schedule, tables, forward corruption),
``protein_generation/losses.py`` (LVB loss, single-sequence and MSA variants)
and the step maths of ``protein_generation/generate.py``.

``torch.multinomial(P, 1)`` on CPU draws ``E = empty_like(P).exponential_(1)``
and returns ``argmax(P / E)``; the samplers here take ``E`` explicitly so that
a recorded draw can be replayed.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch

# protein_generation/constants.py:1-10
ALPHABET = "ACDEFGHIKLMNPQRSTVWYX-#!"   # PROTEIN_ALPHABET, 24 symbols
K_MSA = 22                              # len(MSA_AAS): 20 aa + X + gap


def beta_schedule(num_steps: int, start: float = 1e-5, end: float = 0.999,
                  schedule: str = "linear") -> torch.Tensor:
    """utils.py:28-41 (float32 throughout)."""
    if schedule == "linear":
        return torch.linspace(start, end, num_steps)
    if schedule == "cosine":
        b = torch.linspace(np.pi / 2, 0, num_steps)
        return torch.cos(b) * (end - start) + start
    if schedule == "sohl-dickstein":
        b = torch.linspace(0, num_steps - 1, num_steps)
        return 1 / (num_steps - b + 1)
    raise NotImplementedError(schedule)


def running_products(mats: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    """utils.py:7-18 (``cumprod_matrix``)."""
    out = [mats[0]]
    acc = mats[0]
    for i in range(len(mats) - 1):
        acc = torch.mm(acc, mats[i + 1])
        out.append(acc)
    return out


def uniform_tables(K: int = K_MSA, num_steps: int = 500, schedule: str = "sohl-dickstein"):
    """utils.py:65-83 (``Tokenizer.q_uniform_schedule``): returns (Q_bar, Q_t),
    both [T,K,K] float64; the off-diagonal is float32 arithmetic promoted."""
    betas = beta_schedule(num_steps=num_steps, schedule=schedule)
    steps = []
    for i in range(len(betas)):
        off = torch.ones((K, K)) / K * betas[i]
        norm = 1 - off.sum(axis=0)
        steps.append(torch.tensor(np.identity(K)) * norm + off)
    q_bar = torch.stack(running_products(steps))
    return q_bar, torch.stack(steps)


def tokenize(seq: str) -> np.ndarray:
    """utils.py:85-86."""
    lut = {c: i for i, c in enumerate(ALPHABET)}
    return np.array([lut[c] for c in seq])


def untokenize(tokens) -> str:
    """utils.py:88-92."""
    return "".join(ALPHABET[int(i)] for i in tokens)


def one_hot(tokens: torch.Tensor, K: int = K_MSA) -> torch.Tensor:
    """utils.py:94-97."""
    return torch.nn.functional.one_hot(tokens, num_classes=K).to(torch.double)


def corrupt(x0_onehot: torch.Tensor, q_bar_t: torch.Tensor, expo: torch.Tensor):
    """utils.py:44-52 (``sample_transition_matrix``) with the exponential draw
    of ``torch.multinomial`` passed in: returns (x_t [L], probs [L,K])."""
    p = torch.mm(x0_onehot, q_bar_t)
    return torch.argmax(p / expo, dim=1), p


def _posterior_pieces(pred64, x_t_oh, x0_oh, Qt, Qm):
    """losses.py:70-76 / generate.py:34-38: A, B and the p-squared marginal."""
    K = Qt.shape[0]
    A = torch.mm(x_t_oh, torch.t(Qt))
    Bm = torch.mm(x0_oh, Qm) if x0_oh is not None else None
    Qe = Qm.unsqueeze(0).expand(A.shape[0], K, K)
    q_t = torch.mul(A.unsqueeze(1), torch.mul(pred64.unsqueeze(2), Qe))
    marg = torch.bmm(torch.transpose(q_t, 1, 2), pred64.unsqueeze(2)).squeeze()
    return A, Bm, marg


def lvb_per_sequence(src_onehot, q, predictions, tgt, tgt_onehot, timestep, Q, Q_bar,
                     K: int = K_MSA, tmax: int = 500, msa: bool = False) -> torch.Tensor:
    """losses.py:48-90 (and :120-153 with ``msa=True``): one loss per batch
    element, branch on t==1 / t==tmax / else."""
    kl = torch.nn.KLDivLoss(reduction="batchmean", log_target=False)
    ce = torch.nn.CrossEntropyLoss(weight=None, reduction="mean")
    p = torch.nn.functional.softmax(predictions[..., :K], dim=-1)
    out = []
    for i in range(tgt.shape[0]):
        ti = timestep[i]
        if ti == 1:
            out.append(ce(predictions[i].reshape(-1, predictions.shape[-1]), tgt[i].reshape(-1)))
        elif ti == tmax:
            prior = torch.ones_like(torch.empty(*q[i].shape)) / K
            out.append(kl(prior.log(), q[i]))
        else:
            pred = p[i].reshape(-1, K).to(torch.float64)
            x_t = src_onehot[i][..., :K].reshape(-1, K)
            x_0 = tgt_onehot[i][..., :K].reshape(-1, K)
            A, Bm, marg = _posterior_pieces(pred, x_t, x_0, Q[ti], Q_bar[ti - 1])
            num = torch.mul(A, Bm)
            Qden = Q_bar[ti - 1] if msa else Q_bar[ti]      # losses.py:148 vs :79
            den = torch.bmm(torch.mm(x_0, Qden).unsqueeze(1), x_t.unsqueeze(2))
            post = num / den.squeeze().unsqueeze(1)
            marg = marg / marg.sum(dim=1, keepdim=True)
            out.append(kl(marg.log(), post))
    return torch.stack(out)


def lvb(src_onehot, q, predictions, tgt, tgt_onehot, timestep, Q, Q_bar, **kw) -> torch.Tensor:
    """losses.py:88-93: sum of the per-sequence losses over the batch size."""
    per = lvb_per_sequence(src_onehot, q, predictions, tgt, tgt_onehot, timestep, Q, Q_bar, **kw)
    return per.sum() / tgt.shape[0]


def driver_loss(src_onehot, q, predictions, tgt, tgt_onehot, timestep, Q, Q_bar, lam: float, **kw):
    """Protein Generation Notebook.ipynb cell-12 / cell-13, the loss the training loop differentiates:
    ``(lvb + lam * ce) * n_tokens`` with ``ce = CrossEntropyLoss(mean)(outputs.reshape(-1, n), tgt.reshape(-1))``,
    both terms cast to float32 and ``n_tokens = src.shape[0] * src.shape[1]`` (B*L for sequences, B*D for MSAs).
    Returns (loss, lvb, ce)."""
    lvb_loss = lvb(src_onehot, q, predictions, tgt, tgt_onehot, timestep, Q, Q_bar, **kw)
    ce_loss = torch.nn.CrossEntropyLoss(weight=None, reduction="mean")(
        predictions.reshape(-1, predictions.shape[-1]), tgt.reshape(-1))
    n_tokens = tgt.shape[0] * tgt.shape[1]
    lvb_loss = lvb_loss.to(torch.float32)
    ce_loss = ce_loss.to(torch.float32)
    return (lvb_loss + (lam * ce_loss)) * n_tokens, lvb_loss, ce_loss


def reverse_marginal(logits: torch.Tensor, x_t: torch.Tensor, Qt: torch.Tensor, Qm: torch.Tensor,
                     K: int = K_MSA) -> torch.Tensor:
    """generate.py:28-39 for one sequence: normalised p_theta(x_{t-1}|x_t)
    [L,K] float64 from denoiser logits [L,n_tokens]."""
    p = torch.nn.functional.softmax(logits[:, :K], dim=-1).to(torch.float64)
    _, _, marg = _posterior_pieces(p, one_hot(x_t, K), None, Qt, Qm)
    return marg / marg.sum(dim=1, keepdim=True)


def reverse_step(logits: torch.Tensor, x_t: torch.Tensor, t: int, Q: torch.Tensor, Q_bar: torch.Tensor,
                 expo: torch.Tensor, K: int = K_MSA) -> torch.Tensor:
    """generate.py:32-41 for a batch: logits [B,L,n_tokens], x_t [B,L],
    expo [B,L,K] float64 -> x_{t-1} [B,L] int64."""
    out = x_t.clone()
    for i in range(x_t.shape[0]):
        P = reverse_marginal(logits[i], x_t[i], Q[t], Q_bar[t - 1], K)
        out[i] = torch.argmax(P / expo[i], dim=1)
    return out


def generate(model, Q, Q_bar, timesteps: int, x_init: torch.Tensor, expo_fn, K: int = K_MSA):
    """generate.py:21-44: the full reverse loop t = T-1 .. 1 from ``x_init``;
    ``expo_fn(step, i)`` supplies the [L,K] exponential draw of sequence i."""
    sample = x_init.clone()
    B = sample.shape[0]
    ts = torch.linspace(timesteps - 1, 1, int((timesteps - 1) / 1), dtype=int)
    with torch.no_grad():
        for step, t in enumerate(ts):
            logits = model(sample, torch.tensor([t] * B))
            nxt = sample.clone()
            for i in range(B):
                P = reverse_marginal(logits[i], sample[i], Q[t], Q_bar[t - 1], K)
                nxt[i] = torch.argmax(P / expo_fn(step, i), dim=1)
            sample = nxt
    return sample, [untokenize(s) for s in sample]


def reverse_step_msa(logits: torch.Tensor, x_t: torch.Tensor, t: int, Q: torch.Tensor, Q_bar: torch.Tensor,
                     expo_all: torch.Tensor, expo_query: torch.Tensor, penalty_value: float, start_query: bool,
                     K: int = K_MSA) -> torch.Tensor:
    """generate.py:70-93 for a batch of MSAs: logits [B,D,L,n_tokens], x_t
    [B,D,L]; ``expo_all`` [B,D*L,K] and ``expo_query`` [B,L,K-1] are the draws of
    the two ``torch.multinomial`` calls.  The gap class is divided by
    1 + penalty (a float32 tensor in the reference) after normalisation; the
    query row is drawn without the gap class; with ``start_query`` it is kept."""
    B, D, L = x_t.shape
    out = x_t.clone()
    for i in range(B):
        P = reverse_marginal(logits[i].flatten(0, 1), x_t[i].flatten(), Q[t], Q_bar[t - 1], K)
        penalty = torch.ones(P.shape)
        penalty[:, -1] += penalty_value
        P = P / penalty
        tmp = torch.argmax(P / expo_all[i], dim=1)
        tmp[:L] = torch.argmax(P[:L, :-1] / expo_query[i], dim=1)
        tmp = tmp.reshape(D, L)
        if start_query:
            out[i, 1:, :] = tmp[1:, :]
        else:
            out[i] = tmp
    return out


def generate_msa(model, Q, Q_bar, timesteps: int, x_init: torch.Tensor, expo_fn, penalty_value: float = 0.0,
                 start_query: bool = False, K: int = K_MSA):
    """generate.py:48-99: the MSA reverse loop from ``x_init`` [B,D,L];
    ``expo_fn(step) -> (expo_all [B,D*L,K], expo_query [B,L,K-1])``."""
    sample = x_init.clone()
    B = sample.shape[0]
    ts = torch.linspace(timesteps - 1, 1, int((timesteps - 1) / 1), dtype=int)
    with torch.no_grad():
        for step, t in enumerate(ts):
            logits = model(sample, torch.tensor([t] * B))
            ea, eq = expo_fn(step)
            sample = reverse_step_msa(logits, sample, int(t), Q, Q_bar, ea, eq, penalty_value, start_query, K)
    return sample, [[untokenize(s.flatten())] for s in sample]


def collate_v2(tokens: torch.Tensor, t: int, num_steps: int, Q_bar: torch.Tensor, expo: torch.Tensor, K: int = K_MSA):
    """utils.py:262-293 (``CollaterV2.__call__``) on tokenised sequences
    [B,L] with the multinomial draws [B,L,K] passed in: returns (src [B,L],
    timesteps [B], q_x [B,L,K]).  The loop variable is decremented after the
    first sequence's timestep has been recorded when t == num_steps."""
    src, ts, qx = [], [], []
    for i in range(tokens.shape[0]):
        ts.append(t)
        if t == num_steps:
            t = t - 1
        x_t, p = corrupt(one_hot(tokens[i], K), Q_bar[t], expo[i])
        src.append(x_t)
        qx.append(p)
    return torch.stack(src).to(torch.long), torch.tensor(ts, dtype=torch.long), torch.stack(qx).to(torch.double)


def collate(tokens: torch.Tensor, num_steps: int, Q_bar: torch.Tensor, expo: torch.Tensor, K: int = K_MSA):
    """dataset.py:219-250 (``Collater.__call__``) on tokenised sequences [B,L]: per sequence, in order, one
    ``np.random.randint(1, num_steps)`` (numpy's global generator) and one corruption with Q_bar[t]; the multinomial
    draws [B,L,K] are passed in.  Returns (src [B,L], src_one_hot, timesteps [B], tokenized, one_hot, q_x)."""
    src, ts, qx = [], [], []
    for i in range(tokens.shape[0]):
        t = np.random.randint(1, num_steps)
        ts.append(t)
        x_t, p = corrupt(one_hot(tokens[i], K), Q_bar[t], expo[i])
        src.append(x_t)
        qx.append(p)
    src = torch.stack(src).to(torch.long)
    return (src, one_hot(src, K), torch.tensor(ts, dtype=torch.long), tokens.to(torch.long), one_hot(tokens, K),
            torch.stack(qx).to(torch.double))


def collate_msa(tokens: torch.Tensor, num_steps: int, Q_bar: torch.Tensor, expo: torch.Tensor, K: int = K_MSA):
    """dataset.py:262-300 (``CollaterMSA.__call__``) on tokenised MSAs [B,D,L]: one timestep per MSA, the flattened
    [D*L] rows corrupted with it (draws [B,D*L,K] passed in); every MSA's probabilities are appended twice
    (:287, :291), so q_x is [2B,D,L,K].  Returns (src [B,D,L], src_one_hot, timesteps [B], tokenized, tgt_one_hot,
    q_x)."""
    B, D, L = tokens.shape
    src, ts, qx = [], [], []
    for i in range(B):
        t = np.random.randint(1, num_steps)
        ts.append(t)
        x_t, p = corrupt(one_hot(tokens[i], K).flatten(0, 1), Q_bar[t], expo[i])
        src.append(x_t.reshape(D, L))
        qx.append(p.reshape(D, L, K))
        qx.append(p.reshape(D, L, K))
    src = torch.stack(src).to(torch.long)
    return (src, one_hot(src, K), torch.tensor(ts), tokens.to(torch.long), one_hot(tokens, K),
            torch.stack(qx).to(torch.double))


def mutant_scores(tokens: torch.Tensor, model, Q, Q_bar, steps: int, n_iter: int, minibatch: int, order_fn, t_fn,
                  expo_fn, K: int = K_MSA) -> np.ndarray:
    """utils.py:152-222 (the scoring loop of ``custom_mutant_matrix``) on
    tokenised sequences [N,L], wild type first: returns delta ELBOs [N-1].
    ``order_fn(i)`` is the shuffled order of repetition i, ``t_fn(i, j)`` the
    random middle timestep, ``expo_fn(i, j, h)`` the corruption draws."""
    n = tokens.shape[0]
    pm = np.zeros((n, n_iter))
    with torch.no_grad():
        for i in range(n_iter):
            order = order_fn(i)
            for j in range(0, n, minibatch):
                idx = order[j:j + minibatch]
                batch = tokens[idx]
                losses = torch.zeros(len(idx))
                for h in range(3):
                    t = 1 if h == 0 else (t_fn(i, j) if h == 1 else steps)
                    src, ts, q = collate_v2(batch, t, steps, Q_bar, expo_fn(i, j, h), K)
                    src_oh, tgt_oh = one_hot(src, K), one_hot(batch, K)
                    outputs = torch.zeros_like(src_oh) if t == steps else model(src, ts)
                    l = lvb_per_sequence(src_oh, q, outputs, batch, tgt_oh, ts, Q, Q_bar, K=K, tmax=steps).to(torch.float32)
                    if h == 1:
                        l = l * (steps - 1)
                    losses += l
                pm[idx, i] = losses.numpy()
    mean = pm.mean(axis=1)
    return -(mean[1:] - mean[0])
