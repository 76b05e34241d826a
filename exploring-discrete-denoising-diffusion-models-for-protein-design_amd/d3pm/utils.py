"""Stand-alone numerics helpers with the names of the reference's
``d3pm/utils.py``.  Inside ``DiscreteDiffusion`` their work is fused into the
HIP kernels; these versions exist for callers that use them directly and run as
ordinary tensor expressions on whatever device the inputs live on."""
import torch
import torch.nn.functional as F


def normalize_data(x, mode=None):
    """d3pm/utils.py:6-12."""
    if mode in (None, "rgb"):
        return x / 127.5 - 1.0
    if mode == "rgb_unit_var":
        return 2.0 * (x / 127.5 - 1.0)
    raise NotImplementedError(mode)


def log_min_exp(a, b, epsilon=1.0e-6):
    """log(exp(a) - exp(b)) for b < a (d3pm/utils.py:15-18)."""
    return a + torch.log1p(epsilon - torch.exp(b - a))


def sample_categorical(logits, uniform_noise):
    """Gumbel-max sample, returned one-hot (d3pm/utils.py:21-38)."""
    u = uniform_noise.clamp(min=torch.finfo(uniform_noise.dtype).tiny, max=1.0)
    idx = (logits - torch.log(-torch.log(u))).argmax(dim=-1)
    return F.one_hot(idx, num_classes=logits.size(-1))


def categorical_kl_logits(logits1, logits2, eps=1.0e-6):
    """KL(C(logits1) || C(logits2)) over the last axis (d3pm/utils.py:41-58)."""
    la, lb = F.log_softmax(logits1 + eps, dim=-1), F.log_softmax(logits2 + eps, dim=-1)
    return (F.softmax(logits1 + eps, dim=-1) * (la - lb)).sum(dim=-1)


def categorical_kl_probs(probs1, probs2, eps=1.0e-6):
    """KL between probability vectors (d3pm/utils.py:61-75)."""
    return (probs1 * ((probs1 + eps).log() - (probs2 + eps).log())).sum(dim=-1)


def categorical_log_likelihood(x, logits):
    """log softmax(logits)[x] (d3pm/utils.py:78-92)."""
    return F.log_softmax(logits, dim=-1).gather(-1, x.to(torch.int64).unsqueeze(-1)).squeeze(-1)


def meanflat(x):
    """Mean over every axis but the first (d3pm/utils.py:95-97)."""
    return x.flatten(1).mean(dim=1) if x.dim() > 1 else x
