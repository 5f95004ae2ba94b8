"""Oracle, multi-threaded legs: the O(T*V) part of the learner loss (logits -> log-probabilities and
entropy -> d loss / d logits) restated with torch CPU ops and autograd, the way the reference
itself executes on a host (fp32 torch kernels over all intra-op threads).  The O(T) token math is
the numpy restatement in `oracle/rl_loss.py` (one source of truth).

Follows reference pipelinerl/finetune/rl/__init__.py:204-233 (temperature divide, gather minus
logsumexp, entropy; the entropy is computed without a graph in 4096-row chunks when no entropy
bonus is configured) and lets autograd produce what the closed form of `rl_loss.rl_step` produces.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ and by bench.py's `cpu_baseline`.
Pinned against the reference's own outputs in tests/test_oracle_golden.py.
"""

from __future__ import annotations

from typing import Any

import numpy as np
import torch

from . import rl_loss as _np_oracle

ENTROPY_CHUNK = 4096  # rl/__init__.py:219


def _entropy(z: torch.Tensor) -> torch.Tensor:
    """H = logsumexp(z) - sum softmax(z) * z over the vocabulary axis."""
    lse = torch.logsumexp(z, dim=-1)
    return lse - (torch.softmax(z, dim=-1) * z).sum(dim=-1)


def rl_step(logits: np.ndarray | torch.Tensor, batch: dict[str, np.ndarray], config: Any, current_step: int, max_step: int,
            is_packed: bool, want_grad: bool = True, value: np.ndarray | None = None) -> dict[str, Any]:
    """Same contract as `oracle.rl_loss.rl_step`; `grad_logits` is a torch tensor [B, L, V]."""
    temperature = float(_np_oracle._cfg(config, "temperature", 1.0))
    use_entropy = _np_oracle._cfg(config, "entropy_bonus", 0.0) != 0.0 or _np_oracle._cfg(config, "final_entropy_bonus", 0.0) != 0.0
    lg = torch.as_tensor(logits, dtype=torch.float32).detach().requires_grad_(want_grad)
    ids = torch.as_tensor(np.ascontiguousarray(batch["input_ids"][:, 1:]), dtype=torch.int64)
    z = lg[:, :-1, :] / temperature
    nlp = z.gather(-1, ids.unsqueeze(-1)).squeeze(-1) - torch.logsumexp(z, dim=-1)
    if use_entropy:
        ent = _entropy(z)
    else:
        with torch.no_grad():
            flat = z.reshape(-1, z.shape[-1])
            ent = torch.cat([_entropy(flat[i : i + ENTROPY_CHUNK]) for i in range(0, flat.shape[0], ENTROPY_CHUNK)]).reshape(nlp.shape)
    res = _np_oracle.token_loss(batch, nlp.detach().numpy(), ent.detach().numpy(), config, current_step, max_step, is_packed, value=value)
    res.update(new_logprobs=nlp.detach().numpy(), entropy=ent.detach().numpy())
    if want_grad:
        heads, seeds = [nlp], [torch.from_numpy(np.ascontiguousarray(res["g_nlp"]))]
        if use_entropy:
            heads.append(ent)
            seeds.append(torch.from_numpy(np.ascontiguousarray(res["g_ent"])))
        torch.autograd.backward(heads, seeds)
        res["grad_logits"] = lg.grad
    return res


def rl_step_closed_form(logits: np.ndarray | torch.Tensor, batch: dict[str, np.ndarray], config: Any, current_step: int,
                        max_step: int, is_packed: bool, value: np.ndarray | None = None) -> dict[str, Any]:
    """Same result without autograd: vectorised multi-threaded torch CPU kernels for the O(T*V)
    passes and the closed-form d loss / d logits of SURVEY.md App. A (what a tuned host
    implementation would run; `bench.py` times this one as the CPU baseline).

        dz = g (1[v == id] - p) - g_H p (log p + H),   d logits = dz / temperature
    """
    temperature = float(_np_oracle._cfg(config, "temperature", 1.0))
    use_entropy = _np_oracle._cfg(config, "entropy_bonus", 0.0) != 0.0 or _np_oracle._cfg(config, "final_entropy_bonus", 0.0) != 0.0
    with torch.no_grad():
        lg = torch.as_tensor(logits, dtype=torch.float32)
        ids = torch.as_tensor(np.ascontiguousarray(batch["input_ids"][:, 1:]), dtype=torch.int64).unsqueeze(-1)
        z = lg[:, :-1, :] / temperature
        lse = torch.logsumexp(z, dim=-1)
        nlp = z.gather(-1, ids).squeeze(-1) - lse
        p = torch.softmax(z, dim=-1)
        ent = lse - (p * z).sum(dim=-1)
        res = _np_oracle.token_loss(batch, nlp.numpy(), ent.numpy(), config, current_step, max_step, is_packed, value=value)
        g = torch.from_numpy(np.ascontiguousarray(res["g_nlp"]))
        grad = torch.zeros_like(lg)
        dz = grad[:, :-1, :]
        if use_entropy:
            gh = torch.from_numpy(np.ascontiguousarray(res["g_ent"]))
            z.sub_(lse.unsqueeze(-1)).add_(ent.unsqueeze(-1))            # log p + H
            torch.mul(p, z, out=z)                                          # p (log p + H)
            dz.copy_(z.mul_(-gh.unsqueeze(-1)))
            dz.addcmul_(p, g.unsqueeze(-1), value=-1.0)
        else:
            torch.mul(p, -g.unsqueeze(-1), out=dz)
        dz.scatter_add_(-1, ids, g.unsqueeze(-1))
        if temperature != 1.0:
            dz.div_(temperature)
    res.update(new_logprobs=nlp.numpy(), entropy=ent.numpy(), grad_logits=grad)
    return res
