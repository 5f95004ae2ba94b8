"""The reference-generated `rl_step` goldens that exercise the two branches the fused output head did not serve before round 6 - the
sequence-level GSPO term (c11, c12, c21; rl/__init__.py:310-352) and the value head (c18-c23; :265-272, 367-381, 441-448) - driven
THROUGH the fused head (`fused_head_loss`: hidden states -> MFMA head -> token loss, no `[T, V]` logits).

A golden gives logits, not hidden states.  The head computes hidden @ W^T, so the test hands it hidden = the identity (row t is the
unit vector e_t, exact in bf16) and W[v, t] = logits[t, v]: the product IS the golden's logits (to the head's two-bf16-plane weight
split, 2^-17 relative - the oracle, pinned to the reference by the same goldens, is evaluated on exactly those values too), and
d loss / d W transposed IS d loss / d logits."""

import numpy as np
import pytest
import torch

from helpers import load_rl_case, rel_err
from oracle import rl_loss as orl

pytestmark = pytest.mark.gpu

FP_TOL = 1e-4
CASES = ["c11_gspo", "c12_gspo_groupnorm_sp", "c18_ppo_value_head", "c20_ppo_value_head_rewards_sp", "c21_gspo_value_head",
         "c22_ppo_value_head_groupnorm_overlong", "c23_sentinel_value_head", "c0_ppo", "c2_reinforce"]


def _two_plane(x: np.ndarray) -> np.ndarray:
    """fp32 -> the value the head's hi + lo bf16 planes represent."""
    t = torch.from_numpy(x)
    hi = t.to(torch.bfloat16).float()
    lo = (t - hi).to(torch.bfloat16).float()
    return (hi + lo).numpy()


@pytest.mark.parametrize("name", CASES)
def test_golden_through_the_fused_head(libprl, cuda_device, name):
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.fused_head import FusedLmHead, fused_head_loss

    case = load_rl_case(name)
    B, T, V = case["logits"].shape
    assert B == 1
    H = -(-T // 64) * 64
    Vp = -(-V // 64) * 64  # the head's backward contracts over whole 64-wide vocabulary tiles: the goldens' 97 entries are padded with
    logits_q = _two_plane(case["logits"])  # entries of logit -30000 (probability exactly 0: no term of the loss, the entropy or a gradient sees them)
    hidden = torch.zeros(1, T, H, dtype=torch.bfloat16, device=cuda_device)
    hidden[0, torch.arange(T), torch.arange(T)] = 1.0
    W = torch.zeros(Vp, H, dtype=torch.float32)
    W[V:, :T] = -30000.0
    W[:V, :T] = torch.from_numpy(case["logits"][0]).t()
    h = hidden.clone().requires_grad_(True)
    w = W.to(cuda_device).requires_grad_(True)
    batch = PipelineBatchEncoding(**{k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in case["batch"].items()}).to_device(cuda_device)
    cfg = RLConfig(**case["config"])
    values = torch.from_numpy(case["value"]).to(cuda_device).requires_grad_(True) if "value" in case else None
    head = FusedLmHead(w, chunk_rows=4096)
    loss, stats = fused_head_loss(h, w, head, batch, cfg, *case["steps"], values=values)
    loss.backward()
    torch.cuda.synchronize()

    want = orl.rl_step(logits_q, case["batch"], case["config"], *case["steps"], bool(case["batch"].get("is_packed", True)), value=case.get("value"))
    assert list(stats.keys()) == list(case["stats"].keys())  # the reference's dict, key for key (37 with a value head)
    assert abs(loss.item() - float(want["loss"])) <= FP_TOL * max(abs(float(want["loss"])), 1e-6) + 1e-7
    assert abs(loss.item() - case["loss"]) <= 1e-3 * max(abs(case["loss"]), 1e-6) + 1e-6  # and the golden itself, to the weight split
    for k, v in want["stats"].items():
        assert abs(float(stats[k]) - float(v)) <= FP_TOL * max(abs(float(v)), 1.0), (k, stats[k], v)
    for k in ("num_output_tokens_sum", "input_size"):
        assert stats.get(k) == case["stats"].get(k)  # (a batch without a labelled token reports `input_size` only, rl/__init__.py:388-392)
    assert float(w.grad[V:].abs().max()) == 0.0 if Vp > V else True
    got = w.grad[:V, :T].t().float().cpu().numpy()  # d loss / d logits
    scale = np.abs(want["grad_logits"]).max()
    if scale == 0:
        assert np.abs(got).max() == 0
    else:
        assert rel_err(got, want["grad_logits"][0]) <= FP_TOL, rel_err(got, want["grad_logits"][0])
        assert rel_err(got, case["grad_logits"][0]) <= 2e-3  # the reference's own autograd, to the weight split
    if values is not None:
        np.testing.assert_allclose(values.grad.cpu().numpy(), case["grad_value"], rtol=FP_TOL, atol=1e-9)
