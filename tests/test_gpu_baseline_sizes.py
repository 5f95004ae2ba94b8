"""Oracle parity AT the BASELINE sizes, ragged (round-2 review item 3) and the reference's failure behaviour
on non-finite logits (item 2).

* K5 / K6 on the SURVEY §8(d) ragged distribution at seq_length 8192 (64 rollouts, prompt U{64..512}, completion
  U{2048..8192 - P}), packed first-fit into 8192-token micro-batches the way the reference's loop does
  (preprocess.py:596-662: a micro-batch closes when the next sequence would overflow the budget), with
  seq_parallel 1 and 2: every integer field `array_equal` and every fp32 column against
  `oracle.preprocess.collate_packed`, the K5 scalars against `oracle.preprocess.sequence_scalars`.  The
  per-workgroup binary search over the sequence table and the 2-tokens-per-lane walk cross real sequence
  boundaries here (several sequences of thousands of tokens per micro-batch).
* the fused logits -> loss -> d-logits kernel (default dispatch = the row-resident kernel bench.py times) at
  T = 2048 rows x V = 152 064 against `oracle.rl_loss_torch.rl_step_closed_form`: 2047 workgroups on 256 CUs,
  eight rounds per CU, in place and out of place.
* `rl_step` raises AssertionError for a NaN in a PROMPT row like reference rl/__init__.py:213; the row skip is an
  opt-in (`RLConfig.skip_unlabelled_rows`)."""

import ctypes
import types

import numpy as np
import pytest
import torch

from oracle import preprocess as opre
from oracle import rl_loss_torch as orlt

from helpers import assert_batch_equal, rel_err

pytestmark = pytest.mark.gpu

FP_TOL = 1e-4


def _batch_to_np(batch) -> dict:
    out = {}
    for k, v in batch.model_dump().items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        elif v is not None:
            out[k] = v
    return out


def _first_fit(lengths, budget):
    """The reference's packing rule (preprocess.py:617-640): sequences in arrival order, a micro-batch is
    closed as soon as the next sequence does not fit."""
    mbs, cur, used = [], [], 0
    for i, n in enumerate(lengths):
        if cur and used + n > budget:
            mbs.append(cur)
            cur, used = [], 0
        cur.append(i)
        used += n
    if cur:
        mbs.append(cur)
    return mbs


@pytest.mark.parametrize("budget", [8192, 32768], ids=["budget8192", "budget32768"])
@pytest.mark.parametrize("seq_parallel", [1, 2])
def test_ragged_pack_at_seq8192_vs_oracle(libprl, cuda_device, seq_parallel, budget):
    from pipelinerl_amd.finetune.data import pack_prepared
    from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data_ragged
    from pipelinerl_amd.synthetic import make_ragged, ragged_to_entries

    seq = 8192
    rag_h, reasons = make_ragged(8, attempts=8, seq_length=seq, vocab=152064, seed=1236, dense=False, with_ref=True)
    raw = ragged_to_entries(rag_h, reasons)
    lengths = [len(e["input_ids"]) for e in raw]
    # odd lengths too, so that seq_parallel = 2 needs filler tokens in some micro-batches
    assert any(n % 2 for n in lengths) and min(lengths) >= seq // 4
    # budget 8192: the BASELINE packing budget (one or two rollouts per micro-batch); 32768: the same rollouts under a
    # larger `seq_length` budget, up to nine sequence boundaries inside one micro-batch
    mbs = _first_fit(lengths, budget)
    assert max(len(m) for m in mbs) >= (2 if budget == seq else 5), "the point of this test: several long sequences per micro-batch"
    rag = rag_h.to(cuda_device)
    prep = populate_rl_data_ragged(rag, 2, RLConfig(divide_advantage_by_std=False))
    data = opre.preprocess_chunk(raw, 2, False)
    want_scalars = opre.sequence_scalars(data, 2, False)
    np.testing.assert_allclose(prep.advantage64.cpu().numpy(), [s[0] for s in want_scalars], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(prep.group_tokens64.cpu().numpy(), [s[1] for s in want_scalars], rtol=1e-12)
    assert prep.overflow.cpu().tolist() == [s[2] for s in want_scalars]
    assert prep.num_labels.cpu().tolist() == [float(s[3]) for s in want_scalars]
    pads = [(seq_parallel - sum(lengths[i] for i in mb) % seq_parallel) % seq_parallel for mb in mbs]
    if seq_parallel == 2:
        assert any(pads), "at least one micro-batch must receive sequence-parallel filler tokens"
    got = pack_prepared(prep, mbs, eos_token_id=2, sentinel_pad=pads)
    assert len(got) == len(mbs)
    for mb, g in zip(mbs, got):
        want = opre.collate_packed([data[i] for i in mb], 2, seq_parallel)
        g_np = _batch_to_np(g)
        assert_batch_equal(g_np, want, float_tol=1e-6)
        for key in ("input_ids", "labels", "position_ids", "segment_ids", "attention_mask", "seq_boundaries"):
            assert np.array_equal(np.asarray(g_np[key]), want[key]), key
        # the fp32 copies (old / ref log-probs, rewards) are copies: bit for bit
        for key in ("old_logprobs", "ref_logprobs", "rewards"):
            assert np.array_equal(np.asarray(g_np[key]), np.asarray(want[key], dtype=np.float32)), key


def _big_case(T, V, cfg, device):
    """Seeded logits generated ON the device (311 M normals take a minute on the host), four packed sequences."""
    rng = np.random.default_rng(T + V)
    gen = torch.Generator(device=device).manual_seed(T + V)
    lt = torch.randn((1, T, V), generator=gen, device=device, dtype=torch.float32) * 2.0
    logits = lt.cpu().numpy()
    ids = rng.integers(3, V, size=(1, T), dtype=np.int64)
    # four packed sequences with prompts, a sprinkling of unlabelled observation tokens
    bounds = [0, T // 5, T // 2, (3 * T) // 4, T]
    pos = np.concatenate([np.arange(b - a) for a, b in zip(bounds[:-1], bounds[1:])])[None].astype(np.int64)
    seg = np.concatenate([np.full(b - a, k) for k, (a, b) in enumerate(zip(bounds[:-1], bounds[1:]))])[None].astype(np.int64)
    labels = ids.copy()
    for a in bounds[:-1]:
        labels[0, a:a + 40] = -100
    labels[0, rng.random(T) < 0.05] = -100
    # exact log-probs of the next tokens (fp64, on the device in row chunks) to place old_logprobs relative to them
    nxt = torch.from_numpy(ids[0, 1:]).to(device)
    parts = []
    for r0 in range(0, T - 1, 256):
        z = lt[0, r0:min(r0 + 256, T - 1)].double() / cfg["temperature"]
        parts.append(z.gather(-1, nxt[r0:r0 + z.shape[0], None])[:, 0] - torch.logsumexp(z, -1))
    nlp = np.concatenate([[0.0], torch.cat(parts).cpu().numpy()])
    del lt
    old = nlp + rng.normal(0, 0.01, T)
    adv = rng.normal(0, 1, T)
    for t in range(100, 400):  # PPO-clipped rows: zero token gradient unless kl / entropy terms are on
        up = t % 2 == 0
        old[t] = nlp[t] - (0.5 if up else -0.5)
        adv[t] = abs(adv[t]) + 0.1 if up else -abs(adv[t]) - 0.1
    f32 = lambda a: np.asarray(a, dtype=np.float32)[None]  # noqa: E731
    batch = {
        "input_ids": ids, "labels": labels, "position_ids": pos, "attention_mask": np.ones_like(ids), "segment_ids": seg,
        "old_logprobs": f32(old), "ref_logprobs": f32(old + rng.normal(0, 0.05, T)), "advantages": f32(adv),
        "rewards": f32(rng.integers(0, 2, T)), "group_tokens": f32(np.full(T, 37.0)),
        "num_labels": f32(np.full(T, float((labels != -100).sum()))), "overflow": f32(np.zeros(T)),
    }
    return logits, batch


@pytest.mark.parametrize("cfg_name,T", [("grpo_clip", 2048), ("kl_ent_temp", 2048), pytest.param("grpo_clip", 8192, marks=pytest.mark.slow)],
                         ids=["grpo_clip_2048_rows", "kl_ent_temp_2048_rows", "grpo_clip_8192_rows_the_benchmarked_launch"])
def test_fused_logits_loss_at_2048_rows_vs_oracle(libprl, cuda_device, monkeypatch, cfg_name, T):
    """The row-resident fused kernel vs `oracle.rl_loss_torch.rl_step_closed_form`, in and out of place.  T = 8192 x
    V = 152 064 is EXACTLY the launch bench.py times (one row per workgroup, 8191 workgroups = 32 rounds per CU); the
    oracle needs ~15 GB of host memory and a few seconds of torch CPU kernels for it (slow, still part of -m gpu)."""
    from test_gpu_fullvocab import CONFIGS

    from pipelinerl_amd import _lib
    from pipelinerl_amd.finetune.rl import RLConfig, make_loss_config

    monkeypatch.delenv("PRL_FUSED_VARIANT", raising=False)
    V = 152064
    cfg = CONFIGS[cfg_name]
    logits, batch = _big_case(T, V, cfg, cuda_device)
    want = orlt.rl_step_closed_form(logits, batch, cfg, 2, 10, True)
    want_grad = want["grad_logits"].numpy()
    c_cfg, _, _ = make_loss_config(RLConfig(**cfg), 2, 10)
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(cuda_device) for k, v in batch.items()}
    lt = torch.from_numpy(logits).to(cuda_device)
    zero_rows = np.flatnonzero(np.abs(want["g_nlp"][0]) + np.abs(want["g_ent"][0]) == 0)
    for inplace in (False, True):
        nlp, ent, lse = (torch.full((1, T), 7.0, device=cuda_device) for _ in range(3))
        src = lt.clone()
        grad = src if inplace else torch.full_like(src, 3.0)
        _lib.check(libprl.prl_fused_logits_loss(
            ctypes.byref(c_cfg), 1, T, V, src.data_ptr(), 0, V, cfg["temperature"], d["input_ids"].data_ptr(),
            d["labels"].data_ptr(), d["old_logprobs"].data_ptr(), d["ref_logprobs"].data_ptr(), d["advantages"].data_ptr(),
            d["rewards"].data_ptr(), d["group_tokens"].data_ptr(), d["overflow"].data_ptr(), nlp.data_ptr(), ent.data_ptr(),
            lse.data_ptr(), grad.data_ptr(), _lib.current_stream_ptr(cuda_device)))
        torch.cuda.synchronize()
        assert "keep" in libprl.prl_last_fused_kernel().decode(), "the row-resident kernel is the one bench.py times"
        g_nlp, g_ent = nlp.cpu().numpy(), ent.cpu().numpy()
        assert g_nlp[0, 0] == 0 and g_ent[0, 0] == 0
        np.testing.assert_allclose(g_nlp[:, 1:], want["new_logprobs"], rtol=FP_TOL, atol=2e-5)
        np.testing.assert_allclose(g_ent[:, 1:], want["entropy"], rtol=FP_TOL, atol=2e-5)
        got = grad.cpu().numpy()
        assert rel_err(got, want_grad) <= FP_TOL, inplace
        assert np.count_nonzero(got[0, zero_rows]) == 0 and np.count_nonzero(got[0, -1]) == 0
        if cfg_name == "grpo_clip":
            assert set(range(99, 399)) <= set(zero_rows.tolist())
        del got, src, grad


def _nan_case(device):
    from test_gpu_fullvocab import CONFIGS, _case

    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    V, T = 152064, 40
    logits, batch, want, _ = _case(V, T, "kl_ent_temp", "f32")
    assert batch["labels"][0, 4] == -100  # row 3 predicts token 4: a prompt row
    pb = PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(device)
    lt = torch.from_numpy(logits).to(device)
    lt[0, 3, 1000] = float("nan")
    return lt, pb, want, CONFIGS["kl_ent_temp"]


def test_rl_step_raises_on_a_non_finite_prompt_row(libprl, cuda_device):
    """Reference rl/__init__.py:213: `assert torch.isfinite(new_logprobs).all()` looks at EVERY position, labelled or
    not.  The default drop-in `rl_step` keeps that; `skip_unlabelled_rows=True` is the documented opt-out."""
    from pipelinerl_amd.finetune.rl import RLConfig, rl_step

    lt, pb, want, cfg = _nan_case(cuda_device)
    for fused in (True, False):
        src = lt.clone().requires_grad_(True)
        model = lambda **kw: types.SimpleNamespace(logits=src)  # noqa: E731
        with pytest.raises(AssertionError, match="new_logprobs is not finite"):
            rl_step(model, pb, 2, 10, RLConfig(**cfg, fused_logits_grad=fused))
    src = lt.clone().requires_grad_(True)
    model = lambda **kw: types.SimpleNamespace(logits=src)  # noqa: E731
    loss, stats = rl_step(model, pb, 2, 10, RLConfig(**cfg, skip_unlabelled_rows=True))
    loss.backward()
    assert abs(loss.item() - float(want["loss"])) <= FP_TOL * abs(float(want["loss"]))
    assert torch.isfinite(src.grad).all() and not src.grad[0, 3].any()


def test_fused_head_raises_on_non_finite_hidden_states_of_a_prompt_row(libprl, cuda_device):
    """The fused head hands the kernels only the rows that predict a labelled token; the reference's assertion over
    every position is kept by checking the hidden states the skipped rows would have been computed from."""
    from test_gpu_lmhead_fused import CFG, _problem

    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.fused_head import FusedLmHead, fused_head_loss

    hidden, W, batch, _ = _problem(300, 64, 1088, cuda_device, seed=5)
    assert batch["labels"][0, 4] == -100
    pb = PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(cuda_device)
    head = FusedLmHead(W)
    loss, _ = fused_head_loss(hidden, W, head, pb, RLConfig(**CFG), 2, 10)
    assert torch.isfinite(loss)
    bad = hidden.clone()
    bad[0, 3, 7] = float("nan")
    with pytest.raises(AssertionError, match="new_logprobs is not finite"):
        fused_head_loss(bad, W, head, pb, RLConfig(**CFG), 2, 10)
