"""Parity of the HIP path (through the C ABI) against the golden vectors produced by the
reference and against the CPU oracle.  Needs a real MI355X: `pytest -m gpu`."""

import copy
import types

import numpy as np
import pytest
import torch

from oracle import preprocess as opre
from oracle import rl_loss as orl

from helpers import (
    PREPROCESS_CASES,
    RL_STEP_CASES,
    assert_batch_equal,
    load_preprocess_case,
    load_rl_case,
    rel_err,
)

pytestmark = pytest.mark.gpu

FP_TOL = 1e-4  # north_star: fp loss / grad within 1e-4 relative of the reference CPU path


class FakeModel(torch.nn.Module):
    def __init__(self, logits):
        super().__init__()
        self.logits = torch.nn.Parameter(logits)

    def forward(self, **kwargs):
        return types.SimpleNamespace(logits=self.logits)


class FakeValueModel(FakeModel):
    """What rl_step sees of the reference's AutoModelForCausalLMWithValueHead (finetune/value_model.py:54-116)."""

    def __init__(self, logits, value):
        super().__init__(logits)
        self.value_head = torch.nn.Identity()
        self.value = torch.nn.Parameter(value)

    def forward(self, **kwargs):
        return types.SimpleNamespace(logits=self.logits, value=self.value)


class Tok:
    def __init__(self, eos_token_id=2, padding_side="right"):
        self.eos_token_id = eos_token_id
        self.padding_side = padding_side


def _batch_from_np(b: dict, device):
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    return PipelineBatchEncoding(**{k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v) for k, v in b.items()}).to_device(device)


def _batch_to_np(batch) -> dict:
    out = {}
    for k, v in batch.model_dump().items():
        if isinstance(v, torch.Tensor):
            out[k] = v.detach().cpu().numpy()
        elif v is not None:
            out[k] = v
    return out


@pytest.mark.parametrize("mode", ["two_pass", "fused", "fused_inplace", "two_pass_inplace"])
@pytest.mark.parametrize("name", RL_STEP_CASES)
def test_rl_step_matches_reference(libprl, cuda_device, name, mode):
    from pipelinerl_amd.finetune.rl import RLConfig, rl_step

    case = load_rl_case(name)
    cur, mx = case["steps"]
    cfg = RLConfig(**case["config"], fused_logits_grad=mode.startswith("fused"), inplace_logits_grad=mode.endswith("inplace"))
    batch = _batch_from_np(case["batch"], cuda_device)
    logits = torch.from_numpy(case["logits"]).to(cuda_device)
    model = FakeValueModel(logits, torch.from_numpy(case["value"]).to(cuda_device)) if "value" in case else FakeModel(logits)
    advantages_before = batch.advantages.clone()
    loss, stats = rl_step(model, batch, cur, mx, cfg)
    assert loss.requires_grad and loss.dim() == 0
    assert torch.equal(batch.advantages, advantages_before)  # a value head replaces the column in a COPY of the batch
    if mode.endswith("inplace"):
        # the gradient is written over the logits storage; autograd still routes it to .grad
        pass
    loss.backward()
    want = case["stats"]
    assert list(stats.keys()) == list(want.keys())
    assert abs(loss.item() - case["loss"]) <= FP_TOL * max(abs(case["loss"]), 1e-6) + 1e-7
    for k, w in want.items():
        g = float(stats[k])
        assert abs(g - w) <= FP_TOL * max(abs(w), 1.0), f"{k}: {g} vs {w}"
    # integer-valued stats are exact
    for k in ("num_output_tokens_sum", "input_size"):
        if k in want:
            assert int(stats[k]) == int(want[k])
    grad = model.logits.grad.cpu().numpy()
    scale = np.abs(case["grad_logits"]).max()
    if scale == 0:
        assert np.abs(grad).max() == 0
    else:
        assert rel_err(grad, case["grad_logits"]) <= FP_TOL
    if "value" in case:  # d loss / d outputs.value vs the reference's autograd (rl/__init__.py:367-381)
        assert len(stats) == (1 if "sentinel" in name else 37)
        np.testing.assert_allclose(model.value.grad.cpu().numpy(), case["grad_value"], rtol=FP_TOL, atol=1e-9)


def test_rl_step_sentinel_scaled_loss_has_zero_grad(libprl, cuda_device):
    """finetune_loop.py:784-786: sentinel batches are multiplied by 0 before backward."""
    from pipelinerl_amd.finetune.rl import RLConfig, rl_step
    from pipelinerl_amd.finetune.utils import create_sentinel_batch

    batch = create_sentinel_batch(cuda_device, tokenizer=Tok(5), model_version=1)
    model = FakeModel(torch.randn(1, 8, 64, device=cuda_device))
    loss, stats = rl_step(model, batch, 0, 10, RLConfig(batch_size=4))
    assert stats == {"input_size": 8.0}
    (loss * 0.0).backward()
    assert torch.count_nonzero(model.logits.grad).item() == 0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("vocab,temperature", [(152064, 1.0), (151936, 0.7), (1003, 1.0), (7, 1.3)])
def test_logprob_entropy_vs_oracle(libprl, cuda_device, vocab, temperature, dtype):
    """K1 at real vocabulary sizes (and ragged ones that defeat 16-byte alignment)."""
    from pipelinerl_amd.finetune.rl import logprob_entropy

    rng = np.random.default_rng(vocab)
    B, L = 2, 9
    logits = (rng.standard_normal((B, L, vocab)) * 2).astype(np.float32)
    logits[0, 3, :] += 30.0 * (rng.random(vocab) < 0.001)  # a spiky row: exercises the running-max rescale
    ids = rng.integers(0, vocab, size=(B, L), dtype=np.int64)
    t = torch.from_numpy(logits).to(cuda_device)
    if dtype == "bf16":
        t = t.to(torch.bfloat16)
        logits = t.float().cpu().numpy()
    nlp, ent, lse2, _ = logprob_entropy(t, torch.from_numpy(ids).to(cuda_device), temperature)
    w_nlp, w_ent, _, _ = orl.logprob_entropy(logits.astype(np.float64).astype(np.float32), ids, temperature)
    # fp64 witness for the tolerance
    z = logits[:, :-1].astype(np.float64) / temperature
    lse = np.log(np.exp(z - z.max(-1, keepdims=True)).sum(-1)) + z.max(-1)
    w64 = np.take_along_axis(z, ids[:, 1:, None], -1)[..., 0] - lse
    p = np.exp(z - lse[..., None])
    h64 = -(p * (z - lse[..., None])).sum(-1)
    g_nlp = nlp.cpu().numpy()[:, 1:]
    g_ent = ent.cpu().numpy()[:, 1:]
    assert np.all(nlp.cpu().numpy()[:, 0] == 0) and np.all(ent.cpu().numpy()[:, 0] == 0)
    np.testing.assert_allclose(g_nlp, w64, rtol=FP_TOL, atol=1e-5)
    np.testing.assert_allclose(g_ent, h64, rtol=FP_TOL, atol=1e-5)
    np.testing.assert_allclose(g_nlp, w_nlp, rtol=FP_TOL, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_logits_backward_vs_autograd(libprl, cuda_device, dtype):
    """K1 backward with entropy gradient against torch autograd on the same device."""
    from pipelinerl_amd import _lib
    from pipelinerl_amd.finetune.rl import logprob_entropy

    torch.manual_seed(0)
    B, L, V = 2, 6, 1031
    temp = 0.8
    logits = (torch.randn(B, L, V, device=cuda_device) * 2).to(dtype)
    ids = torch.randint(0, V, (B, L), device=cuda_device)
    g = torch.randn(B, L, device=cuda_device)
    gh = torch.randn(B, L, device=cuda_device) * 0.1
    g[:, 0] = 0
    gh[:, 0] = 0
    up = torch.tensor(0.5, device=cuda_device)
    nlp, ent, lse2, lg = logprob_entropy(logits, ids, temp)
    grad = torch.empty_like(lg)
    _lib.check(_lib.load().prl_logprob_entropy_bwd(
        B, L, V, lg.data_ptr(), 0 if dtype == torch.float32 else 1, V, ids.data_ptr(), temp, lse2.data_ptr(), ent.data_ptr(),
        g.data_ptr(), gh.data_ptr(), up.data_ptr(), grad.data_ptr(), _lib.current_stream_ptr(cuda_device)))
    x = logits.float().detach().requires_grad_(True)
    z = x[:, :-1] / temp
    lp = torch.log_softmax(z, -1)
    r_nlp = lp.gather(2, ids[:, 1:, None])[..., 0]
    r_ent = -(lp.exp() * lp).sum(-1)
    ((r_nlp * g[:, 1:]).sum() * 0.5 + (r_ent * gh[:, 1:]).sum() * 0.5).backward()
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert rel_err(grad.float().cpu().numpy(), x.grad.cpu().numpy()) <= tol


@pytest.mark.parametrize("name", PREPROCESS_CASES)
def test_populate_rl_data_matches_reference(libprl, cuda_device, name):
    from pipelinerl_amd.finetune.data import preprocess_fn
    from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data

    case = load_preprocess_case(name)
    tok = Tok(case["eos_token_id"])
    dataset = []
    for e in copy.deepcopy(case["raw"]):
        if not e.get("ref_logprobs"):
            e["ref_logprobs"] = e["logprobs"]
        entry = dict(e)
        entry.update(preprocess_fn(e, tok, seq_length=10**6, is_rl=True))
        entry["model_version"] = e["metadata"]["model_version"]
        entry["rollout_index"] = e["metadata"]["rollout_index"]
        entry["step_index"] = e["metadata"]["step_index"]
        dataset.append(entry)
    out = populate_rl_data(dataset, case["eos_token_id"], RLConfig(divide_advantage_by_std=case["divide_advantage_by_std"]))
    for key, col in (("advantage", "advantages"), ("group_tokens", "group_tokens"), ("overflow", "overflow"), ("num_labels", "num_labels")):
        got = np.array([e[col][0] for e in out], dtype=np.float64)
        np.testing.assert_allclose(got, case["scalars"][key], rtol=1e-12, atol=1e-15, err_msg=key)
        for e in out:
            assert len(e[col]) == len(e["input_ids"]) and all(x == e[col][0] for x in e[col])
    # the wrapper also has to agree with the oracle's preprocess_fn restatement on the list fields
    want = opre.preprocess_chunk(case["raw"], case["eos_token_id"], case["divide_advantage_by_std"])
    for g, w in zip(out, want):
        for col in ("old_logprobs", "ref_logprobs", "rewards", "attention_mask"):
            assert g[col] == w[col], col


@pytest.mark.parametrize("name", PREPROCESS_CASES)
def test_collate_matches_reference(libprl, cuda_device, name):
    """List-of-dicts collate_packed / collate: integer fields bit-exact, float copies exact."""
    from pipelinerl_amd.finetune.data import collate, collate_packed

    case = load_preprocess_case(name)
    data = opre.preprocess_chunk(case["raw"], case["eos_token_id"], case["divide_advantage_by_std"])
    for plan, want in case["packed"].items():
        idxs = [int(i) for i in want["__idx"]]
        got = collate_packed([data[i] for i in idxs], Tok(case["eos_token_id"]), int(want["__seq_parallel"]))
        assert got.input_ids.is_cuda
        assert_batch_equal(_batch_to_np(got), want)
    for side, want in case["padded"].items():
        idxs = [int(i) for i in want["__idx"]]
        exs = [{k: v for k, v in data[i].items() if k != "finish_reason"} for i in idxs]
        got = collate(exs, Tok(case["eos_token_id"], side))
        assert_batch_equal(_batch_to_np(got), want)


@pytest.mark.parametrize("name", PREPROCESS_CASES)
def test_ragged_device_pipeline_matches_reference(libprl, cuda_device, name):
    """The MI355X path proper: ragged SoA upload -> K5 -> one K6 launch for several micro-batches."""
    from pipelinerl_amd.finetune.data import pack_prepared, pad_prepared
    from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data_ragged
    from pipelinerl_amd.ragged import RaggedRollouts

    case = load_preprocess_case(name)
    rag = RaggedRollouts.from_entries(case["raw"]).to(cuda_device)
    prep = populate_rl_data_ragged(rag, case["eos_token_id"], RLConfig(divide_advantage_by_std=case["divide_advantage_by_std"]))
    np.testing.assert_allclose(prep.advantage64.cpu().numpy(), case["scalars"]["advantage"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(prep.group_tokens64.cpu().numpy(), case["scalars"]["group_tokens"], rtol=1e-12)
    assert np.array_equal(prep.overflow.cpu().numpy().astype(np.float64), case["scalars"]["overflow"])
    assert np.array_equal(prep.num_labels.cpu().numpy().astype(np.float64), case["scalars"]["num_labels"])
    plans = list(case["packed"].items())
    mbs = [[int(i) for i in want["__idx"]] for _, want in plans]
    pads = []
    for (_, want), idxs in zip(plans, mbs):
        sp = int(want["__seq_parallel"])
        tot = sum(len(case["raw"][i]["input_ids"]) for i in idxs)
        pads.append((sp - tot % sp) % sp)
    batches = pack_prepared(prep, mbs, eos_token_id=case["eos_token_id"], sentinel_pad=pads)
    for (plan, want), got in zip(plans, batches):
        assert_batch_equal(_batch_to_np(got), want)
    for side, want in case["padded"].items():
        got = pad_prepared(prep, [int(i) for i in want["__idx"]], padding_side=side)
        assert_batch_equal(_batch_to_np(got), want)


def test_pack_edge_cases(libprl, cuda_device):
    """Ragged edge cases: single-token sequences, a sequence with no completion, odd total length
    (defeats the 4-token vector path), zero-logprob prompts, many tiny sequences."""
    from pipelinerl_amd.finetune.data import pack_prepared
    from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data_ragged
    from pipelinerl_amd.ragged import RaggedRollouts

    rng = np.random.default_rng(3)
    raw = []
    for i, (p, c) in enumerate([(1, 0), (1, 1), (2, 5), (7, 1), (3, 3), (1, 0), (5, 9), (4, 2), (1, 2)] + [(1, 1)] * 40):
        ids = rng.integers(3, 50, size=p + c).tolist()
        raw.append({
            "input_ids": ids, "labels": [-100] * p + ids[p:], "logprobs": (-rng.random(c)).tolist(), "ref_logprobs": [],
            "reward": float(i % 3), "group_id": f"g{i // 3}", "finished": bool(i % 2),
            "metadata": {"model_version": i, "rollout_index": i % 3, "step_index": 0},
        })
    rag = RaggedRollouts.from_entries(raw).to(cuda_device)
    prep = populate_rl_data_ragged(rag, 2, RLConfig(divide_advantage_by_std=True))
    data = opre.preprocess_chunk(raw, 2, True)
    mbs = [[0, 1, 2], [3], [4, 5, 6, 7, 8], list(range(9, len(raw)))]
    got = pack_prepared(prep, mbs, eos_token_id=2)
    for idxs, g in zip(mbs, got):
        want = opre.collate_packed([data[i] for i in idxs], 2, 1)
        assert_batch_equal(_batch_to_np(g), want)


def test_loss_kernel_unaligned_and_tail(libprl, cuda_device):
    """K2+K3 on views that defeat the 16-byte vector path and on lengths with a tail (n % 4 != 0):
    both paths must agree with the oracle."""
    from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, make_loss_config

    case = load_rl_case("c1_ppo_kl_temp")
    cur, mx = case["steps"]
    cfg = RLConfig(**case["config"])
    b = {k: v for k, v in case["batch"].items()}
    T = b["input_ids"].shape[1]
    for cut in (T, T - 1, T - 2, T - 3):
        nb = {k: (v[:, :cut] if isinstance(v, np.ndarray) and v.ndim == 2 else v) for k, v in b.items()}
        nlp, ent, _, _ = orl.logprob_entropy(case["logits"][:, :cut], nb["input_ids"], cfg.temperature)
        want = orl.token_loss(nb, nlp, ent, case["config"], cur, mx, True)
        for misalign in (False, True):
            def dev(a, dtype):
                t = torch.from_numpy(np.ascontiguousarray(a)).to(cuda_device).to(dtype)
                if misalign:  # place the data 4 bytes (fp32) / 8 bytes (i64) off 16-byte alignment
                    buf = torch.empty(t.numel() + 1, dtype=dtype, device=cuda_device)
                    buf[1:] = t.reshape(-1)
                    return buf[1:].view_as(t)
                return t
            batch = _batch_from_np(nb, cuda_device)
            for k in ("labels", "position_ids"):
                setattr(batch, k, dev(nb[k], torch.long))
            for k in ("old_logprobs", "ref_logprobs", "advantages", "rewards", "group_tokens", "num_labels", "overflow"):
                setattr(batch, k, dev(nb[k], torch.float32))
            a_nlp = np.zeros((1, cut), dtype=np.float32)
            a_ent = np.zeros((1, cut), dtype=np.float32)
            a_nlp[:, 1:] = nlp
            a_ent[:, 1:] = ent
            c_cfg, _, _ = make_loss_config(cfg, cur, mx)
            loss, stats, g_nlp, _ = grpo_loss_from_logprobs(c_cfg, batch, dev(a_nlp, torch.float32), dev(a_ent, torch.float32))
            assert abs(loss.item() - float(want["loss"])) <= FP_TOL * max(abs(float(want["loss"])), 1e-6)
            np.testing.assert_allclose(g_nlp.cpu().numpy()[:, 1:], want["g_nlp"], rtol=FP_TOL, atol=1e-9)
            assert g_nlp[0, 0].item() == 0
            s = stats.cpu().numpy()
            assert int(s[1]) == want["stats"]["num_output_tokens_sum"]
            assert int(s[2]) == want["num_sequences"]


def test_loss_nonfinite_raises(libprl, cuda_device):
    """Reference asserts (rl/__init__.py:213) surface as AssertionError from the device counters."""
    from pipelinerl_amd.finetune.rl import RLConfig, rl_step

    case = load_rl_case("c0_ppo")
    batch = _batch_from_np(case["batch"], cuda_device)
    logits = torch.from_numpy(case["logits"]).to(cuda_device)
    logits[0, 5, :] = float("nan")
    with pytest.raises(AssertionError):
        rl_step(FakeModel(logits), batch, 0, 10, RLConfig(**case["config"]))


def test_cpu_tensors_are_refused(libprl):
    """No silent fallback: the product path fails loudly without device tensors."""
    from pipelinerl_amd.finetune.rl import RLConfig, rl_step
    from pipelinerl_amd.finetune.utils import create_sentinel_batch

    batch = create_sentinel_batch(None)
    with pytest.raises(RuntimeError):
        rl_step(FakeModel(torch.randn(1, 8, 16)), batch, 0, 1, RLConfig(batch_size=1))


def test_full_size_properties(libprl, cuda_device):
    """BASELINE sizes (7B GRPO micro-batch: 8192 tokens, V = 152 064; a 64-sequence slice of the
    4096 x 8192 step) through size-independent properties:
      * pack: sum(input_ids) and sum(old_logprobs) are preserved, position/segment ids are the
        closed-form ones, seq_boundaries tile the batch;
      * loss: additivity - the loss of the whole step in ONE launch equals the sum of the
        per-micro-batch losses; gradients are identical;
      * K1 backward: every non-masked row of d logits sums to ~0 and masked rows are exactly 0."""
    from pipelinerl_amd.finetune.data import pack_prepared
    from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, logprob_entropy, make_loss_config, populate_rl_data_ragged
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.synthetic import make_ragged

    seq_len, V = 8192, 152064
    rag_h, _ = make_ragged(8, attempts=8, seq_length=seq_len, vocab=V, seed=1236, dense=True)
    rag = rag_h.to(cuda_device)
    cfg = RLConfig(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0,
                   clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False, batch_size=4096)
    prep = populate_rl_data_ragged(rag, 2, cfg)
    S = rag.n_seqs
    mbs = [[i] for i in range(S)]
    batches = pack_prepared(prep, mbs, eos_token_id=2)
    tok_sum = int(rag_h.tokens.to(torch.int64).sum())
    assert sum(int(b.input_ids.sum()) for b in batches) == tok_sum
    lp_sum = float(rag_h.logprobs.double().sum())
    got_lp = sum(float(b.old_logprobs.double().sum()) for b in batches)
    assert abs(got_lp - lp_sum) <= 1e-9 * abs(lp_sum)
    for b in batches:
        T = b.input_ids.shape[1]
        assert T == seq_len
        assert torch.equal(b.position_ids[0], torch.arange(T, device=cuda_device))
        assert int(b.segment_ids.abs().sum()) == 0
        assert b.seq_boundaries.tolist() == [0, T]
    # one launch over the whole slice vs per-micro-batch launches
    c_cfg, _, _ = make_loss_config(cfg, 0, 10)
    torch.manual_seed(1)
    whole = {}
    for k in ("labels", "position_ids", "old_logprobs", "ref_logprobs", "advantages", "rewards", "group_tokens", "num_labels", "overflow", "input_ids", "attention_mask"):
        whole[k] = torch.cat([getattr(b, k) for b in batches], dim=1)
    big = PipelineBatchEncoding(**whole, model_version=0, is_packed=True)
    nlp = big.old_logprobs + 0.02 * torch.randn_like(big.old_logprobs)
    ent = 3 * torch.rand_like(nlp)
    loss_all, stats_all, g_all, _ = grpo_loss_from_logprobs(c_cfg, big, nlp, ent)
    parts, grads = [], []
    for j, b in enumerate(batches):
        sl = slice(j * seq_len, (j + 1) * seq_len)
        l, s, g, _ = grpo_loss_from_logprobs(c_cfg, b, nlp[:, sl].contiguous(), ent[:, sl].contiguous())
        parts.append(s[0].item())
        grads.append(g)
    assert abs(sum(parts) - stats_all[0].item()) <= 1e-9 * max(abs(stats_all[0].item()), 1e-12)
    g_cat = torch.cat(grads, dim=1)
    # first token of each later micro-batch is a boundary in `big` but column 0 in its own batch
    assert torch.equal(g_cat, g_all)
    assert int(stats_all[2].item()) == S
    # K1 at full micro-batch size
    logits = torch.randn(1, 1024, V, device=cuda_device) * 2
    b0 = batches[0]
    ids = b0.input_ids[:, :1024].contiguous()
    nlp1, ent1, lse2, lg = logprob_entropy(logits, ids, 1.0)
    assert torch.isfinite(nlp1).all() and torch.isfinite(ent1).all()
    assert (nlp1[:, 1:] < 0).all() and (ent1[:, 1:] > 0).all() and (ent1[:, 1:] < np.log(V) + 1e-3).all()
    from pipelinerl_amd import _lib
    g = torch.randn(1, 1024, device=cuda_device)
    g[:, ::3] = 0
    grad = torch.empty_like(lg)
    _lib.check(_lib.load().prl_logprob_entropy_bwd(1, 1024, V, lg.data_ptr(), 0, V, ids.data_ptr(), 1.0, lse2.data_ptr(),
                                                   ent1.data_ptr(), g.data_ptr(), None, None, grad.data_ptr(),
                                                   _lib.current_stream_ptr(cuda_device)))
    rows = grad[0].double().sum(-1)
    assert rows.abs().max().item() < 1e-3
    zero_rows = (g[0, 1:] == 0).nonzero().flatten()
    assert torch.count_nonzero(grad[0, zero_rows]).item() == 0
    assert torch.count_nonzero(grad[0, -1]).item() == 0


def test_wsync_single_rank_group(libprl, cuda_device):
    """RCCL is resolved at run time (dlopen) and a 1-rank communicator can be created; bucket
    planning round-trips parameters through the flat layout.  (Multi-rank transfers are covered
    by the world_size-2 gloo tests of the bucket logic and by bench.py --gpus N.)"""
    from pipelinerl_amd.weight_sync import BucketedReceiver, BucketedSender, WeightSyncGroup

    grp = WeightSyncGroup._init(WeightSyncGroup._new_uid(), 0, 1, cuda_device)
    assert grp.comm_size() == (1, 0)  # what RCCL itself reports for the communicator
    params = [("a.weight", torch.randn(33, 7, device=cuda_device).bfloat16()), ("a.bias", torch.randn(7, device=cuda_device)),
              ("b.weight", torch.randn(1025, device=cuda_device).half())]
    sender = BucketedSender(grp, bucket_bytes=4096)
    specs = sender.send(params)
    torch.cuda.synchronize()
    # with one rank the staging buffers ARE the received data: unflatten and compare
    got = {}
    recv = BucketedReceiver(grp, bucket_bytes=4096)
    recv._staging = sender._staging
    n = recv.receive(specs, lambda views: got.update({k: v.clone() for k, v in views}))
    assert n == 3
    # bucket k uses staging[k % 2]; with 3 one-parameter buckets bucket 0 was overwritten by bucket 2
    for name, t in params[1:]:
        assert torch.equal(got[name], t), name
    grp.close()


@pytest.mark.parametrize("variant", [4, 6, 21])
def test_fused_kernel_variants_agree(libprl, cuda_device, variant, monkeypatch):
    """The launch-geometry variants of the fused logits kernel (block size, reversed second pass,
    non-temporal stores, residency cap) are the same arithmetic per element: bitwise equal."""
    import ctypes

    from pipelinerl_amd import _lib
    from pipelinerl_amd.finetune.rl import RLConfig, make_loss_config

    torch.manual_seed(variant)
    T, V = 48, 152064
    cfg = RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.05, final_kl_coef=0.05, batch_size=8,
                   entropy_bonus=0.01, final_entropy_bonus=0.01)
    c_cfg, _, _ = make_loss_config(cfg, 0, 10)
    logits = torch.randn(1, T, V, device=cuda_device) * 2
    ids = torch.randint(0, V, (1, T), device=cuda_device)
    labels = ids.clone()
    labels[:, :5] = -100
    f = lambda: torch.randn(1, T, device=cuda_device)  # noqa: E731
    old, ref, adv, rew = -f().abs() - 12, -f().abs() - 12, f(), f()
    gt, ovf = torch.full((1, T), 30.0, device=cuda_device), torch.zeros(1, T, device=cuda_device)

    def run():
        nlp, ent, lse = (torch.empty(1, T, device=cuda_device) for _ in range(3))
        grad = torch.empty_like(logits)
        _lib.check(_lib.load().prl_fused_logits_loss(
            ctypes.byref(c_cfg), 1, T, V, logits.data_ptr(), 0, V, 1.0, ids.data_ptr(), labels.data_ptr(), old.data_ptr(),
            ref.data_ptr(), adv.data_ptr(), rew.data_ptr(), gt.data_ptr(), ovf.data_ptr(), nlp.data_ptr(), ent.data_ptr(),
            lse.data_ptr(), grad.data_ptr(), _lib.current_stream_ptr(cuda_device)))
        torch.cuda.synchronize()
        return nlp, ent, grad

    monkeypatch.setenv("PRL_FUSED_VARIANT", "0")
    n0, e0, g0 = run()
    monkeypatch.setenv("PRL_FUSED_VARIANT", str(variant))
    n1, e1, g1 = run()
    assert torch.count_nonzero(g0).item() > 0
    # block-size changes alter the reduction tree of the online softmax: fp32-rounding-level differences
    assert torch.allclose(n0, n1, rtol=1e-6, atol=1e-6) and torch.allclose(e0, e1, rtol=1e-5, atol=1e-6)
    assert rel_err(g1.cpu().numpy(), g0.cpu().numpy()) <= 1e-5


@pytest.mark.parametrize("seed", range(12))
def test_random_ragged_pipeline_vs_oracle(libprl, cuda_device, seed):
    """Randomised ragged shapes (sequence lengths 1..70, random grouping / steps / finish flags,
    random micro-batch plans with sequence-parallel fillers): K5 + K6 against the oracle,
    bit-exact on every integer field and fp32 copy."""
    from pipelinerl_amd.finetune.data import pack_prepared
    from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data_ragged
    from pipelinerl_amd.ragged import RaggedRollouts

    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(1, 60))
    raw = []
    for i in range(n):
        p, c = int(rng.integers(1, 20)), int(rng.integers(0, 50))
        ids = rng.integers(2, 40, size=p + c).tolist()  # 2 == EOS may appear anywhere
        e = {
            "input_ids": ids, "labels": [-100] * p + ids[p:], "logprobs": (-rng.random(c)).astype(np.float32).tolist(),
            "ref_logprobs": (-rng.random(c)).astype(np.float32).tolist() if seed % 2 else [],
            "reward": float(rng.integers(0, 3)) if seed % 3 else float(rng.normal()),
            "group_id": f"g{int(rng.integers(0, max(1, n // 4) + 1))}", "finished": bool(rng.integers(0, 2)),
            "metadata": {"model_version": int(rng.integers(0, 9)), "rollout_index": int(rng.integers(0, 4)), "step_index": int(rng.integers(0, 2))},
        }
        fr = rng.integers(0, 4)
        if fr == 1:
            e["finish_reason"] = " Length "
        elif fr == 2:
            e["finish_reason"] = "stop"
        raw.append(e)
    divide = bool(seed % 2)
    rag = RaggedRollouts.from_entries(raw).to(cuda_device)
    prep = populate_rl_data_ragged(rag, 2, RLConfig(divide_advantage_by_std=divide))
    data = opre.preprocess_chunk(raw, 2, divide)
    want_scalars = opre.sequence_scalars(data, 2, divide)
    np.testing.assert_allclose(prep.advantage64.cpu().numpy(), [s[0] for s in want_scalars], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(prep.group_tokens64.cpu().numpy(), [s[1] for s in want_scalars], rtol=1e-12)
    assert prep.overflow.cpu().tolist() == [s[2] for s in want_scalars]
    assert prep.num_labels.cpu().tolist() == [float(s[3]) for s in want_scalars]
    # random partition of a random permutation into micro-batches
    order = rng.permutation(n).tolist()
    mbs, k = [], 0
    while k < n:
        m = int(rng.integers(1, 8))
        mbs.append(order[k:k + m])
        k += m
    sp = int(rng.choice([1, 1, 2, 4, 8]))
    pads = [(sp - sum(len(raw[i]["input_ids"]) for i in mb) % sp) % sp for mb in mbs]
    got = pack_prepared(prep, mbs, eos_token_id=2, sentinel_pad=pads)
    for mb, g in zip(mbs, got):
        # the fp32 scalar columns come from the device fp64 results: compare against the oracle's
        # fp64 -> fp32 cast (identical unless the fp64 values differ in the last bit)
        want = opre.collate_packed([data[i] for i in mb], 2, sp)
        assert_batch_equal(_batch_to_np(g), want, float_tol=1e-6)
        for key in ("input_ids", "labels", "position_ids", "segment_ids", "attention_mask", "seq_boundaries"):
            assert np.array_equal(np.asarray(_batch_to_np(g)[key]), want[key]), key


def test_annotate_ref_logprobs(libprl, cuda_device):
    """Reference-model logprobs computed on the learner GPU equal torch log_softmax on labelled
    tokens and are zero elsewhere (the layout prepare_rl_fields gives ref_logprobs)."""
    from pipelinerl_amd.finetune.rl import annotate_ref_logprobs

    case = load_rl_case("c1_ppo_kl_temp")
    batch = _batch_from_np(case["batch"], cuda_device)
    logits = torch.from_numpy(case["logits"]).to(cuda_device)
    ref_model = lambda **kw: types.SimpleNamespace(logits=logits)  # noqa: E731
    annotate_ref_logprobs(ref_model, batch)
    lp = torch.log_softmax(logits[:, :-1].double(), -1).gather(2, batch.input_ids[:, 1:, None])[..., 0]
    mask = batch.labels[:, 1:] != -100
    got = batch.ref_logprobs[:, 1:]
    assert torch.allclose(got[mask].double(), lp[mask], rtol=1e-5, atol=1e-5)
    assert torch.count_nonzero(got[~mask]).item() == 0 and batch.ref_logprobs[0, 0].item() == 0


def test_rl_step_bf16_logits(libprl, cuda_device):
    """bf16 logits (a model whose lm_head is not forced to fp32): both logits kernels accept them;
    compare with the oracle evaluated on the bf16-rounded values."""
    from pipelinerl_amd.finetune.rl import RLConfig, rl_step

    case = load_rl_case("c1_ppo_kl_temp")
    cur, mx = case["steps"]
    lg16 = torch.from_numpy(case["logits"]).to(cuda_device).to(torch.bfloat16)
    want = orl.rl_step(lg16.float().cpu().numpy(), case["batch"], case["config"], cur, mx, True)
    for fused in (False, True):
        batch = _batch_from_np(case["batch"], cuda_device)
        model = FakeModel(lg16.clone())
        loss, stats = rl_step(model, batch, cur, mx, RLConfig(**case["config"], fused_logits_grad=fused))
        loss.backward()
        assert abs(loss.item() - float(want["loss"])) <= 1e-4 * max(abs(float(want["loss"])), 1e-6)
        g = model.logits.grad.float().cpu().numpy()
        assert rel_err(g, want["grad_logits"]) <= 1e-2  # the gradient itself is rounded to bf16


@pytest.mark.parametrize("name", PREPROCESS_CASES)
def test_zero_advantage_group_mask(libprl, cuda_device, name):
    """Device-path group filter == the reference's list-based filter_zero_advantage_groups."""
    from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data_ragged
    from pipelinerl_amd.preprocess import filter_zero_advantage_groups, nonzero_advantage_mask
    from pipelinerl_amd.ragged import RaggedRollouts

    case = load_preprocess_case(name)
    rag = RaggedRollouts.from_entries(case["raw"]).to(cuda_device)
    prep = populate_rl_data_ragged(rag, case["eos_token_id"], RLConfig(divide_advantage_by_std=case["divide_advantage_by_std"]))
    keep = nonzero_advantage_mask(prep)
    data = opre.preprocess_chunk(case["raw"], case["eos_token_id"], case["divide_advantage_by_std"])
    for i, e in enumerate(data):
        e["uid"] = i
    kept, dropped = filter_zero_advantage_groups(data)
    assert sorted(e["uid"] for e in kept) == np.flatnonzero(keep).tolist() and dropped == int((~keep).sum())


def test_segment_sums_are_reproducible_and_checked(libprl, cuda_device):
    """The GSPO per-segment sums: equal to an fp64 index_add (reference rl/utils.py:106-208), bitwise
    identical from run to run (fixed reduction order, no atomics), NaN when the segment ids are not
    the non-decreasing ids of a packed batch."""
    from pipelinerl_amd.finetune.rl import segment_sums

    torch.manual_seed(3)
    T, S = 20000, 37
    cuts = torch.sort(torch.randperm(T - 2)[: S - 1] + 1).values
    seg = torch.zeros(T, dtype=torch.int64)
    seg[cuts] = 1
    seg = torch.cumsum(seg, 0)[None].to(cuda_device)
    labels = torch.randint(0, 100, (1, T), device=cuda_device)
    labels[torch.rand(1, T, device=cuda_device) < 0.3] = -100
    a = torch.randn(1, T, device=cuda_device) * 1e3
    b = torch.randn(1, T, device=cuda_device)
    runs = [segment_sums(seg, labels, a, b, S) for _ in range(4)]
    for r in runs[1:]:
        for x, y in zip(runs[0], r):
            assert torch.equal(x, y)
    m = (labels[0, 1:] != -100)
    idx = seg[0, 1:][m]
    for got, src in zip(runs[0], (a[0, 1:][m].double(), b[0, 1:][m].double(), torch.ones(int(m.sum()), dtype=torch.float64, device=cuda_device))):
        want = torch.zeros(S, dtype=torch.float64, device=cuda_device).index_add_(0, idx, src)
        assert torch.allclose(got, want, rtol=1e-12, atol=1e-9)
    bad = seg.clone()
    bad[0, 500], bad[0, 501] = 5, 4
    assert all(torch.isnan(x).all() for x in segment_sums(bad, labels, a, b, S))
