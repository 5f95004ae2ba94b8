"""The value-head (actor-critic) branch of rl_step on the GPU (csrc/prl_value.hip through the C ABI) against the CPU
oracle (`oracle.rl_loss.token_loss(value=...)`, pinned to the reference's own outputs by the c18-c23 goldens, which
tests/test_gpu_parity.py runs through `rl_step` in all four logits modes).  Reference: rl/__init__.py:162, 265-272,
367-381, 441-448; finetune/value_model.py."""

import ctypes
import types

import numpy as np
import pytest
import torch

from oracle import rl_loss as orl

pytestmark = pytest.mark.gpu

FP_TOL = 1e-4


def _packed_case(T, seed, group_normalization):
    rng = np.random.default_rng(seed)
    bounds = [0, T // 7, T // 3, (2 * T) // 3, T]
    pos = np.concatenate([np.arange(b - a) for a, b in zip(bounds[:-1], bounds[1:])])[None].astype(np.int64)
    seg = np.concatenate([np.full(b - a, k) for k, (a, b) in enumerate(zip(bounds[:-1], bounds[1:]))])[None].astype(np.int64)
    ids = rng.integers(3, 1000, size=(1, T), dtype=np.int64)
    labels = ids.copy()
    for a in bounds[:-1]:
        labels[0, a:a + 23] = -100
    labels[0, rng.random(T) < 0.05] = -100
    f32 = lambda a: np.asarray(a, dtype=np.float32)[None]  # noqa: E731
    per_seq = lambda vals: np.concatenate([np.full(b - a, v) for v, (a, b) in zip(vals, zip(bounds[:-1], bounds[1:]))])  # noqa: E731
    n_lab = [(labels[0, a:b] != -100).sum() for a, b in zip(bounds[:-1], bounds[1:])]
    batch = {
        "input_ids": ids, "labels": labels, "position_ids": pos, "attention_mask": np.ones_like(ids), "segment_ids": seg,
        "old_logprobs": f32(rng.normal(-2, 0.5, T)), "ref_logprobs": f32(rng.normal(-2, 0.5, T)), "advantages": f32(rng.normal(0, 1, T)),
        "rewards": f32(per_seq([1.0, 0.0, 0.25, 1.0])), "group_tokens": f32(per_seq([311.0, 311.0, 977.0, 977.0])),
        "num_labels": f32(per_seq(n_lab)), "overflow": f32(per_seq([0.0, 1.0, 0.0, 0.0])),
    }
    value = (0.4 + 0.3 * rng.standard_normal((1, T))).astype(np.float32)
    config = {"policy_loss": "ppo", "epsilon_low": 0.2, "epsilon_high": 0.2, "kl_coef": 0.0, "final_kl_coef": 0.0, "batch_size": 64,
              "group_normalization": group_normalization, "overlong_filtering": True, "value_loss_coef": 0.25, "use_advantages": True,
              "relu_log_p_weights": False, "clamp_log_ratio_ref_new_value": 10, "temperature": 1.0}
    return batch, value, config


def _to_batch(b, device):
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    return PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in b.items()}, is_packed=True, model_version=0).to_device(device)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
@pytest.mark.parametrize("T,group_normalization", [(8192, True), (8192, False), (65536, True), (257, False)])
def test_value_head_kernel_vs_oracle(libprl, cuda_device, T, group_normalization, dtype):
    """One micro-batch (8192 tokens), a whole step's worth (65 536: more blocks than one wave of the grid) and a ragged
    short row: advantages column, value loss, its gradient and the five statistics."""
    from pipelinerl_amd.finetune.rl import VALUE_STAT_KEYS, RLConfig, make_loss_config, value_head_terms

    b, value, config = _packed_case(T, seed=T + group_normalization, group_normalization=group_normalization)
    vt = torch.from_numpy(value).to(cuda_device)
    if dtype == "bf16":  # what autocast hands over; the oracle gets the same rounded numbers
        vt = vt.to(torch.bfloat16)
        value = vt.float().cpu().numpy()
    batch = _to_batch(b, cuda_device)
    cfg, _, _ = make_loss_config(RLConfig(**config), 0, 10)
    vloss, adv, vstats, g_val = value_head_terms(cfg, batch, vt)
    again = value_head_terms(cfg, batch, vt)
    assert torch.equal(vstats, again[2]) and torch.equal(vloss, again[0])  # fixed-order reduction: bitwise reproducible

    nlp = b["old_logprobs"].copy()  # on-policy stand-in: the value terms do not depend on it
    want = orl.token_loss(b, nlp[:, 1:], np.zeros_like(nlp[:, 1:]), config, 0, 10, True, value=value)
    # the advantages column: rewards - V one column to the left, EVERY position, bit for bit (one fp32 subtraction)
    want_adv = np.zeros_like(value)
    want_adv[:, 1:] = b["rewards"][:, 1:] - value[:, :-1]
    assert np.array_equal(adv.cpu().numpy(), want_adv)
    got = dict(zip(VALUE_STAT_KEYS, vstats.cpu().tolist()))
    for k in VALUE_STAT_KEYS:
        assert abs(got[k] - want["stats"][k]) <= FP_TOL * max(1.0, abs(want["stats"][k])), (k, got[k], want["stats"][k])
    assert got["value_max"] == want["stats"]["value_max"] and got["value_min"] == want["stats"]["value_min"]
    assert abs(vloss.item() - want["stats"]["value_loss"]) <= FP_TOL * max(1.0, abs(want["stats"]["value_loss"]))
    # oracle's g_value = d loss / d value = coef * d value_loss / d value
    np.testing.assert_allclose(config["value_loss_coef"] * g_val.cpu().numpy().astype(np.float64), want["g_value"], rtol=1e-5, atol=1e-12)
    assert np.count_nonzero(g_val.cpu().numpy()[0, :-1][b["labels"][0, 1:] == -100]) == 0 and g_val[0, -1].item() == 0.0


def test_value_head_non_finite_and_unlabelled(libprl, cuda_device):
    """nan_to_num semantics of sum_sum (rl/utils.py:26-31): a NaN prediction contributes 0 and gets no gradient, an
    overflowing square saturates at FLT_MAX; a batch without labels reports zeros (the `.any() else 0.0` of :443-444)."""
    from pipelinerl_amd.finetune.rl import VALUE_STAT_KEYS, RLConfig, make_loss_config, value_head_terms

    b, value, config = _packed_case(512, seed=5, group_normalization=False)
    lab = np.flatnonzero(b["labels"][0, 1:] != -100)
    value[0, lab[3]] = np.nan
    value[0, lab[7]] = 3e19  # (V - r)^2 overflows fp32
    batch = _to_batch(b, cuda_device)
    cfg, _, _ = make_loss_config(RLConfig(**config), 0, 10)
    vloss, adv, vstats, g_val = value_head_terms(cfg, batch, torch.from_numpy(value).to(cuda_device))
    with np.errstate(all="ignore"):
        want = orl.token_loss(b, b["old_logprobs"][:, 1:], np.zeros_like(b["old_logprobs"][:, 1:]), config, 0, 10, True, value=value)
    got = dict(zip(VALUE_STAT_KEYS, vstats.cpu().tolist()))
    for k in ("value_loss", "value_mse", "value_mean"):
        assert np.isfinite(got[k]) and abs(got[k] - want["stats"][k]) <= FP_TOL * max(1.0, abs(want["stats"][k])), k
    g = g_val.cpu().numpy()
    assert g[0, lab[3]] == 0.0 and g[0, lab[7]] == 0.0 and want["g_value"][0, lab[3]] == 0.0 and want["g_value"][0, lab[7]] == 0.0
    assert np.isnan(adv[0, lab[3] + 1].item())  # the advantages tensor itself is not sanitised, as in the reference

    b["labels"][:] = -100
    vloss, adv, vstats, g_val = value_head_terms(cfg, _to_batch(b, cuda_device), torch.from_numpy(np.nan_to_num(value, posinf=0.0)).to(cuda_device))
    assert vstats.cpu().tolist() == [0.0] * 5 and vloss.item() == 0.0 and torch.count_nonzero(g_val).item() == 0


def test_value_head_entry_refuses_bad_arguments(libprl, cuda_device):
    from pipelinerl_amd import _lib
    from pipelinerl_amd.finetune.rl import PrlLossConfig

    t = torch.zeros(1 << 16, dtype=torch.uint8, device=cuda_device)
    P, s, cfg = t.data_ptr(), _lib.current_stream_ptr(cuda_device), PrlLossConfig()
    need = ctypes.c_size_t()
    _lib.check(libprl.prl_value_head_workspace_bytes(1, 8192, ctypes.byref(need)))
    assert 0 < need.value <= 1 << 16
    call = lambda rows, dtype, ws_bytes, stats=P: libprl.prl_value_head_fwd_bwd(  # noqa: E731
        ctypes.byref(cfg), rows, 64, P, P, dtype, P, P, P, P, P, None, None, stats, P, ws_bytes, s)
    assert call(1, _lib.PRL_DTYPE_F32, need.value) == _lib.PRL_OK  # grad_values / value_loss_out are optional
    assert call(0, _lib.PRL_DTYPE_F32, need.value) == _lib.PRL_EINVAL
    assert call(1, 7, need.value) == _lib.PRL_EINVAL and "float32 or bfloat16" in libprl.prl_last_error().decode()
    assert call(1, _lib.PRL_DTYPE_F32, 64) == _lib.PRL_ENOMEM
    assert call(1, _lib.PRL_DTYPE_F32, need.value, None) == _lib.PRL_EINVAL
    torch.cuda.synchronize()


class TinyActorCritic(torch.nn.Module):
    """The reference's AutoModelForCausalLMWithValueHead layout (finetune/value_model.py:54-116): `.pretrained_model` is a
    Hugging Face style causal LM (`.model` body + bias-free `.lm_head`), `.value_head` maps the last hidden states to [B, L]."""

    class Body(torch.nn.Module):
        def __init__(self, vocab, dim):
            super().__init__()
            self.emb = torch.nn.Embedding(vocab, dim)

        def forward(self, input_ids=None, **kw):
            return types.SimpleNamespace(last_hidden_state=torch.tanh(self.emb(input_ids)).to(torch.bfloat16))

    class LM(torch.nn.Module):
        def __init__(self, vocab, dim):
            super().__init__()
            self.model = TinyActorCritic.Body(vocab, dim)
            self.lm_head = torch.nn.Linear(dim, vocab, bias=False)

    class ValueHead(torch.nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.output = torch.nn.Linear(dim, 1)

        def forward(self, hidden):
            return self.output(hidden.float()).squeeze(-1)

    def __init__(self, vocab=1024, dim=128):
        super().__init__()
        self.pretrained_model = TinyActorCritic.LM(vocab, dim)
        self.value_head = TinyActorCritic.ValueHead(dim)

    def forward(self, input_ids=None, **kw):
        h = self.pretrained_model.model(input_ids=input_ids).last_hidden_state
        return types.SimpleNamespace(logits=self.pretrained_model.lm_head(h.float()), value=self.value_head(h))


def test_actor_critic_model_through_rl_step_and_the_fused_head(libprl, cuda_device):
    """A model with a value head end to end, both drop-in entry points: `rl_step` (logits from the model's forward) and
    `rl_step_fused_head` (no logits; the critic reads the same hidden states) agree with each other and with the oracle
    fed the model's own logits and values - loss, 37 statistics, gradients of the critic, the head and the embedding."""
    from pipelinerl_amd.finetune.rl import RLConfig, rl_step
    from pipelinerl_amd.fused_head import rl_step_fused_head

    torch.manual_seed(3)
    T, V = 1024, 1024
    b, _, config = _packed_case(T, seed=9, group_normalization=True)
    b["input_ids"] = b["input_ids"] % V
    b["labels"] = np.where(b["labels"] == -100, -100, b["input_ids"])
    model = TinyActorCritic(V, 128).to(cuda_device)
    torch.nn.init.normal_(model.value_head.output.weight, std=0.3)
    batch = _to_batch(b, cuda_device)
    cfg = RLConfig(**config)
    with torch.no_grad():
        out = model(input_ids=batch.input_ids)
        # on-policy-ish: old = new + noise, so that the PPO ratio sits near 1
        z = out.logits.double()[:, :-1]
        nlp = (z.gather(-1, batch.input_ids[:, 1:, None])[..., 0] - torch.logsumexp(z, -1)).float()
        batch.old_logprobs[:, 1:] = nlp + 0.02 * torch.randn_like(nlp)
        batch.ref_logprobs = batch.old_logprobs.clone()
    b["old_logprobs"] = batch.old_logprobs.cpu().numpy()
    b["ref_logprobs"] = batch.ref_logprobs.cpu().numpy()
    want = orl.rl_step(out.logits.cpu().numpy(), b, config, 0, 10, True, value=out.value.cpu().numpy())

    from pipelinerl_amd.fused_head import install_fused_head

    class Wrapper(torch.nn.Module):  # what DistributedDataParallel / accelerate present: the model as `.module`, calls forwarded
        def __init__(self, module):
            super().__init__()
            self.module = module

        def forward(self, *a, **k):
            return self.module(*a, **k)

    def installed_step(m, *a, **k):  # the loss produced INSIDE the wrapped model's forward (what DDP / FSDP need)
        if getattr(m, "_prl_fused_head", None) is None:
            install_fused_head(m)
        return rl_step_fused_head(Wrapper(m), *a, **k)

    results = []
    for step in (rl_step, rl_step_fused_head, installed_step):
        model.zero_grad(set_to_none=True)
        loss, stats = step(model, batch, 0, 10, cfg)
        loss.backward()
        results.append((loss.item(), stats, {n: p.grad.detach().clone() for n, p in model.named_parameters()}))
        assert len(stats) == 37 and list(stats)[-5:] == ["value_mean", "value_max", "value_min", "value_loss", "value_mse"]
        assert abs(loss.item() - float(want["loss"])) <= FP_TOL * max(1.0, abs(float(want["loss"])))
        for k, w in want["stats"].items():
            assert abs(float(stats[k]) - float(w)) <= 2 * FP_TOL * max(1.0, abs(float(w))), (step.__name__, k, stats[k], w)
    (l0, s0, g0), (l1, s1, g1), (l2, s2, g2) = results
    assert abs(l0 - l1) <= FP_TOL * max(1.0, abs(l0)) and l1 == l2 and s1 == s2  # bare and installed: the same launches
    for n in g0:
        scale = g0[n].abs().max().item()
        assert scale > 0 and (g0[n] - g1[n]).abs().max().item() <= 2e-3 * scale, n  # bf16x2 head vs fp32 autograd of a bf16 hidden state
        assert (g1[n] - g2[n]).abs().max().item() <= 1e-6 * scale, n  # (the embedding's backward accumulates with atomics)
    assert model(input_ids=batch.input_ids).logits.shape == (1, T, V)  # every other call is still the model's own forward
    # the critic's weight gradient = hidden^T (coef * d value_loss / d V), from the oracle's closed form
    h = model.pretrained_model.model(input_ids=batch.input_ids).last_hidden_state.detach().float()[0].double().cpu().numpy()
    want_gw = want["g_value"][0].astype(np.float64) @ h
    np.testing.assert_allclose(g0["value_head.output.weight"].cpu().numpy()[0], want_gw, rtol=1e-3, atol=1e-6 * np.abs(want_gw).max())


def test_native_step_with_a_value_head_equals_per_micro_batch_rl_step(libprl, cuda_device):
    """`NativeLearnerStep` with an actor-critic model: one K6 launch, per micro-batch the value kernel (advantages := rewards - V written INTO
    the step batch, value-loss gradient) + the fused logits kernel, ONE statistics launch - against the drop-in loop of `rl_step` (which the
    c18-c23 goldens pin to the reference) over the same micro-batches: parameter gradients of the critic, the head and the embedding,
    the combined loss, the 32 statistics and the five value statistics."""
    import copy

    from pipelinerl_amd.finetune.rl import VALUE_STAT_KEYS, RLConfig, rl_step
    from pipelinerl_amd.finetune_loop import NativeLearnerStep
    from pipelinerl_amd.hotpath import HotPathStep, dense_micro_batches
    from pipelinerl_amd.synthetic import make_ragged

    V = 256
    rag_h, _ = make_ragged(4, attempts=4, seq_length=48, vocab=V, seed=7, prompt_min=3, prompt_max=9, with_ref=True)
    rag = rag_h.to(cuda_device)
    mbs = dense_micro_batches(rag_h, 120)
    assert len(mbs) >= 3
    rl = RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.05, final_kl_coef=0.05, divide_advantage_by_std=False,
                  clamp_log_ratio_ref_new_value=5, value_loss_coef=0.25, group_normalization=True, overlong_filtering=True)
    torch.manual_seed(1)
    model_a = TinyActorCritic(V, 64).to(cuda_device)
    torch.nn.init.normal_(model_a.value_head.output.weight, std=0.3)
    model_b = copy.deepcopy(model_a)

    captured = {}
    opt_a = torch.optim.SGD(model_a.parameters(), lr=0.0)
    orig_zero = opt_a.zero_grad
    opt_a.zero_grad = lambda *a, **k: captured.update(g={n: p.grad.detach().clone() for n, p in model_a.named_parameters()}) or orig_zero(*a, **k)
    native = NativeLearnerStep(model_a, opt_a, rl, eos_token_id=2, samples_per_step=16, max_train_steps=10, skip_unlabelled=False)
    assert native.has_value_head
    res = native.step(rag, mbs)
    stats_a = native.stats_dict(res["stats"])
    assert list(stats_a)[-5:] == list(VALUE_STAT_KEYS) and len(stats_a) == 37

    cfg_b = rl.model_copy()
    cfg_b.batch_size = 16
    hp = HotPathStep(cfg_b, 2, 0, 10)
    batches = hp.preprocess(rag, mbs)
    agg, total = {}, 0.0
    for b in batches:
        loss, st = rl_step(model_b, b, 0, 10, cfg_b)
        loss.backward()
        total += loss.item()
        for k, v in st.items():
            agg.setdefault(k, []).append(v)
    for n, pb in model_b.named_parameters():
        ga = captured["g"][n]
        assert pb.grad is not None and float(pb.grad.abs().max()) > 0, n
        assert torch.allclose(ga, pb.grad, rtol=2e-4, atol=1e-6 * float(pb.grad.abs().max())), n
    assert abs(res["loss"].item() - total) <= FP_TOL * max(1.0, abs(total))
    assert abs(stats_a["loss"] - sum(agg["loss"])) <= FP_TOL * max(1.0, abs(sum(agg["loss"])))
    for k in ("reward", "kl", "ratio_new_old", "advantage", "token_weight", "value_mean", "value_loss", "value_mse"):
        assert abs(stats_a[k] - sum(agg[k])) <= FP_TOL * max(1.0, abs(sum(agg[k]))), k
    for k in ("max_advantage", "value_max"):
        assert abs(stats_a[k] - max(agg[k])) <= 1e-6 * max(1.0, abs(max(agg[k]))), k
    for k in ("min_advantage", "value_min"):
        assert abs(stats_a[k] - min(agg[k])) <= 1e-6 * max(1.0, abs(min(agg[k]))), k
    # the advantages the policy term used are rewards - V, not the column the preprocessor wrote
    assert abs(stats_a["advantage"] - sum(agg["advantage"])) <= FP_TOL and stats_a["value_loss"] > 0
