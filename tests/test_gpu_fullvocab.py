"""Oracle parity of the kernels the benchmark times, at the BASELINE vocabulary sizes.

The row-resident fused kernel (`fused_logits_loss_keep_kernel`) is only dispatched for V >= 102 400
(fp32), so the golden cases (V <= 128) never reach it.  Here every selectable variant - and the
default dispatch with PRL_FUSED_VARIANT unset - runs at V = 152 064 (Qwen2.5-7B) and V = 151 936
(Qwen2.5-0.5B), fp32 and bf16, in place and out of place, with masked rows, PPO-clipped rows (zero
gradient), kl > 0, an entropy bonus and temperature != 1, and is compared with the numpy oracle
(`oracle.rl_loss.rl_step`, the restatement of reference rl/__init__.py:207-439) and an fp64 closed
form.  K1 backward and the step-scale K2+K3 launch (64 x 8192 tokens, flat_micro_batches) are
pinned to the oracle the same way."""

import ctypes
import functools
import itertools

import numpy as np
import pytest
import torch

from oracle import rl_loss as orl

from helpers import rel_err

pytestmark = pytest.mark.gpu

FP_TOL = 1e-4  # north_star: fp loss / grad within 1e-4 relative

CONFIGS = {
    # kl > 0, entropy bonus, temperature != 1: every labelled row carries a gradient
    "kl_ent_temp": dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.05, final_kl_coef=0.05,
                        entropy_bonus=0.01, final_entropy_bonus=0.01, temperature=0.7, batch_size=8,
                        clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False),
    # the BASELINE loss (conf/finetune/grpo.yaml): no kl, no entropy bonus -> clipped rows have zero gradient
    "grpo_clip": dict(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0,
                      temperature=1.0, batch_size=4096, clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False),
    "reinforce": dict(policy_loss="reinforce", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.02, final_kl_coef=0.02,
                      temperature=1.0, batch_size=64, clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False),
}


@functools.lru_cache(maxsize=2)
def _case(V: int, T: int, cfg_name: str, dtype: str):
    """Seeded logits + one packed micro-batch (two sequences) and the oracle's outputs."""
    cfg = CONFIGS[cfg_name]
    rng = np.random.default_rng(V + T + len(cfg_name))
    logits = (rng.standard_normal((1, T, V)) * 2).astype(np.float32)
    logits[0, 7] += 25.0 * (rng.random(V) < 0.0005)  # a spiky row: running-max rescale
    if dtype == "bf16":
        logits = torch.from_numpy(logits).to(torch.bfloat16).float().numpy()
    ids = rng.integers(3, V, size=(1, T), dtype=np.int64)
    ids[0, 11] = V - 1  # last column of the row (the scalar tail when V % 4 != 0)
    ids[0, 12] = 0
    half = T // 2
    pos = np.concatenate([np.arange(half), np.arange(T - half)])[None].astype(np.int64)
    labels = ids.copy()
    labels[0, :5] = -100
    labels[0, half:half + 4] = -100
    labels[0, rng.random(T) < 0.1] = -100
    # exact log-probs of the next tokens (fp64) to place old_logprobs relative to them
    z = logits[0, :-1].astype(np.float64) / cfg["temperature"]
    m = z.max(-1, keepdims=True)
    lse = np.log(np.exp(z - m).sum(-1)) + m[:, 0]
    nlp64 = np.concatenate([[0.0], z[np.arange(T - 1), ids[0, 1:]] - lse])
    old = nlp64 + rng.normal(0, 0.01, T)
    adv = rng.normal(0, 1, T)
    # rows 20.. : ratio far outside the clip range on the side where PPO clips (A > 0 and ratio > 1 + eps,
    # A < 0 and ratio < 1 - eps) -> d loss / d logits == 0 unless kl / entropy terms are on
    for t in range(20, min(30, T)):
        up = t % 2 == 0
        old[t] = nlp64[t] - (0.5 if up else -0.5)
        adv[t] = abs(adv[t]) + 0.1 if up else -abs(adv[t]) - 0.1
    ref = old + rng.normal(0, 0.05, T)
    f32 = lambda a: np.asarray(a, dtype=np.float32)[None]  # noqa: E731
    batch = {
        "input_ids": ids, "labels": labels, "position_ids": pos, "attention_mask": np.ones_like(ids),
        "segment_ids": (np.arange(T) >= half).astype(np.int64)[None],
        "old_logprobs": f32(old), "ref_logprobs": f32(ref), "advantages": f32(adv), "rewards": f32(rng.integers(0, 2, T)),
        "group_tokens": f32(np.full(T, 37.0)), "num_labels": f32(np.full(T, float((labels != -100).sum()))),
        "overflow": f32(np.zeros(T)),
    }
    want = orl.rl_step(logits, batch, cfg, 2, 10, True)
    # fp64 witness of d loss / d logits built from the oracle's per-token gradients
    zt = torch.from_numpy(logits[0, :-1]).double() / cfg["temperature"]
    logp = torch.log_softmax(zt, -1)
    p = logp.exp()
    H = -(p * logp).sum(-1, keepdim=True)
    g = torch.from_numpy(want["g_nlp"][0]).double()[:, None]
    gh = torch.from_numpy(want["g_ent"][0]).double()[:, None]
    onehot = torch.zeros_like(p)
    onehot[torch.arange(T - 1), torch.from_numpy(ids[0, 1:])] = 1.0
    g64 = torch.zeros(T, V, dtype=torch.float64)
    g64[:-1] = (g * (onehot - p) - gh * p * (logp + H)) / cfg["temperature"]
    return logits, batch, want, g64.numpy()


def _launch_fused(lib, dev, logits_t, batch, cfg_name, inplace):
    from pipelinerl_amd import _lib
    from pipelinerl_amd.finetune.rl import RLConfig, make_loss_config

    c_cfg, _, _ = make_loss_config(RLConfig(**CONFIGS[cfg_name]), 2, 10)
    T, V = logits_t.shape[1], logits_t.shape[2]
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in batch.items()}
    nlp, ent, lse = (torch.full((1, T), 7.0, device=dev) for _ in range(3))
    src = logits_t.clone()
    grad = src if inplace else torch.full_like(src, 3.0)
    _lib.check(lib.prl_fused_logits_loss(
        ctypes.byref(c_cfg), 1, T, V, src.data_ptr(), 0 if src.dtype == torch.float32 else 1, V, CONFIGS[cfg_name]["temperature"],
        d["input_ids"].data_ptr(), d["labels"].data_ptr(), d["old_logprobs"].data_ptr(), d["ref_logprobs"].data_ptr(),
        d["advantages"].data_ptr(), d["rewards"].data_ptr(), d["group_tokens"].data_ptr(), d["overflow"].data_ptr(),
        nlp.data_ptr(), ent.data_ptr(), lse.data_ptr(), grad.data_ptr(), _lib.current_stream_ptr(dev)))
    torch.cuda.synchronize()
    return nlp, ent, grad, lib.prl_last_fused_kernel().decode()


def _check_against_oracle(nlp, ent, grad, want, g64, dtype, cfg_name):
    g_nlp, g_ent = nlp.cpu().numpy(), ent.cpu().numpy()
    assert g_nlp[0, 0] == 0 and g_ent[0, 0] == 0
    np.testing.assert_allclose(g_nlp[:, 1:], want["new_logprobs"], rtol=FP_TOL, atol=2e-5)
    np.testing.assert_allclose(g_ent[:, 1:], want["entropy"], rtol=FP_TOL, atol=2e-5)
    got = grad.float().cpu().numpy()
    tol = FP_TOL if dtype == "f32" else 1e-2  # a bf16 gradient is rounded to 8 bits of mantissa
    assert rel_err(got, want["grad_logits"]) <= tol
    assert rel_err(got[0], g64) <= tol
    # masked rows, the last row and (for the plain GRPO loss) PPO-clipped rows are exactly zero
    zero_rows = np.flatnonzero(np.abs(want["g_nlp"][0]) + np.abs(want["g_ent"][0]) == 0)
    assert np.count_nonzero(got[0, zero_rows]) == 0 and np.count_nonzero(got[0, -1]) == 0
    if cfg_name == "grpo_clip":
        assert set(range(19, 29)) <= set(zero_rows.tolist()), "the crafted clipped rows must have zero gradient"
    assert np.count_nonzero(got) > 0


@pytest.mark.parametrize("vocab,cfg_name,dtype,inplace", list(itertools.product(
    [152064, 151936], ["kl_ent_temp", "grpo_clip", "reinforce"], ["f32", "bf16"], [False, True])))
def test_default_fused_dispatch_vs_oracle(libprl, cuda_device, monkeypatch, vocab, cfg_name, dtype, inplace):
    """PRL_FUSED_VARIANT unset: the kernel bench.py times (row-resident for fp32, two-sweep for bf16)."""
    monkeypatch.delenv("PRL_FUSED_VARIANT", raising=False)
    T = 64
    logits, batch, want, g64 = _case(vocab, T, cfg_name, dtype)
    lt = torch.from_numpy(logits).to(cuda_device)
    if dtype == "bf16":
        lt = lt.to(torch.bfloat16)
    nlp, ent, grad, kernel = _launch_fused(libprl, cuda_device, lt, batch, cfg_name, inplace)
    print(f"[kernel] V={vocab} {dtype}: {kernel}")
    if dtype == "f32":
        assert kernel.startswith("fused_logits_loss_keep_kernel<F32,1024,2,16,9"), kernel
    else:
        assert kernel.startswith("fused_logits_loss_kernel<BF16,512"), kernel
    _check_against_oracle(nlp, ent, grad, want, g64, dtype, cfg_name)


@pytest.mark.parametrize("dtype,variant", list(itertools.product(["f32", "bf16"], [0, 4, 6, 21])))
def test_every_fused_variant_vs_oracle(libprl, cuda_device, monkeypatch, variant, dtype):
    """Each selectable launch geometry against the oracle itself (not against variant 0)."""
    monkeypatch.setenv("PRL_FUSED_VARIANT", str(variant))
    logits, batch, want, g64 = _case(152064, 48, "kl_ent_temp", dtype)
    lt = torch.from_numpy(logits).to(cuda_device)
    if dtype == "bf16":
        lt = lt.to(torch.bfloat16)
    nlp, ent, grad, kernel = _launch_fused(libprl, cuda_device, lt, batch, "kl_ent_temp", False)
    # a bf16 row is 19 008 sixteen-byte vectors: the 25-vector-per-lane head of the row-resident shape does not fit and falls back
    if variant == 21 and dtype == "f32":
        assert "keep_kernel" in kernel, kernel
    _check_against_oracle(nlp, ent, grad, want, g64, dtype, "kl_ent_temp")


def test_short_and_unaligned_rows_fall_back(libprl, cuda_device, monkeypatch):
    """Rows the resident kernel cannot take (V too short for the on-chip head, or a row stride that
    breaks 16-byte alignment) use the two-sweep kernel - same answers."""
    monkeypatch.delenv("PRL_FUSED_VARIANT", raising=False)
    for V in (4099, 102396, 102401):
        logits, batch, want, g64 = _case(V, 40, "kl_ent_temp", "f32")
        nlp, ent, grad, kernel = _launch_fused(libprl, cuda_device, torch.from_numpy(logits).to(cuda_device), batch, "kl_ent_temp", False)
        assert kernel.startswith("fused_logits_loss_kernel<F32"), (V, kernel)
        _check_against_oracle(nlp, ent, grad, want, g64, "f32", "kl_ent_temp")


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_rl_step_at_full_vocab_and_scaled_loss(libprl, cuda_device, monkeypatch, dtype):
    """The drop-in rl_step on a full-vocabulary micro-batch: loss, every statistic and d loss/d logits
    against the oracle; then the same with the loss scaled before backward (accelerate's
    1/accumulation): once announced through `expected_loss_scale` (folded into the launch) and once
    unannounced (repaired on the device by prl_scale_unless) - both without a host sync in backward."""
    import types

    from pipelinerl_amd.finetune.rl import RLConfig, rl_step
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    monkeypatch.delenv("PRL_FUSED_VARIANT", raising=False)
    logits, batch, want, _ = _case(152064, 64, "kl_ent_temp", dtype)
    pb = PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(cuda_device)
    tol = FP_TOL if dtype == "f32" else 1e-2
    for expected, applied in ((1.0, 1.0), (0.25, 0.25), (1.0, 0.25), (0.5, 3.0)):
        lt = torch.from_numpy(logits).to(cuda_device)
        if dtype == "bf16":
            lt = lt.to(torch.bfloat16)
        lt.requires_grad_(True)
        model = lambda **kw: types.SimpleNamespace(logits=lt)  # noqa: E731
        cfg = RLConfig(**CONFIGS["kl_ent_temp"], expected_loss_scale=expected)
        loss, stats = rl_step(model, pb, 2, 10, cfg)
        (loss * applied).backward()
        assert abs(loss.item() - float(want["loss"])) <= FP_TOL * abs(float(want["loss"]))
        for k, w in want["stats"].items():
            assert abs(float(stats[k]) - float(w)) <= FP_TOL * max(abs(float(w)), 1.0), k
        assert rel_err(lt.grad.float().cpu().numpy(), want["grad_logits"] * applied) <= tol, (expected, applied)


@pytest.mark.parametrize("V,T", [(152064, 40), (1003, 50)])
@pytest.mark.parametrize("inplace", [False, True])
def test_unlabelled_rows_are_not_read(libprl, cuda_device, V, T, inplace):
    """`prl_loss_config.skip_unlabelled`: rows whose next token carries no label are not read - their log-prob / entropy
    come out as 0 and their gradient rows as zeros; everything else is bit for bit the skip = 0 result (row-resident
    kernel at V = 152 064, two-sweep kernel at V = 1003)."""
    from pipelinerl_amd import _lib
    from pipelinerl_amd.finetune.rl import RLConfig, make_loss_config

    logits, batch, want, g64 = _case(V, T, "kl_ent_temp", "f32")
    lt = torch.from_numpy(logits).to(cuda_device)
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(cuda_device) for k, v in batch.items()}
    out = {}
    for skip in (0, 1):
        c_cfg, _, _ = make_loss_config(RLConfig(**CONFIGS["kl_ent_temp"]), 2, 10)
        c_cfg.skip_unlabelled = skip
        nlp, ent, lse = (torch.full((1, T), 7.0, device=cuda_device) for _ in range(3))
        src = lt.clone()
        src[0, 3] = float("nan")  # row 3 predicts token 4, which is unlabelled: with skip = 1 the row must not be touched
        grad = src if inplace else torch.full_like(src, 3.0)
        _lib.check(libprl.prl_fused_logits_loss(
            ctypes.byref(c_cfg), 1, T, V, src.data_ptr(), 0, V, CONFIGS["kl_ent_temp"]["temperature"], d["input_ids"].data_ptr(),
            d["labels"].data_ptr(), d["old_logprobs"].data_ptr(), d["ref_logprobs"].data_ptr(), d["advantages"].data_ptr(),
            d["rewards"].data_ptr(), d["group_tokens"].data_ptr(), d["overflow"].data_ptr(), nlp.data_ptr(), ent.data_ptr(),
            lse.data_ptr(), grad.data_ptr(), _lib.current_stream_ptr(cuda_device)))
        torch.cuda.synchronize()
        out[skip] = (nlp.cpu().numpy(), ent.cpu().numpy(), grad.cpu().numpy())
    lab = np.asarray(batch["labels"])[0] != -100
    assert not lab[4]
    a, b = out[0], out[1]
    assert np.array_equal(a[0][0, lab], b[0][0, lab]) and np.array_equal(a[1][0, lab], b[1][0, lab])
    assert not b[0][0, ~lab].any() and not b[1][0, ~lab].any()
    assert np.isnan(a[0][0, 4]) and b[0][0, 4] == 0  # skip = 0 computed a log-prob from the NaN row, skip = 1 never read it
    keep = np.ones(T, dtype=bool)
    keep[3] = False
    assert np.array_equal(a[2][0, keep], b[2][0, keep])  # every gradient row, bit for bit
    assert not b[2][0, 3].any()
    np.testing.assert_allclose(b[0][0, 1:][lab[1:]], want["new_logprobs"][0][lab[1:]], rtol=FP_TOL, atol=2e-5)


def test_sentinel_batch_skips_the_logits_kernel(libprl, cuda_device):
    """finetune_loop.py:784-786: a sentinel batch is forwarded, its loss multiplied by 0 and
    back-propagated.  The fused path neither reads the logits nor launches the [T, V] kernel for it;
    the gradient is an exact zero tensor (also with NaN in the logits, which the reference's
    `0 * nan` would have turned into NaN gradients)."""
    import types

    from pipelinerl_amd.finetune.rl import RLConfig, rl_step
    from pipelinerl_amd.finetune.utils import create_sentinel_batch

    class Tok:
        eos_token_id = 5
        padding_side = "right"

    batch = create_sentinel_batch(cuda_device, tokenizer=Tok(), model_version=1)
    for inplace in (False, True):
        lt = torch.randn(1, 8, 152064, device=cuda_device)
        lt.requires_grad_(True)
        before = libprl.prl_last_fused_kernel()
        loss, stats = rl_step(lambda **kw: types.SimpleNamespace(logits=lt), batch, 0, 10,
                              RLConfig(batch_size=4, inplace_logits_grad=inplace))
        assert libprl.prl_last_fused_kernel() == before
        assert stats == {"input_size": 8.0} and loss.item() == 0.0
        (loss * 0.0).backward()
        assert torch.count_nonzero(lt.grad).item() == 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("vocab", [152064, 151936])
def test_logits_backward_full_vocab_vs_fp64(libprl, cuda_device, vocab, dtype):
    """K1 backward (with an entropy gradient, an upstream device scalar and temperature != 1) at
    the real vocabulary against the fp64 closed form."""
    from pipelinerl_amd import _lib
    from pipelinerl_amd.finetune.rl import logprob_entropy

    torch.manual_seed(vocab)
    B, L, V, temp = 2, 24, vocab, 0.8
    logits = (torch.randn(B, L, V, device=cuda_device) * 2).to(dtype)
    ids = torch.randint(0, V, (B, L), device=cuda_device)
    g = torch.randn(B, L, device=cuda_device)
    gh = torch.randn(B, L, device=cuda_device) * 0.1
    g[:, 0] = 0
    gh[:, 0] = 0
    g[0, 5] = 0
    gh[0, 5] = 0  # a row without gradient: written as zeros, logits not read
    up = torch.tensor(0.5, device=cuda_device)
    nlp, ent, lse2, lg = logprob_entropy(logits, ids, temp)
    for inplace in (False, True):
        src = lg.clone()
        grad = src if inplace else torch.empty_like(src)
        _lib.check(libprl.prl_logprob_entropy_bwd(
            B, L, V, src.data_ptr(), 0 if dtype == torch.float32 else 1, V, ids.data_ptr(), temp, lse2.data_ptr(), ent.data_ptr(),
            g.data_ptr(), gh.data_ptr(), up.data_ptr(), grad.data_ptr(), _lib.current_stream_ptr(cuda_device)))
        z = logits[:, :-1].double() / temp
        logp = torch.log_softmax(z, -1)
        p = logp.exp()
        H = -(p * logp).sum(-1, keepdim=True)
        onehot = torch.zeros_like(p).scatter_(2, ids[:, 1:, None], 1.0)
        want = torch.zeros(B, L, V, dtype=torch.float64, device=cuda_device)
        want[:, :-1] = 0.5 * (g[:, 1:, None].double() * (onehot - p) - gh[:, 1:, None].double() * p * (logp + H)) / temp
        tol = 1e-4 if dtype == torch.float32 else 1e-2
        assert rel_err(grad.double().cpu().numpy(), want.cpu().numpy()) <= tol
        assert torch.count_nonzero(grad[0, 4]).item() == 0 and torch.count_nonzero(grad[:, -1]).item() == 0
        # forward values against fp64 too
        w_nlp = logp.gather(2, ids[:, 1:, None])[..., 0]
        assert torch.allclose(nlp[:, 1:].double(), w_nlp, rtol=1e-4, atol=2e-5)
        assert torch.allclose(ent[:, 1:].double(), H[..., 0], rtol=1e-4, atol=2e-5)


@pytest.mark.parametrize("n_mb", [64, pytest.param(4096, marks=pytest.mark.slow)], ids=["64_micro_batches", "whole_step_4096_micro_batches"])
def test_step_scale_loss_launch_vs_oracle(libprl, cuda_device, n_mb):
    """K2+K3 as bench.py launches it - ONE launch over a flat [1, n_mb x 8192] step batch
    (flat_micro_batches = 1, statistics only) - against the oracle evaluated per micro-batch and
    combined the way the reference aggregates (sums add in fp64, max/min combine).  4096 micro-batches = the
    33 554 432 tokens of one BASELINE step (bs 4096 x seq 8192): the exact launch the benchmark times."""
    from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, make_loss_config
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    T = 8192
    cfg_d = dict(CONFIGS["grpo_clip"], kl_coef=0.001, final_kl_coef=0.001)
    rng = np.random.default_rng(11)
    cols: dict[str, list] = {}
    tot_loss, agg = 0.0, {}
    for j in range(n_mb):
        # 1-3 sequences per micro-batch, prompts masked
        cuts = sorted(rng.choice(np.arange(256, T - 256), size=int(rng.integers(0, 3)), replace=False).tolist())
        bounds = [0] + cuts + [T]
        pos = np.concatenate([np.arange(b - a) for a, b in zip(bounds[:-1], bounds[1:])])[None].astype(np.int64)
        ids = rng.integers(3, 152064, size=(1, T), dtype=np.int64)
        labels = ids.copy()
        for a in bounds[:-1]:
            labels[0, a:a + int(rng.integers(16, 200))] = -100
        old = (-np.abs(rng.standard_normal(T)) * 0.7).astype(np.float32)[None]
        b = {
            "input_ids": ids, "labels": labels, "position_ids": pos,
            "old_logprobs": old, "ref_logprobs": (old + rng.normal(0, 0.05, (1, T))).astype(np.float32),
            "advantages": rng.normal(0, 1, (1, T)).astype(np.float32), "rewards": rng.integers(0, 2, (1, T)).astype(np.float32),
            "group_tokens": np.full((1, T), 5000.0, np.float32), "num_labels": np.full((1, T), float((labels != -100).sum()), np.float32),
            "overflow": np.zeros((1, T), np.float32),
        }
        nlp = (old + rng.normal(0, 0.02, (1, T))).astype(np.float32)
        ent = rng.uniform(0, 3, (1, T)).astype(np.float32)
        res = orl.token_loss(b, nlp[:, 1:], ent[:, 1:], cfg_d, 0, 10, True)
        tot_loss += float(res["loss"])
        for k, v in res["stats"].items():
            agg.setdefault(k, []).append(float(v))
        for k, v in b.items():
            cols.setdefault(k, []).append(v)
        cols.setdefault("nlp", []).append(nlp)
        cols.setdefault("ent", []).append(ent)
    cat = {k: torch.from_numpy(np.concatenate(v, axis=1)).to(cuda_device) for k, v in cols.items()}
    nlp_d, ent_d = cat.pop("nlp"), cat.pop("ent")
    big = PipelineBatchEncoding(**cat, attention_mask=torch.ones_like(cat["input_ids"]), model_version=0, is_packed=True)
    c_cfg, _, _ = make_loss_config(RLConfig(**cfg_d), 0, 10)
    c_cfg.flat_micro_batches = 1
    loss, stats, _, _ = grpo_loss_from_logprobs(c_cfg, big, nlp_d, ent_d, want_grad=False)
    s = stats.cpu().numpy()
    from pipelinerl_amd._lib import STAT_INDEX as SI

    assert abs(loss.item() - tot_loss) <= FP_TOL * abs(tot_loss)
    assert int(s[SI["num_output_tokens_sum"]]) == int(sum(agg["num_output_tokens_sum"]))
    assert int(s[SI["num_sequences"]]) == int(round(sum(agg["kl_coef"]) / 0.001))
    for k in ("reward", "entropy", "old_logprobs", "new_logprobs", "ref_logprobs", "advantage", "kl", "kl_new_old",
              "mean_abs_log_ratio_new_old", "ratio_new_old", "ratio_new_old_sum", "ratio_new_old_squared_sum", "ratio_ref_new",
              "ratio_ref_old", "clamp_log_ratio_ref_new_indicator", "clamp_log_ratio_new_old_indicator", "token_weight"):
        w = sum(agg[k])
        assert abs(s[SI[k]] - w) <= FP_TOL * max(abs(w), 1.0), (k, s[SI[k]], w)
    for k in ("max_reward", "max_advantage", "max_kl", "max_token_weight"):
        assert abs(s[SI[k]] - max(agg[k])) <= 1e-6 * max(abs(max(agg[k])), 1.0), k
    for k in ("min_reward", "min_advantage", "min_kl", "min_token_weight"):
        assert abs(s[SI[k]] - min(agg[k])) <= 1e-6 * max(abs(min(agg[k])), 1.0), k
