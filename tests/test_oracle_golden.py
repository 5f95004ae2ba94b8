"""Pin the CPU oracle against the golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only."""

import numpy as np
import pytest

from oracle import preprocess as opre
from oracle import rl_loss as orl

from helpers import PREPROCESS_CASES, RL_STEP_CASES, assert_batch_equal, load_preprocess_case, load_rl_case, rel_err


@pytest.mark.parametrize("name", RL_STEP_CASES)
def test_rl_step_oracle_matches_reference(name):
    case = load_rl_case(name)
    cur, mx = case["steps"]
    res = orl.rl_step(case["logits"], case["batch"], case["config"], cur, mx, bool(case["batch"]["is_packed"]), value=case.get("value"))
    assert res["finite"]
    if "value" in case:  # value-head branch: d loss / d outputs.value vs the reference's autograd
        if "sentinel" in name:  # every label masked: the one-key dict and no gradient anywhere (rl/__init__.py:388-392)
            assert list(case["stats"]) == ["input_size"] and np.abs(case["grad_value"]).max() == 0
        else:
            assert len(case["stats"]) == 37 and np.abs(case["grad_value"]).max() > 0
        np.testing.assert_allclose(res["g_value"], case["grad_value"], rtol=1e-5, atol=1e-9)
    # fp: loss / stats within 1e-5 relative of the reference's fp32 torch result
    assert abs(float(res["loss"]) - case["loss"]) <= 1e-5 * max(1.0, abs(case["loss"]))
    assert list(res["stats"].keys()) == list(case["stats"].keys())
    for k, want in case["stats"].items():
        got = float(res["stats"][k])
        assert abs(got - want) <= 2e-5 * max(1.0, abs(want)), f"{k}: {got} vs {want}"
    # closed-form gradient vs reference autograd
    assert rel_err(res["grad_logits"], case["grad_logits"]) <= 1e-5 or np.abs(case["grad_logits"]).max() == 0
    np.testing.assert_allclose(res["grad_logits"], case["grad_logits"], rtol=0, atol=1e-6 * max(1.0, np.abs(case["grad_logits"]).max()))


@pytest.mark.parametrize("leg", ["rl_step", "rl_step_closed_form"])
@pytest.mark.parametrize("name", RL_STEP_CASES)
def test_rl_step_torch_legs_match_reference(name, leg):
    """The multi-threaded torch legs (autograd / closed form; the latter is bench.py's cpu_baseline): same pins."""
    from oracle import rl_loss_torch as orlt

    case = load_rl_case(name)
    cur, mx = case["steps"]
    res = getattr(orlt, leg)(case["logits"], case["batch"], case["config"], cur, mx, bool(case["batch"]["is_packed"]), value=case.get("value"))
    assert abs(float(res["loss"]) - case["loss"]) <= 1e-5 * max(1.0, abs(case["loss"]))
    for k, want in case["stats"].items():
        assert abs(float(res["stats"][k]) - want) <= 2e-5 * max(1.0, abs(want)), k
    got = res["grad_logits"].numpy()
    np.testing.assert_allclose(got, case["grad_logits"], rtol=0, atol=1e-6 * max(1.0, np.abs(case["grad_logits"]).max()))


@pytest.mark.parametrize("name", PREPROCESS_CASES)
def test_populate_oracle_matches_reference(name):
    case = load_preprocess_case(name)
    data = opre.preprocess_chunk(case["raw"], case["eos_token_id"], case["divide_advantage_by_std"])
    got = {
        "advantage": np.array([e["advantages"][0] for e in data]),
        "group_tokens": np.array([e["group_tokens"][0] for e in data]),
        "overflow": np.array([e["overflow"][0] for e in data]),
        "num_labels": np.array([e["num_labels"][0] for e in data], dtype=np.float64),
    }
    for k, want in case["scalars"].items():
        np.testing.assert_allclose(got[k], want, rtol=1e-12, atol=1e-15, err_msg=k)


@pytest.mark.parametrize("name", PREPROCESS_CASES)
def test_collate_oracle_matches_reference(name):
    case = load_preprocess_case(name)
    data = opre.preprocess_chunk(case["raw"], case["eos_token_id"], case["divide_advantage_by_std"])
    for plan, want in case["packed"].items():
        idxs = [int(i) for i in want["__idx"]]
        got = opre.collate_packed([data[i] for i in idxs], case["eos_token_id"], int(want["__seq_parallel"]))
        assert_batch_equal(got, want)
    for side, want in case["padded"].items():
        idxs = [int(i) for i in want["__idx"]]
        got = opre.collate([data[i] for i in idxs], padding_side=side)
        assert_batch_equal(got, want)


def test_sentinel_oracle_matches_reference():
    from helpers import GOLDEN
    import json

    z = np.load(GOLDEN / "sentinel.npz")
    got = opre.sentinel_batch(eos_token_id=7, model_version=5)
    assert_batch_equal(got, {k: z[k] for k in z.files})
    want_ex = json.loads((GOLDEN / "sentinel_example.json").read_text())
    assert opre.sentinel_example(3, 7, 9) == want_ex


def test_value_head_closed_form_gradient_is_the_derivative_of_the_oracle_loss():
    """Second witness for `g_value` (the goldens compare it with the reference's autograd): central differences of the
    oracle's own fp32 loss along random directions of the value predictions, accumulated in fp64 over the labelled tokens."""
    case = load_rl_case("c18_ppo_value_head")
    cur, mx = case["steps"]
    b, cfg = case["batch"], dict(case["config"], value_loss_coef=0.7)
    nlp, ent, _, _ = orl.logprob_entropy(case["logits"], b["input_ids"], cfg.get("temperature", 1.0))
    base = orl.token_loss(b, nlp, ent, cfg, cur, mx, True, value=case["value"])
    g = base["g_value"].astype(np.float64)
    rng = np.random.default_rng(0)
    for _ in range(3):
        d = rng.standard_normal(case["value"].shape)
        # advantages = rewards - V is DETACHED in the reference (:274): only the value loss moves with V, so evaluate it alone
        lo, hi = (orl.token_loss(b, nlp, ent, cfg, cur, mx, True, value=(case["value"] + s * 1e-2 * d).astype(np.float32))["stats"]["value_loss"]
                  for s in (-1.0, 1.0))
        numeric = cfg["value_loss_coef"] * (hi - lo) / 2e-2
        analytic = float((g * d).sum())
        assert abs(numeric - analytic) <= 2e-3 * max(1.0, abs(analytic)), (numeric, analytic)
    assert np.count_nonzero(g[:, :-1][(b["labels"] == -100)[:, 1:]]) == 0 and np.all(g[:, -1] == 0)
