"""The per-token formulas the HIP kernels execute (pipelinerl_amd/csrc/prl_token_math.h), compiled
for the host with g++ and compared with the oracle and the reference's golden gradients.  This
checks the device MATH on a machine without a GPU; the kernels themselves are covered by the
`-m gpu` tests."""

import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

from oracle import rl_loss as orl
from pipelinerl_amd._lib import PrlLossConfig
from pipelinerl_amd.finetune.rl import RLConfig, make_loss_config

from helpers import RL_STEP_CASES, load_rl_case

ROOT = Path(__file__).resolve().parent.parent
F = ctypes.POINTER(ctypes.c_float)


@pytest.fixture(scope="module")
def tokmath(tmp_path_factory):
    out = tmp_path_factory.mktemp("harness") / "libtokmath.so"
    subprocess.check_call(
        ["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", f"-I{ROOT / 'include'}", "-o", str(out),
         str(ROOT / "tests" / "harness" / "token_math_host.cpp")]
    )
    lib = ctypes.CDLL(str(out))
    lib.prl_host_token_eval.restype = None
    lib.prl_host_token_eval.argtypes = [ctypes.POINTER(PrlLossConfig), ctypes.c_long] + [F] * 15
    return lib


@pytest.mark.parametrize("name", [c for c in RL_STEP_CASES if "sentinel" not in c and "gspo" not in c])
def test_device_token_math_matches_oracle(tokmath, name):
    case = load_rl_case(name)
    cur, mx = case["steps"]
    cfg = RLConfig(**case["config"])
    c_cfg, kl_coef, ent_coef = make_loss_config(cfg, cur, mx)
    b = case["batch"]
    nlp, ent, _, _ = orl.logprob_entropy(case["logits"], b["input_ids"], cfg.temperature)
    ref = orl.token_loss(b, nlp, ent, case["config"], cur, mx, bool(b["is_packed"]), value=case.get("value"))
    mask = (b["labels"] != -100)[:, 1:]
    want_loss = case["loss"]
    if "value" in case:  # value head: the advantages column is rewards - V, the value loss is a separate kernel's
        b = dict(b, advantages=np.concatenate([np.zeros_like(b["rewards"][:, :1]), b["rewards"][:, 1:] - case["value"][:, :-1]], axis=1))
        want_loss = case["loss"] - case["config"]["value_loss_coef"] * case["stats"]["value_loss"]
    sel = lambda a: np.ascontiguousarray(a[mask], dtype=np.float32)  # noqa: E731
    sh = lambda k: sel(b[k][:, 1:])  # noqa: E731
    ins = [sel(nlp), sel(ent), sh("old_logprobs"), sh("ref_logprobs"), sh("advantages"), sh("rewards"),
           sh("group_tokens"), sh("num_labels"), sh("overflow")]
    n = int(mask.sum())
    outs = [np.zeros(n, dtype=np.float32) for _ in range(6)]
    tokmath.prl_host_token_eval(ctypes.byref(c_cfg), n, *[a.ctypes.data_as(F) for a in ins + outs])
    contrib, g_nlp, g_ent, ratio_stat, kl, clamp_no = outs
    # loss = -sum(contrib) (fp64 accumulate like the kernel)
    loss = -float(contrib.astype(np.float64).sum())
    assert abs(loss - want_loss) <= 1e-5 * max(1.0, abs(want_loss))
    np.testing.assert_allclose(g_nlp, ref["g_nlp"][mask], rtol=1e-4, atol=1e-8)  # cancelling terms: fp32 op order
    np.testing.assert_allclose(g_ent, ref["g_ent"][mask], rtol=2e-6, atol=1e-9)
    want = case["stats"]
    nl = sh("num_labels").astype(np.float64)
    assert abs((ratio_stat / nl).sum() - want["ratio_new_old"]) <= 2e-5 * max(1.0, abs(want["ratio_new_old"]))
    assert abs((kl / nl).sum() - want["kl"]) <= 2e-5 * max(1.0, abs(want["kl"]))
    assert abs((clamp_no / nl).sum() - want["clamp_log_ratio_new_old_indicator"]) <= 2e-5
    assert abs(float(kl.max()) - want["max_kl"]) <= 1e-6 * max(1.0, abs(want["max_kl"]))
