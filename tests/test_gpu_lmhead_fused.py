"""Fused output head (csrc/prl_lmhead_{core.h,fwd.hip,bwd.hip,prepare.hip}, pipelinerl_amd/fused_head.py): the logits are never
written, so parity is established against the oracle FED WITH the fp64 product hidden @ W^T:

    logits64 = hidden.double() @ W.double().T   (torch, on the GPU, fp64)
    oracle.rl_loss.rl_step(logits64 -> fp32)    -> loss, statistics, d loss / d logits
    d hidden = d logits @ W,  d W = d logits^T @ hidden   (fp64)

at the BASELINE head shapes (H = 3584, V = 152 064: Qwen2.5-7B, fp32 weight split into two bf16
planes; H = 896, V = 151 936: Qwen2.5-0.5B with its tied bf16 weight), plus small ragged shapes for
the tile edges, all three workgroup shapes (256 x 256, 256 x 128 with the 3-stage ring, 128 x 128) and chunked backward.
Tolerance: 1e-4 relative (north_star), on loss, d hidden and d W."""

import ctypes

import numpy as np
import pytest
import torch

from oracle import rl_loss as orl

from helpers import rel_err

pytestmark = pytest.mark.gpu

FP_TOL = 1e-4

CFG = dict(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.05, final_kl_coef=0.05, entropy_bonus=0.01,
           final_entropy_bonus=0.01, temperature=0.7, batch_size=8, clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False)


def _problem(T, H, V, dev, weight_dtype=torch.float32, seed=0, hidden_scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    hidden = (torch.randn(1, T, H, generator=g) * hidden_scale).to(torch.bfloat16).to(dev)
    W = (torch.randn(V, H, generator=g) * (2.0 / H ** 0.5)).to(weight_dtype).to(dev)
    rng = np.random.default_rng(seed + 1)
    ids = rng.integers(0, V, size=(1, T), dtype=np.int64)
    labels = ids.copy()
    labels[0, : max(2, T // 10)] = -100
    labels[0, rng.random(T) < 0.1] = -100
    half = T // 2
    pos = np.concatenate([np.arange(half), np.arange(T - half)])[None].astype(np.int64)
    labels[0, half] = -100
    f32 = lambda a: np.asarray(a, dtype=np.float32)[None]  # noqa: E731
    logits64 = hidden[0].double() @ W.double().t()
    lp = torch.log_softmax(logits64 / CFG["temperature"], -1)
    nlp = np.concatenate([[0.0], lp[torch.arange(T - 1), torch.from_numpy(ids[0, 1:]).to(dev)].cpu().numpy()])
    old = nlp + rng.normal(0, 0.05, T)
    batch = {
        "input_ids": ids, "labels": labels, "position_ids": pos, "attention_mask": np.ones_like(ids),
        "old_logprobs": f32(old), "ref_logprobs": f32(old + rng.normal(0, 0.05, T)), "advantages": f32(rng.normal(0, 1, T)),
        "rewards": f32(rng.integers(0, 2, T)), "group_tokens": f32(np.full(T, 31.0)),
        "num_labels": f32(np.full(T, float((labels != -100).sum()))), "overflow": f32(np.zeros(T)),
    }
    return hidden, W, batch, logits64


def _oracle(hidden, W, batch, logits64):
    want = orl.rl_step(logits64.float().cpu().numpy()[None], batch, CFG, 2, 10, True)
    dl = torch.from_numpy(want["grad_logits"][0]).to(hidden.device).double()
    want["d_hidden"] = (dl @ W.double()).cpu().numpy()
    want["d_weight"] = (dl.t() @ hidden[0].double()).cpu().numpy()
    return want


def _run(hidden, W, batch, chunk_rows=None, grad_scale=1.0, keep_logits=None):
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.fused_head import FusedLmHead, fused_head_loss

    dev = hidden.device
    pb = PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(dev)
    h = hidden.clone().requires_grad_(True)
    w = W.clone().requires_grad_(True)
    head = FusedLmHead(w, chunk_rows=chunk_rows or 4096, keep_logits=keep_logits)  # None: the default (kept logits)
    loss, stats = fused_head_loss(h, w, head, pb, RLConfig(**CFG), 2, 10, chunk_rows=chunk_rows)
    (loss * grad_scale).backward()
    torch.cuda.synchronize()
    return loss, stats, h.grad, w.grad, head


def _compare(loss, stats, gh, gw, want, scale=1.0):
    assert abs(loss.item() - float(want["loss"])) <= FP_TOL * abs(float(want["loss"]))
    for k, v in want["stats"].items():
        assert abs(float(stats[k]) - float(v)) <= FP_TOL * max(abs(float(v)), 1.0), k
    assert rel_err(gw.float().cpu().numpy(), want["d_weight"] * scale) <= FP_TOL
    # d hidden is delivered in the hidden states' dtype (bf16): compare before that rounding where possible
    tol_h = FP_TOL if gh.dtype == torch.float32 else 4e-3
    assert rel_err(gh[0].float().cpu().numpy(), want["d_hidden"] * scale) <= tol_h


@pytest.mark.parametrize("keep", [True, False], ids=["kept_logits", "recompute"])
@pytest.mark.parametrize("tile", ["256x256", "256", "128"], ids=["tile256x256", "tile256x128_ring3", "tile128x128"])
@pytest.mark.parametrize("T,H,V", [(130, 64, 192), (257, 128, 320), (64, 192, 4160), (300, 64, 1088), (900, 128, 1088)])
def test_small_ragged_shapes(libprl, cuda_device, monkeypatch, tile, T, H, V, keep):
    """Tile edges everywhere: rows not a multiple of 128, a partly masked last vocabulary tile, one K step
    (fewer tiles than pipeline stages), both workgroup shapes; the backward from the logits the forward kept and the
    backward that recomputes them."""
    monkeypatch.setenv("PRL_LMHEAD_TILE", tile)
    hidden, W, batch, logits64 = _problem(T, H, V, cuda_device, seed=T)
    want = _oracle(hidden, W, batch, logits64)
    loss, stats, gh, gw, _ = _run(hidden, W, batch, keep_logits=keep)
    _compare(loss, stats, gh, gw, want)


@pytest.mark.parametrize("V", [8192, 8200], ids=["whole_tiles", "partial_last_tile"])
def test_forward_values_and_split_count_independence(libprl, cuda_device, monkeypatch, V):
    """new_logprobs / entropy against fp64 directly, for every vocabulary-split count (the partial
    online-softmax states merge to the same answer; with few splits a workgroup sweeps many vocabulary tiles and
    the dual-plane core chains them without refilling its pipeline), with a spiky row that forces running-max
    rescales."""
    from pipelinerl_amd.fused_head import FusedLmHead

    T, H = 200, 256
    hidden, W, batch, logits64 = _problem(T, H, V, cuda_device, seed=5)
    W = W.clone()
    W[4000] = hidden[0, 17].float() * 0.5  # token 17's logit for id 4000 towers over the rest (~ |h|^2 / 2)
    logits64 = hidden[0].double() @ W.double().t()
    ids = torch.from_numpy(batch["input_ids"]).to(cuda_device)
    z = logits64 / 0.9
    lp = torch.log_softmax(z, -1)
    w_nlp = lp[torch.arange(T - 1), ids[0, 1:]]
    w_ent = -(lp.exp() * lp).sum(-1)
    head = FusedLmHead(W, backward=False)
    for tile, ns in (("256x256", "1"), ("256x256", "3"), ("256x256", "64"), ("256", "1"), ("256", "2"), ("256", "7"), ("256", "64"), ("128", "1"), ("128", "5"), ("128", "64")):
        monkeypatch.setenv("PRL_LMHEAD_NSPLIT", ns)
        monkeypatch.setenv("PRL_LMHEAD_TILE", tile)
        nlp, ent, lse2, _ = head.logprob_entropy(hidden, ids, 0.9)
        assert nlp[0, 0].item() == 0 and ent[0, 0].item() == 0
        assert torch.allclose(nlp[0, 1:].double(), w_nlp, rtol=FP_TOL, atol=2e-5), ns
        assert torch.allclose(ent[0, 1:].double(), w_ent[:-1], rtol=FP_TOL, atol=2e-5), ns


@pytest.mark.parametrize("T,H,V,wdt", [(700, 256, 1088, torch.float32), (520, 3584, 2048, torch.float32), (520, 896, 2048, torch.bfloat16)])
def test_hand_placed_streams_are_stable_run_to_run(libprl, cuda_device, monkeypatch, T, H, V, wdt):
    """The 256 x 256 forwards are hand-placed instruction streams (asm MFMAs the compiler's hazard recogniser does not see, LDS-DMA
    pieces and fragment reads pinned between them): the dual-plane phase-shifted stream for an fp32 weight, the generic 64-deep
    stream for a bf16 weight.  A race or a missed hazard in one of them shows up as a run-to-run difference or as a difference to
    the compiler-scheduled 256 x 128 shape beyond the summation order: outputs are identical to the bit run to run, kept logits
    included, and agree with the other shape to fp32 rounding."""
    from pipelinerl_amd.fused_head import FusedLmHead

    hidden, W, batch, _ = _problem(T, H, V, cuda_device, weight_dtype=wdt, seed=T + H)
    ids = torch.from_numpy(batch["input_ids"]).to(cuda_device)
    head = FusedLmHead(W, backward=False)
    monkeypatch.setenv("PRL_LMHEAD_TILE", "256x256")
    want = [t.clone() for t in head.logprob_entropy(hidden, ids, 0.9, keep=True)]
    for _ in range(4):
        got = head.logprob_entropy(hidden, ids, 0.9, keep=True)
        torch.cuda.synchronize()
        assert all(torch.equal(a, b) for a, b in zip(got[:3] + got[4:], want[:3] + want[4:]))
    monkeypatch.setenv("PRL_LMHEAD_TILE", "256")
    other = head.logprob_entropy(hidden, ids, 0.9, keep=True)
    assert torch.allclose(other[4], want[4], rtol=0, atol=1e-4 * float(want[4].abs().max()))
    assert torch.allclose(other[0], want[0], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("tile", ["256x256", "256", "128", None, "recompute"],
                         ids=["tile256x256", "tile256x128_ring3", "tile128x128", "default_dispatch", "default_dispatch_recompute"])
def test_qwen7b_head_shape_vs_oracle(libprl, cuda_device, monkeypatch, tile):
    """H = 3584, V = 152 064, fp32 weight (two bf16 planes): loss, statistics, d hidden, d W."""
    keep = tile != "recompute"  # the default keeps the logits for the backward; "recompute": no logits anywhere
    tile = None if tile == "recompute" else tile
    if tile is None:
        monkeypatch.delenv("PRL_LMHEAD_TILE", raising=False)
    else:
        monkeypatch.setenv("PRL_LMHEAD_TILE", tile)
    hidden, W, batch, logits64 = _problem(160, 3584, 152064, cuda_device, seed=3)
    want = _oracle(hidden, W, batch, logits64)
    loss, stats, gh, gw, head = _run(hidden, W, batch, keep_logits=keep)
    _compare(loss, stats, gh, gw, want)
    # fp32 d hidden straight from the C ABI: the 1e-4 bar without the bf16 rounding of the autograd path
    from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, make_loss_config
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    pb = PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(cuda_device)
    nlp, ent, lse2, h = head.logprob_entropy(hidden, pb.input_ids, CFG["temperature"])
    c_cfg, _, _ = make_loss_config(RLConfig(**CFG), 2, 10)
    _, _, g_nlp, g_ent = grpo_loss_from_logprobs(c_cfg, pb, nlp, ent)
    gh32 = head.backward_from_token_grads(h, pb.input_ids, CFG["temperature"], lse2, ent, g_nlp, g_ent, None, grad_hidden_dtype=torch.float32)
    assert rel_err(gh32[0].cpu().numpy(), want["d_hidden"]) <= FP_TOL


def test_qwen0p5b_tied_bf16_head_vs_oracle(libprl, cuda_device):
    """H = 896, V = 151 936, bf16 weight (tied embedding): a single exact plane."""
    hidden, W, batch, logits64 = _problem(150, 896, 151936, cuda_device, weight_dtype=torch.bfloat16, seed=9)
    want = _oracle(hidden, W, batch, logits64)
    loss, stats, gh, gw, head = _run(hidden, W, batch)
    assert head.w_lo is None
    assert abs(loss.item() - float(want["loss"])) <= FP_TOL * abs(float(want["loss"]))
    assert rel_err(gh[0].float().cpu().numpy(), want["d_hidden"]) <= 4e-3
    assert rel_err(gw.float().cpu().numpy(), want["d_weight"]) <= 4e-3  # delivered in the weight's dtype (bf16)


@pytest.mark.parametrize("tile", ["256x256", None], ids=["tile256x256", "default_dispatch"])
def test_chunked_backward_and_upstream_scale(libprl, cuda_device, monkeypatch, tile):
    """Row chunks that do not divide the batch (and are not multiples of 128) give the same gradients;
    an upstream factor on the loss scales both."""
    if tile:
        monkeypatch.setenv("PRL_LMHEAD_TILE", tile)
    else:
        monkeypatch.delenv("PRL_LMHEAD_TILE", raising=False)
    hidden, W, batch, logits64 = _problem(300, 128, 1024, cuda_device, seed=21)
    want = _oracle(hidden, W, batch, logits64)
    for chunk, scale in ((None, 1.0), (128, 1.0), (100, 0.25), (299, 3.0)):
        for keep in (True, False):
            loss, stats, gh, gw, _ = _run(hidden, W, batch, chunk_rows=chunk, grad_scale=scale, keep_logits=keep)
            _compare(loss, stats, gh, gw, want, scale)


@pytest.mark.parametrize("tile", ["256x256", "256", "128"], ids=["tile256x256", "tile256x128_ring3", "tile128x128"])
@pytest.mark.parametrize("ksplit", ["2", "3", "8"])
def test_split_k_hidden_gradient(libprl, cuda_device, monkeypatch, tile, ksplit):
    """d hidden contracts over the vocabulary; its split-K path (slices of the vocabulary reduced in a fixed order)
    gives the same result for even and uneven slices (65 steps of 64: 33 + 32, 22 + 22 + 21, 7 x 9 + 2)."""
    from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, make_loss_config
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.fused_head import FusedLmHead

    monkeypatch.setenv("PRL_LMHEAD_TILE", tile)
    hidden, W, batch, logits64 = _problem(300, 128, 4160, cuda_device, seed=33)
    want = _oracle(hidden, W, batch, logits64)
    pb = PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(cuda_device)
    head = FusedLmHead(W)
    nlp, ent, lse2, h = head.logprob_entropy(hidden, pb.input_ids, CFG["temperature"])
    c_cfg, _, _ = make_loss_config(RLConfig(**CFG), 2, 10)
    _, _, g_nlp, g_ent = grpo_loss_from_logprobs(c_cfg, pb, nlp, ent)
    out = {}
    for ks in ("1", ksplit):
        monkeypatch.setenv("PRL_LMHEAD_KSPLIT", ks)
        for dt in (torch.float32, torch.bfloat16):
            gh = head.backward_from_token_grads(h, pb.input_ids, CFG["temperature"], lse2, ent, g_nlp, g_ent, None, grad_hidden_dtype=dt)
            torch.cuda.synchronize()
            out[ks, dt] = gh[0].float().cpu().numpy()
            assert rel_err(out[ks, dt], want["d_hidden"]) <= (FP_TOL if dt == torch.float32 else 4e-3)
    # the slices are added in a fixed order: run to run the split result is bitwise stable
    gh = head.backward_from_token_grads(h, pb.input_ids, CFG["temperature"], lse2, ent, g_nlp, g_ent, None, grad_hidden_dtype=torch.float32)
    assert np.array_equal(gh[0].cpu().numpy(), out[ksplit, torch.float32])


@pytest.mark.parametrize("T,H,V,wdt", [(300, 128, 4160, torch.float32), (257, 64, 1088, torch.bfloat16)])
def test_kept_logits_are_the_logits_and_give_the_recomputed_gradients(libprl, cuda_device, T, H, V, wdt):
    """`logprob_entropy(keep=True)` leaves [T, V] fp32 logits in base-2 units (logit * log2(e) / temperature) next to the same
    three outputs; the backward from them equals the recomputing backward up to one fp32 rounding of the logit (the recompute
    fuses the scale into a multiply-add) - including vocabularies whose last tile is partial and rows without a gradient."""
    import math

    from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, make_loss_config
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.fused_head import FusedLmHead

    hidden, W, batch, logits64 = _problem(T, H, V, cuda_device, weight_dtype=wdt, seed=V)
    pb = PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(cuda_device)
    head = FusedLmHead(W, chunk_rows=128)
    plain = head.logprob_entropy(hidden, pb.input_ids, CFG["temperature"])
    nlp, ent, lse2, h, kept = head.logprob_entropy(hidden, pb.input_ids, CFG["temperature"], keep=True)
    for a, b in zip(plain[:3], (nlp, ent, lse2)):
        assert torch.equal(a, b)
    assert kept.shape == (T, V) and kept.dtype == torch.float32
    want2 = logits64 * (math.log2(math.e) / CFG["temperature"])
    tol = 2e-5  # of the largest logit: the plane split
    assert float((kept.double() - want2).abs().max()) <= tol * float(want2.abs().max())
    c_cfg, _, _ = make_loss_config(RLConfig(**CFG), 2, 10)
    _, _, g_nlp, g_ent = grpo_loss_from_logprobs(c_cfg, pb, nlp, ent)
    out = {}
    for name, kw in (("recompute", {}), ("kept", {"kept_logits": kept})):
        gw = torch.full((V, H), float("nan"), device=cuda_device)
        gh = head.backward_from_token_grads(h, pb.input_ids, CFG["temperature"], lse2, ent, g_nlp, g_ent, None, grad_weight=gw,
                                            overwrite_weight_grad=True, grad_hidden_dtype=torch.float32, **kw)
        torch.cuda.synchronize()
        out[name] = (gh[0].cpu().numpy(), gw.cpu().numpy())
    for a, b in zip(out["kept"], out["recompute"]):
        assert np.isfinite(a).all() and rel_err(a, b) <= 5e-6
    with pytest.raises(ValueError):
        head.backward_from_token_grads(h, pb.input_ids, CFG["temperature"], lse2, ent, g_nlp, g_ent, None, kept_logits=kept[:, :-8])


def test_leading_term_hidden_gradient(libprl, cuda_device):
    """hidden_grad_terms = 1: d hidden from d logits_hi x W_hi only - within the bf16 rounding of the full result."""
    from pipelinerl_amd.finetune.rl import RLConfig, grpo_loss_from_logprobs, make_loss_config
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.fused_head import FusedLmHead

    hidden, W, batch, logits64 = _problem(200, 256, 2048, cuda_device, seed=13)
    want = _oracle(hidden, W, batch, logits64)
    pb = PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(cuda_device)
    head = FusedLmHead(W, hidden_grad_terms=1)
    nlp, ent, lse2, h = head.logprob_entropy(hidden, pb.input_ids, CFG["temperature"])
    c_cfg, _, _ = make_loss_config(RLConfig(**CFG), 2, 10)
    _, _, g_nlp, g_ent = grpo_loss_from_logprobs(c_cfg, pb, nlp, ent)
    gh = head.backward_from_token_grads(h, pb.input_ids, CFG["temperature"], lse2, ent, g_nlp, g_ent, None, grad_hidden_dtype=torch.float32)
    err = rel_err(gh[0].cpu().numpy(), want["d_hidden"])
    assert 1e-5 < err <= 6e-3, err  # 2^-9 relative per product, not the 1e-4 of the three-term default


def test_rl_step_fused_head_on_a_huggingface_style_model(libprl, cuda_device):
    """`rl_step_fused_head` == `rl_step` on a model with `.model` (body) and `.lm_head`: same loss,
    same parameter gradients, without the [T, V] logits."""
    import copy
    import types

    from pipelinerl_amd.finetune.rl import RLConfig, rl_step
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.fused_head import rl_step_fused_head

    V, H, T = 1024, 128, 96

    class Body(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Embedding(V, H)
            self.lin = torch.nn.Linear(H, H)

        def forward(self, input_ids=None, **kw):
            return (torch.tanh(self.lin(self.emb(input_ids))).to(torch.bfloat16),)

    class LM(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model = Body()
            self.lm_head = torch.nn.Linear(H, V, bias=False)

        def forward(self, **kw):
            h = self.model(**kw)[0]
            return types.SimpleNamespace(logits=h.float() @ self.lm_head.weight.t())

    torch.manual_seed(0)
    a = LM().to(cuda_device)
    b = copy.deepcopy(a)
    _, _, batch, _ = _problem(T, H, V, cuda_device, seed=2)
    pb = PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(cuda_device)
    cfg = RLConfig(**CFG)
    la, sa = rl_step(a, pb, 2, 10, cfg)
    la.backward()
    lb, sb = rl_step_fused_head(b, pb, 2, 10, cfg)
    lb.backward()
    assert abs(la.item() - lb.item()) <= FP_TOL * abs(la.item())
    assert list(sa) == list(sb)
    for k in sa:
        assert abs(sa[k] - sb[k]) <= FP_TOL * max(abs(sa[k]), 1.0), k
    for (n, pa), (_, pbb) in zip(a.named_parameters(), b.named_parameters()):
        # the body's gradients pass through a bf16 d hidden in both models
        assert rel_err(pbb.grad.cpu().numpy(), pa.grad.cpu().numpy()) <= 2e-2, n
    assert rel_err(b.lm_head.weight.grad.cpu().numpy(), a.lm_head.weight.grad.cpu().numpy()) <= 1e-3


@pytest.mark.parametrize("cfg_name,cfg", [
    ("grpo_no_kl_no_entropy", dict(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.0, final_kl_coef=0.0, temperature=1.0,
                                   batch_size=4096, clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False)),
    ("reinforce_kl", dict(policy_loss="reinforce", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.02, final_kl_coef=0.02, temperature=1.3,
                          batch_size=64, clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False)),
])
def test_other_loss_configurations_vs_oracle(libprl, cuda_device, cfg_name, cfg):
    """The BASELINE GRPO loss (no KL, no entropy bonus: the entropy gradient is absent and PPO-clipped rows carry a zero
    gradient through the whole backward) and REINFORCE with KL at another temperature."""
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.fused_head import FusedLmHead, fused_head_loss

    T, H, V = 300, 128, 1088
    g = torch.Generator(device="cpu").manual_seed(77)
    hidden = torch.randn(1, T, H, generator=g).to(torch.bfloat16).to(cuda_device)
    W = (torch.randn(V, H, generator=g) * (2.0 / H ** 0.5)).to(cuda_device)
    rng = np.random.default_rng(78)
    ids = rng.integers(0, V, size=(1, T), dtype=np.int64)
    labels = ids.copy()
    labels[0, :30] = -100
    half = T // 2
    labels[0, half] = -100
    pos = np.concatenate([np.arange(half), np.arange(T - half)])[None].astype(np.int64)
    logits64 = hidden[0].double() @ W.double().t()
    lp = torch.log_softmax(logits64 / cfg["temperature"], -1)
    nlp = np.concatenate([[0.0], lp[torch.arange(T - 1), torch.from_numpy(ids[0, 1:]).to(cuda_device)].cpu().numpy()])
    old = nlp + rng.normal(0, 0.01, T)
    adv = rng.normal(0, 1, T)
    for t in range(40, 60):  # far outside the clip range on the clipping side: zero gradient when kl = entropy = 0
        up = t % 2 == 0
        old[t] = nlp[t] - (0.5 if up else -0.5)
        adv[t] = abs(adv[t]) + 0.1 if up else -abs(adv[t]) - 0.1
    f32 = lambda a: np.asarray(a, dtype=np.float32)[None]  # noqa: E731
    batch = {"input_ids": ids, "labels": labels, "position_ids": pos, "attention_mask": np.ones_like(ids), "old_logprobs": f32(old),
             "ref_logprobs": f32(old + rng.normal(0, 0.05, T)), "advantages": f32(adv), "rewards": f32(rng.integers(0, 2, T)),
             "group_tokens": f32(np.full(T, 31.0)), "num_labels": f32(np.full(T, float((labels != -100).sum()))), "overflow": f32(np.zeros(T))}
    want = orl.rl_step(logits64.float().cpu().numpy()[None], batch, cfg, 2, 10, True)
    dl = torch.from_numpy(want["grad_logits"][0]).to(cuda_device).double()
    if cfg_name == "grpo_no_kl_no_entropy":
        assert not want["grad_logits"][0][39:59].any()  # the clipped rows (row q predicts token q + 1)
    pb = PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(cuda_device)
    h = hidden.clone().requires_grad_(True)
    w = W.clone().requires_grad_(True)
    loss, stats = fused_head_loss(h, w, FusedLmHead(w), pb, RLConfig(**cfg), 2, 10)
    loss.backward()
    assert abs(loss.item() - float(want["loss"])) <= FP_TOL * abs(float(want["loss"]))
    for k, v in want["stats"].items():
        assert abs(float(stats[k]) - float(v)) <= FP_TOL * max(abs(float(v)), 1.0), k
    assert rel_err(w.grad.cpu().numpy(), (dl.t() @ hidden[0].double()).cpu().numpy()) <= FP_TOL
    assert rel_err(h.grad[0].float().cpu().numpy(), (dl @ W.double()).cpu().numpy()) <= 4e-3  # delivered in bf16
    if cfg_name == "grpo_no_kl_no_entropy":
        assert torch.count_nonzero(h.grad[0, 39:59]) == 0


@pytest.mark.parametrize("layout", ["packed", "padded_2_rows"])
def test_unlabelled_rows_are_skipped_without_changing_the_result(libprl, cuda_device, layout):
    """`skip_unlabelled` hands the kernels only the rows that predict a labelled token (here ~60 % of them: long prompts):
    loss, statistics and gradients equal the full computation and the oracle; d hidden of the skipped rows is exactly 0."""
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.fused_head import FusedLmHead, fused_head_loss

    T, H, V = 320, 128, 1088
    hidden, W, batch, logits64 = _problem(T, H, V, cuda_device, seed=55)
    batch = {k: v.copy() for k, v in batch.items()}
    batch["labels"][0, :70] = -100          # a long prompt in the first sequence
    batch["labels"][0, 160:230] = -100      # and in the second
    packed = layout == "packed"
    if not packed:  # the same tokens as two padded rows [2, 160]: row boundaries replace the packed sequence start
        batch = {k: v.reshape(2, T // 2) for k, v in batch.items()}
        batch.pop("position_ids")
        hidden = hidden.reshape(2, T // 2, H)
        logits64 = logits64.reshape(2, T // 2, V)
    nl = float((batch["labels"][:, 1:] != -100).sum())
    batch["num_labels"] = np.full_like(batch["num_labels"], nl)
    want = orl.rl_step(logits64.float().cpu().numpy().reshape(hidden.shape[0], hidden.shape[1], V), batch, CFG, 2, 10, packed)
    pb = PipelineBatchEncoding(**{k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in batch.items()}, model_version=0, is_packed=packed).to_device(cuda_device)
    out = {}
    for skip in (True, False):
        h = hidden.clone().requires_grad_(True)
        w = W.clone().requires_grad_(True)
        head = FusedLmHead(w, skip_unlabelled=skip)
        loss, stats = fused_head_loss(h, w, head, pb, RLConfig(**CFG), 2, 10)
        loss.backward()
        out[skip] = (loss.item(), stats, h.grad.float().cpu().numpy(), w.grad.cpu().numpy())
        assert abs(loss.item() - float(want["loss"])) <= FP_TOL * abs(float(want["loss"]))
        for k, v in want["stats"].items():
            assert abs(float(stats[k]) - float(v)) <= FP_TOL * max(abs(float(v)), 1.0), (skip, k)
    a, b = out[True], out[False]
    assert abs(a[0] - b[0]) <= 1e-6 * abs(b[0]) and list(a[1]) == list(b[1])
    assert rel_err(a[3], b[3]) <= 1e-5 and rel_err(a[2], b[2]) <= 1e-5
    dead = (np.concatenate([batch["labels"][:, 1:], np.full((batch["labels"].shape[0], 1), -100)], axis=1) == -100)
    assert dead.mean() > 0.5 and not a[2][dead].any() and not b[2][dead].any()
    dl = torch.from_numpy(want["grad_logits"]).to(cuda_device).double().reshape(-1, V)
    assert rel_err(a[3], (dl.t() @ hidden.reshape(-1, H).double()).cpu().numpy()) <= FP_TOL


