"""BASELINE.json configs[4] on the device: Qwen2.5-32B (H = 5120, V = 152 064, untied fp32 head), KL-to-reference on,
TP = 2 receivers.

  * the fused output head at H = 5120 against the oracle fed the fp64 product - a small case on every run and the whole
    8192-token micro-batch (`slow`, still part of `-m gpu`) with the config's own loss settings (kl_coef 0.001,
    ref != old; conf/deepscaler15b.yaml:34 over conf/finetune/grpo.yaml, rl/__init__.py:262-288);
  * the reference-policy forward through the MFMA head (no `[T, V]` logits) against `oracle.rl_loss.logprob_entropy`;
  * the 771-tensor / 65.5 GB parameter set through the bucket gather / scatter kernels, byte for byte;
  * the TP = 2 sharded update at the 32B layer shapes (kv_heads 8) into stacked engine storage."""

import json

import numpy as np
import pytest
import torch

from oracle import rl_loss as orl
from oracle import rl_loss_torch as orlt

from helpers import rel_err

pytestmark = pytest.mark.gpu

FP_TOL = 1e-4
H32, V32 = 5120, 152064

# the loss settings of configs[4]: conf/finetune/grpo.yaml over base.yaml:100-114, KL on (deepscaler15b.yaml:34)
CFG5 = dict(policy_loss="ppo", epsilon_low=0.02, epsilon_high=0.02, kl_coef=0.001, final_kl_coef=0.001, temperature=1.0,
            batch_size=4096, clamp_log_ratio_ref_new_value=5, divide_advantage_by_std=False, group_normalization=False)


def _problem(T, H, V, dev, cfg, seed, weight_dtype=torch.float32, n_seqs=2):
    """hidden bf16 [1, T, H], W [V, H], a packed batch of `n_seqs` sequences with prompts; old log-probs sit within the PPO
    clip range of the true ones for most tokens, the reference log-probs are a DIFFERENT column (ref = old + N(0, 0.05))."""
    g = torch.Generator(device=dev).manual_seed(seed)
    hidden = torch.randn((1, T, H), generator=g, device=dev).to(torch.bfloat16)
    W = (torch.randn((V, H), generator=g, device=dev) * (2.0 / H ** 0.5)).to(weight_dtype)
    rng = np.random.default_rng(seed + 1)
    ids = rng.integers(3, V, size=(1, T), dtype=np.int64)
    bounds = np.linspace(0, T, n_seqs + 1).astype(int)
    pos = np.concatenate([np.arange(b - a) for a, b in zip(bounds[:-1], bounds[1:])])[None].astype(np.int64)
    labels = ids.copy()
    for a, b in zip(bounds[:-1], bounds[1:]):
        labels[0, a:a + max(2, (b - a) // 8)] = -100  # prompt
    labels[0, rng.random(T) < 0.03] = -100            # observation tokens inside completions
    nxt = torch.from_numpy(ids[0, 1:]).to(dev)
    parts = []
    Wd = W.double()
    for r0 in range(0, T - 1, 512):  # fp64 log-probs of the next tokens, in row chunks
        z = (hidden[0, r0:min(r0 + 512, T - 1)].double() @ Wd.t()) / cfg["temperature"]
        parts.append(z.gather(-1, nxt[r0:r0 + z.shape[0], None])[:, 0] - torch.logsumexp(z, -1))
    nlp = np.concatenate([[0.0], torch.cat(parts).cpu().numpy()])
    del Wd
    old = nlp + rng.normal(0, 0.01, T)
    f32 = lambda a: np.asarray(a, dtype=np.float32)[None]  # noqa: E731
    batch = {
        "input_ids": ids, "labels": labels, "position_ids": pos, "attention_mask": np.ones_like(ids),
        "old_logprobs": f32(old), "ref_logprobs": f32(old + rng.normal(0, 0.05, T)), "advantages": f32(rng.normal(0, 1, T)),
        "rewards": f32(rng.integers(0, 2, T)), "group_tokens": f32(np.full(T, 5000.0)),
        "num_labels": f32(np.full(T, float((labels != -100).sum()))), "overflow": f32(np.zeros(T)),
    }
    return hidden, W, batch, nlp


def _to_device(batch, dev):
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    return PipelineBatchEncoding(**{k: torch.from_numpy(v) for k, v in batch.items()}, model_version=0, is_packed=True).to_device(dev)


def _fp64_products(hidden, W, grad_logits):
    """d hidden = d logits @ W and d W = d logits^T @ hidden in fp64 on the device, row-chunked (`grad_logits`: host fp32)."""
    dev = hidden.device
    T = hidden.shape[1]
    Wd = W.double()
    hd = hidden[0].double()
    d_hidden = torch.empty((T, W.shape[1]), dtype=torch.float64, device=dev)
    d_weight = torch.zeros(W.shape, dtype=torch.float64, device=dev)
    for r0 in range(0, T, 1024):
        dl = torch.as_tensor(grad_logits[0, r0:r0 + 1024]).to(dev).double()
        d_hidden[r0:r0 + 1024] = dl @ Wd
        d_weight += dl.t() @ hd[r0:r0 + 1024]
    return d_hidden.cpu().numpy(), d_weight.cpu().numpy()


def _head_vs_oracle(dev, T, cfg, seed, oracle_step, keep_logits):
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.fused_head import FusedLmHead, fused_head_loss

    hidden, W, batch, _ = _problem(T, H32, V32, dev, cfg, seed)
    # the oracle is FED the fp64 product (rounded once to the fp32 the reference's fp32 head would hand it)
    logits = torch.empty((1, T, V32), dtype=torch.float32)
    Wd = W.double()
    for r0 in range(0, T, 512):
        logits[0, r0:r0 + 512] = (hidden[0, r0:r0 + 512].double() @ Wd.t()).float().cpu()
    del Wd
    want = oracle_step(logits.numpy(), batch, cfg, 2, 10, True)
    grad_logits = want["grad_logits"]
    grad_logits = grad_logits.numpy() if isinstance(grad_logits, torch.Tensor) else grad_logits
    want_dh, want_dw = _fp64_products(hidden, W, grad_logits)
    del logits, grad_logits
    pb = _to_device(batch, dev)
    h = hidden.clone().requires_grad_(True)
    w = W.clone().requires_grad_(True)
    head = FusedLmHead(w, keep_logits=keep_logits)
    loss, stats = fused_head_loss(h, w, head, pb, RLConfig(**cfg), 2, 10)
    loss.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - float(want["loss"])) <= FP_TOL * abs(float(want["loss"]))
    for k, v in want["stats"].items():
        assert abs(float(stats[k]) - float(v)) <= FP_TOL * max(abs(float(v)), 1.0), k
    assert float(stats["kl"]) > 0  # the KL term is live: ref != old
    assert rel_err(w.grad.float().cpu().numpy(), want_dw) <= FP_TOL
    assert rel_err(h.grad[0].float().cpu().numpy(), want_dh) <= 4e-3  # delivered in the hidden states' dtype (bf16)
    # fp32 d hidden straight from the C ABI: the 1e-4 bar without the bf16 rounding of the autograd path
    from pipelinerl_amd.finetune.rl import grpo_loss_from_logprobs, make_loss_config

    c_cfg, _, _ = make_loss_config(RLConfig(**cfg), 2, 10)
    nlp, ent, lse2, hb = head.logprob_entropy(hidden, pb.input_ids, cfg["temperature"])
    _, _, g_nlp, g_ent = grpo_loss_from_logprobs(c_cfg, pb, nlp, ent)
    gh32 = head.backward_from_token_grads(hb, pb.input_ids, cfg["temperature"], lse2, ent, g_nlp, g_ent, None, grad_hidden_dtype=torch.float32)
    assert rel_err(gh32[0].cpu().numpy(), want_dh) <= FP_TOL


@pytest.mark.parametrize("keep", [True, False], ids=["kept_logits", "recompute"])
def test_qwen32b_head_shape_vs_oracle(libprl, cuda_device, keep):
    """H = 5120 (80 contraction steps of 64; 20 hidden tiles of 256 in d W / d hidden), V = 152 064, fp32 weight as two
    bf16 planes, the KL-on loss of configs[4]: loss, all statistics, d hidden, d W."""
    _head_vs_oracle(cuda_device, 192, CFG5, seed=32, oracle_step=orl.rl_step, keep_logits=keep)


@pytest.mark.slow
def test_qwen32b_head_full_micro_batch_vs_oracle(libprl, cuda_device):
    """The whole 8192 x 5120 x 152 064 micro-batch of configs[4] (32 token tiles x 594 vocabulary tiles, one 8192-row
    backward chunk, split-K d hidden over 8 XCD slices) against the oracle's closed form on the host (torch CPU kernels)."""
    _head_vs_oracle(cuda_device, 8192, CFG5, seed=33, oracle_step=orlt.rl_step_closed_form, keep_logits=None)


@pytest.mark.parametrize("weight_dtype", [torch.float32, torch.bfloat16], ids=["fp32_head_two_planes", "bf16_head_one_plane"])
def test_reference_logprobs_through_the_mfma_head_vs_oracle(libprl, cuda_device, weight_dtype):
    """`annotate_ref_logprobs` on a model in the Hugging Face layout: the hidden states of the frozen reference policy go
    through `FusedLmHead(backward=False)` - only the rows that predict a labelled token, no `[T, V]` logits - and the
    column equals `oracle.rl_loss.logprob_entropy` of the fp64 product on labelled tokens, 0 elsewhere
    (preprocess.py:86-104 keeps the completion tokens' values, rl/__init__.py:573-594 left-pads with zeros)."""
    import types

    from pipelinerl_amd.finetune.rl import annotate_ref_logprobs

    T = 700
    hidden, W, batch, _ = _problem(T, H32, V32, cuda_device, CFG5, seed=34, weight_dtype=weight_dtype, n_seqs=3)
    logits64 = hidden[0].double() @ W.double().t()
    want_nlp = orl.logprob_entropy(logits64.float().cpu().numpy()[None], batch["input_ids"], 1.0)[0]  # [1, T - 1]
    want = np.zeros((1, T), dtype=np.float64)
    want[0, 1:] = want_nlp[0]
    want[batch["labels"] == -100] = 0.0

    calls = {"body": 0, "lm_head": 0}

    class Body(torch.nn.Module):
        def forward(self, input_ids=None, attention_mask=None, position_ids=None):
            calls["body"] += 1
            assert position_ids is not None  # packed batch
            return types.SimpleNamespace(last_hidden_state=hidden)

    class Head(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.weight = torch.nn.Parameter(W, requires_grad=False)
            self.bias = None

        def forward(self, x):
            calls["lm_head"] += 1
            raise AssertionError("the fused path must not run the library head")

    class RefLM(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.model, self.lm_head = Body(), Head()

        def forward(self, **kw):
            return types.SimpleNamespace(logits=(self.model(**kw).last_hidden_state.float() @ self.lm_head.weight.float().t()))

    ref = RefLM().eval()
    pb = _to_device(batch, cuda_device)
    from pipelinerl_amd.fused_head import ref_head_for

    ref_head_for(ref)[1].refresh()  # the weight's bf16 planes (3.1 GB for two) are built once per weight, not per micro-batch
    before = torch.cuda.max_memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    annotate_ref_logprobs(ref, pb, 1.0)
    torch.cuda.synchronize()
    peak = torch.cuda.max_memory_allocated() - base
    assert calls == {"body": 1, "lm_head": 0}
    assert peak < T * V32 * 2, f"{peak} bytes: a [T, V] tensor was materialised"  # even bf16 logits would be T * V * 2
    got = pb.ref_logprobs.double().cpu().numpy()
    assert got.shape == (1, T)
    assert np.count_nonzero(got[batch["labels"] == -100]) == 0
    np.testing.assert_allclose(got, want, rtol=FP_TOL, atol=2e-5 if weight_dtype == torch.float32 else 2e-5)
    # the logits path (fused_head=False: stock head + K1) fills the same column
    pb2 = _to_device(batch, cuda_device)
    annotate_ref_logprobs(ref, pb2, 1.0, fused_head=False)
    np.testing.assert_allclose(pb2.ref_logprobs.double().cpu().numpy(), want, rtol=FP_TOL, atol=2e-5)
    # the no-grad head holds the row-major planes only
    fused = ref.lm_head._prl_ref_lm_head
    assert fused.wt_hi is None and fused.wt_lo is None and (fused.w_lo is None) == (weight_dtype == torch.bfloat16)
    del before


def test_the_65gb_parameter_set_through_the_bucket_kernels(libprl, cuda_device):
    """All 771 tensors of the 32B set (65.5 GB, bf16) flattened bucket by bucket with `prl_bucket_gather` and landed in a
    second copy of the set with `prl_bucket_scatter` (1 GiB buckets; embed_tokens / lm_head are 1.557 GB buckets of their
    own; up to 13 segments per bucket): the two sets are equal byte for byte and the bucket padding is untouched."""
    from pipelinerl_amd.weight_sync import ParamSpec, bucket_nbytes, gather_into_bucket, plan_buckets, scatter_from_bucket
    from pipelinerl_amd.weight_sync_probe import qwen25_shapes

    dev = cuda_device
    free, _ = torch.cuda.mem_get_info()
    if free < 140e9:
        pytest.skip("needs 2 x 65.5 GB of device memory")
    shapes = qwen25_shapes("32b")
    specs = [ParamSpec(n, tuple(s), torch.bfloat16) for n, s in shapes]
    total = sum(sp.nbytes for sp in specs)
    src_flat = torch.empty(total, dtype=torch.uint8, device=dev)
    g = torch.Generator(device=dev).manual_seed(65)
    for a in range(0, total, 1 << 30):  # random bytes in 1 GiB pieces
        n = min(1 << 30, total - a)
        src_flat[a:a + n] = torch.randint(0, 256, (n,), dtype=torch.uint8, device=dev, generator=g)
    dst_flat = torch.zeros(total, dtype=torch.uint8, device=dev)
    src, dst, at = {}, {}, 0
    for sp in specs:
        src[sp.name] = src_flat[at:at + sp.nbytes].view(torch.bfloat16).view(sp.shape)
        dst[sp.name] = dst_flat[at:at + sp.nbytes].view(torch.bfloat16).view(sp.shape)
        at += sp.nbytes
    plan = plan_buckets(specs, 1 << 30)
    cap = max(bucket_nbytes(b) for b in plan)
    buf = torch.full((cap,), 0xA5, dtype=torch.uint8, device=dev)
    for bucket in plan:
        gather_into_bucket(buf, bucket, src)
        scatter_from_bucket(buf, bucket, dst)
    torch.cuda.synchronize()
    for a in range(0, total, 4 << 30):
        assert torch.equal(src_flat[a:a + (4 << 30)], dst_flat[a:a + (4 << 30)]), a
    # slot padding (256-byte alignment) of the last bucket was never written
    last = plan[-1]
    for (sp, off), (_, nxt) in zip(last, last[1:]):
        assert torch.all(buf[off + sp.nbytes:nxt] == 0xA5)


def test_tp2_sharded_update_at_the_32b_layer_shapes(libprl, cuda_device):
    """One full-size 32B layer + embedding + head (H 5120, I 27648, 40 heads, 8 KV heads of 128) cut for TP = 2 and landed
    in each rank's STACKED `qkv_proj` / `gate_up_proj` storage, byte for byte against vLLM-style slicing."""
    from test_tp_shard import LoopGroups, _vllm_style_rank_storage, qwen_shapes

    from pipelinerl_amd.finetune_loop import ParameterInfo, WeightUpdateRequest
    from pipelinerl_amd.tp_shard import plan_tp_shards
    from pipelinerl_amd.vllm_worker import StackedShardReceiver
    from pipelinerl_amd.weight_sync import ShardedSender

    tp, heads, kv_heads, head_dim = 2, 40, 8, 128
    shapes = qwen_shapes(layers=1, hidden=H32, inter=27648, heads=heads, kv_heads=kv_heads, head_dim=head_dim, vocab=V32)
    g = torch.Generator(device=cuda_device).manual_seed(41)
    full = [(n, torch.randn(s, device=cuda_device, generator=g).to(torch.bfloat16)) for n, s in shapes]
    cuts = plan_tp_shards(shapes, tp, kv_heads=kv_heads)
    loops = LoopGroups(tp)
    for grp in loops.groups:
        grp.device = cuda_device
    bucket = 1 << 28
    sender = ShardedSender(loops.groups, bucket_bytes=bucket)
    sender.send(full, cuts)
    torch.cuda.synchronize()
    req = WeightUpdateRequest(version=0, transport="sharded", bucket_bytes=bucket, tp_size=tp,
                              parameters_info=[ParameterInfo(name=n, shape=list(s), dtype="torch.bfloat16", shard_dim=cuts[n].dim,
                                                             shard_parts=cuts[n].parts) for n, s in shapes])
    total = sum(x.numel() * 2 for _, x in full)
    assert all(b < 0.51 * total for b in sender.bytes_sent)
    for t in range(tp):
        w = StackedShardReceiver(shapes, lambda n: torch.bfloat16, cuda_device, t, tp, kv_heads=kv_heads)
        for x in w.storage.values():
            x.fill_(7.0)
        r = loops.groups[t].reader(t)
        r.device = cuda_device
        w.model_update_group, w.tp_rank, w.tp_size = r, t, tp
        w.receive_weight_update(json.dumps(req.model_dump()))
        torch.cuda.synchronize()
        want = _vllm_style_rank_storage(dict(full), t, tp, heads, kv_heads, head_dim)
        assert set(want) == set(w.storage)
        for name, x in want.items():
            assert torch.equal(w.storage[name], x), (t, name)
