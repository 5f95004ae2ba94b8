"""d logits delivered as bf16 (hi, lo) planes straight from the fused logits pass
(`prl_fused_logits_loss_planes`, SURVEY §8f-1 library-GEMM variant): bit for bit what `prl_split_bf16` makes
of `prl_fused_logits_loss`'s fp32 gradient - at the BASELINE vocabulary (row-resident kernel) and at small /
unaligned sizes (two-sweep kernel) - plus the oracle on top, and `rl_step_split_head`, the `rl_step` drop-in
that consumes the planes with library GEMMs."""

import ctypes

import numpy as np
import pytest
import torch

from helpers import rel_err
from test_gpu_fullvocab import CONFIGS, FP_TOL, _case, _launch_fused

pytestmark = pytest.mark.gpu


def _launch_planes(lib, dev, logits_t, batch, cfg_name, plane_stride=None):
    from pipelinerl_amd import _lib
    from pipelinerl_amd.finetune.rl import RLConfig, make_loss_config

    c_cfg, _, _ = make_loss_config(RLConfig(**CONFIGS[cfg_name]), 2, 10)
    T, V = logits_t.shape[1], logits_t.shape[2]
    ps = plane_stride or V
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in batch.items()}
    nlp, ent, lse = (torch.full((1, T), 7.0, device=dev) for _ in range(3))
    hi = torch.full((T, ps), 3.0, dtype=torch.bfloat16, device=dev)
    lo = torch.full((T, ps), 3.0, dtype=torch.bfloat16, device=dev)
    rc = lib.prl_fused_logits_loss_planes(
        ctypes.byref(c_cfg), 1, T, V, logits_t.data_ptr(), V, CONFIGS[cfg_name]["temperature"],
        d["input_ids"].data_ptr(), d["labels"].data_ptr(), d["old_logprobs"].data_ptr(), d["ref_logprobs"].data_ptr(),
        d["advantages"].data_ptr(), d["rewards"].data_ptr(), d["group_tokens"].data_ptr(), d["overflow"].data_ptr(),
        nlp.data_ptr(), ent.data_ptr(), lse.data_ptr(), hi.data_ptr(), lo.data_ptr(), ps, _lib.current_stream_ptr(dev))
    _lib.check(rc)
    torch.cuda.synchronize()
    return nlp, ent, hi, lo, lib.prl_last_fused_kernel().decode()


@pytest.mark.parametrize("V,T,kernel", [(152064, 40, "keep_kernel"), (151936, 33, "keep_kernel"), (1000, 50, "fused_logits_loss_kernel"),
                                        (1003, 50, "fused_logits_loss_kernel")])
@pytest.mark.parametrize("cfg_name", ["kl_ent_temp", "grpo_clip"])
def test_planes_are_the_split_of_the_fp32_gradient(libprl, cuda_device, V, T, kernel, cfg_name):
    from pipelinerl_amd import _lib

    logits, batch, want, g64 = _case(V, T, cfg_name, "f32")
    lt = torch.from_numpy(logits).to(cuda_device)
    nlp0, ent0, grad, _ = _launch_fused(libprl, cuda_device, lt, batch, cfg_name, inplace=False)
    hi0, lo0 = (torch.empty((T, V), dtype=torch.bfloat16, device=cuda_device) for _ in range(2))
    _lib.check(libprl.prl_split_bf16(T * V, grad.data_ptr(), hi0.data_ptr(), lo0.data_ptr(), _lib.current_stream_ptr(cuda_device)))
    before = lt.clone()
    nlp, ent, hi, lo, name = _launch_planes(libprl, cuda_device, lt, batch, cfg_name)
    assert kernel in name and "planes" in name, name
    assert torch.equal(lt, before)  # the logits are read-only here
    assert torch.equal(nlp, nlp0) and torch.equal(ent, ent0)
    assert torch.equal(hi.view(torch.int16), hi0.view(torch.int16))
    assert torch.equal(lo.view(torch.int16), lo0.view(torch.int16))
    # and the planes' sum against the fp64 closed form directly
    got = (hi.float() + lo.float()).cpu().numpy()
    assert rel_err(got, g64) <= FP_TOL
    masked = np.where(np.asarray(batch["labels"])[0, 1:] == -100)[0]
    assert not got[masked].any() and not got[-1].any()


def test_plane_stride_alignment_and_aliasing(libprl, cuda_device):
    from pipelinerl_amd import _lib
    from pipelinerl_amd.finetune.rl import RLConfig, make_loss_config

    V, T = 1000, 20
    logits, batch, want, g64 = _case(V, T, "kl_ent_temp", "f32")
    lt = torch.from_numpy(logits).to(cuda_device)
    _, _, hi, lo, _ = _launch_planes(libprl, cuda_device, lt, batch, "kl_ent_temp", plane_stride=V + 24)
    got = (hi[:, :V].float() + lo[:, :V].float()).cpu().numpy()
    assert rel_err(got, g64) <= FP_TOL
    assert (hi[:, V:].float() == 3.0).all() and (lo[:, V:].float() == 3.0).all()  # the padding columns are not touched
    # planes inside the logits buffer are refused (rows are written while later rows are still read)
    c_cfg, _, _ = make_loss_config(RLConfig(**CONFIGS["kl_ent_temp"]), 2, 10)
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(cuda_device) for k, v in batch.items()}
    f = torch.empty(1, T, device=cuda_device)
    rc = libprl.prl_fused_logits_loss_planes(
        ctypes.byref(c_cfg), 1, T, V, lt.data_ptr(), V, 0.7, d["input_ids"].data_ptr(), d["labels"].data_ptr(), d["old_logprobs"].data_ptr(),
        d["ref_logprobs"].data_ptr(), d["advantages"].data_ptr(), d["rewards"].data_ptr(), d["group_tokens"].data_ptr(), d["overflow"].data_ptr(),
        f.data_ptr(), f.data_ptr(), f.data_ptr(), lt.data_ptr(), hi.data_ptr(), V, _lib.current_stream_ptr(cuda_device))
    assert rc == _lib.PRL_EINVAL
    rc = libprl.prl_fused_logits_loss_planes(
        ctypes.byref(c_cfg), 1, T, V, lt.data_ptr(), V, 0.7, d["input_ids"].data_ptr(), d["labels"].data_ptr(), d["old_logprobs"].data_ptr(),
        d["ref_logprobs"].data_ptr(), d["advantages"].data_ptr(), d["rewards"].data_ptr(), d["group_tokens"].data_ptr(), d["overflow"].data_ptr(),
        f.data_ptr(), f.data_ptr(), f.data_ptr(), hi.data_ptr(), lo.data_ptr(), V - 8, _lib.current_stream_ptr(cuda_device))
    assert rc == _lib.PRL_EINVAL  # plane stride shorter than a row


def test_rl_step_split_head_matches_rl_step_and_the_oracle(libprl, cuda_device):
    """Body + fp32 head: `rl_step_split_head` (library GEMMs around the plane-emitting pass) against `rl_step` on the
    model's fp32 logits (autograd through an fp32 matmul) and against the oracle fed with the fp64 logits."""
    import copy
    import types

    from oracle import rl_loss as orl
    from pipelinerl_amd.finetune.rl import RLConfig, rl_step
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.lm_head import rl_step_split_head
    from test_gpu_lmhead_fused import CFG, _problem

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
    (la * 0.25).backward()
    lb, sb = rl_step_split_head(b, pb, 2, 10, cfg)
    (lb * 0.25).backward()  # an upstream factor: applied to d hidden / d W, not to a [T, V] tensor
    assert abs(la.item() - lb.item()) <= FP_TOL * abs(la.item())
    assert list(sa) == list(sb)
    for k in sa:
        assert abs(sa[k] - sb[k]) <= FP_TOL * max(abs(sa[k]), 1.0), k
    for (n, pa), (_, pbb) in zip(a.named_parameters(), b.named_parameters()):
        assert rel_err(pbb.grad.cpu().numpy(), pa.grad.cpu().numpy()) <= 2e-2, n  # the body sees a bf16 d hidden in both
    assert rel_err(b.lm_head.weight.grad.cpu().numpy(), a.lm_head.weight.grad.cpu().numpy()) <= 1e-4
    # the oracle on the fp64 logits of the same hidden states
    with torch.no_grad():
        h = b.model(input_ids=pb.input_ids)[0]
        logits64 = h[0].double() @ b.lm_head.weight.double().t()
    want = orl.rl_step(logits64.float().cpu().numpy()[None], batch, CFG, 2, 10, True)
    assert abs(lb.item() - float(want["loss"])) <= FP_TOL * abs(float(want["loss"]))
    dl = torch.from_numpy(want["grad_logits"][0]).to(cuda_device).double()
    assert rel_err(b.lm_head.weight.grad.cpu().numpy(), 0.25 * (dl.t() @ h[0].double()).cpu().numpy()) <= FP_TOL
    # inference (no graph): the loss without a gradient pass
    with torch.no_grad():
        l0, _ = rl_step_split_head(b, pb, 2, 10, cfg)
    assert abs(l0.item() - lb.item()) <= 1e-6 * abs(lb.item())
