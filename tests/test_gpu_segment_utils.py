"""The reference's segment-reduction helpers (pipelinerl/finetune/rl/utils.py:26-92, 106-208) under their own names in
`pipelinerl_amd.finetune.rl.utils`, on the GPU, against values AND autograd gradients the reference itself produced
(tests/golden/make_segment_utils_golden.py imports /root/reference and writes tests/golden/segment_utils.npz)."""

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu

Z = np.load(GOLDEN / "segment_utils.npz")
CASES = sorted({k.split("/")[0] for k in Z.files})


@pytest.mark.parametrize("name", CASES)
def test_segment_helpers_match_the_reference(libprl, cuda_device, name):
    from pipelinerl_amd.finetune.rl import utils as u

    g = lambda k: Z[f"{name}/{k}"]  # noqa: E731
    dev = cuda_device
    seg = torch.from_numpy(g("segment_ids")).to(dev)
    mask = torch.from_numpy(g("mask")).to(dev)
    a = torch.tensor(g("a"), device=dev, requires_grad=True)
    b = torch.tensor(g("b"), device=dev, requires_grad=True)
    ups = torch.from_numpy(g("upstream")).to(dev)
    lrn, adv, cnt = u.per_segment_sums(seg, mask, a, b)
    np.testing.assert_allclose(lrn.detach().cpu().numpy(), g("lrn_sum"), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(adv.detach().cpu().numpy(), g("adv_sum"), rtol=1e-5, atol=1e-6)
    np.testing.assert_array_equal(cnt.cpu().numpy(), g("count"))
    ((lrn * ups[0]).sum() + (adv * ups[1]).sum()).backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), g("grad_a"), rtol=1e-6, atol=0)
    np.testing.assert_allclose(b.grad.cpu().numpy(), g("grad_b"), rtol=1e-6, atol=0)
    bounds = g("bounds").tolist()
    segments = list(zip(bounds[:-1], bounds[1:]))
    for fn, key in ((u.sum_sum, "sum_sum"), (u.mean_sum, "mean_sum")):
        v = torch.tensor(g("a"), device=dev, requires_grad=True)
        out = fn(v, mask, segments)
        assert out.item() == pytest.approx(float(g(key)), rel=2e-5, abs=1e-6)
        out.backward()
        np.testing.assert_allclose(v.grad.cpu().numpy(), g(f"{key}_grad"), rtol=1e-5, atol=1e-7)
        assert fn(torch.from_numpy(g("a")).to(dev), mask, None).item() == pytest.approx(float(g(f"{key}_unpacked")), rel=2e-5, abs=1e-6)
    va = torch.from_numpy(g("a")).to(dev)
    assert u.mask_sum(va, mask).item() == pytest.approx(float(g("mask_sum")), rel=2e-5, abs=1e-6)
    assert u.mask_mean(va, mask).item() == pytest.approx(float(g("mask_mean")), rel=2e-5, abs=1e-6)


def test_segment_helpers_refuse_bad_arguments(libprl, cuda_device):
    from pipelinerl_amd.finetune.rl import utils as u

    x = torch.zeros(1, 7, device=cuda_device)
    m = torch.ones(1, 7, dtype=torch.bool, device=cuda_device)
    with pytest.raises(ValueError, match="segment_ids must be provided"):
        u.per_segment_sums(None, m, x, x)
    with pytest.raises(ValueError, match=r"\[1, L\]"):
        u.per_segment_sums(torch.zeros(8, dtype=torch.int64, device=cuda_device), m, x, x)
    with pytest.raises(ValueError, match="ascending"):
        u.sum_sum(x, m, [(4, 7), (0, 4)])
    with pytest.raises(RuntimeError, match="HIP device"):
        u.per_segment_sums(torch.zeros(1, 8, dtype=torch.int64), m.cpu(), x.cpu(), x.cpu())
    # a NaN under the mask is ignored, a NaN at a valid position counts as 0 (the reference's nan_to_num)
    v = x.clone()
    v[0, 2], v[0, 5] = float("nan"), 3.0
    m2 = m.clone()
    m2[0, 2] = False
    assert u.sum_sum(v, m2, [(0, 4), (4, 7)]).item() == 3.0
    assert u.sum_sum(v, m, [(0, 4), (4, 7)]).item() == 3.0
