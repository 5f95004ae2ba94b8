"""SplitBf16LmHead: an fp32 output head evaluated as bf16 MFMA GEMMs with fp32 accumulation must agree
with the fp32 GEMM it replaces (reference numerics: checkpoints.py:87-103 keep the head in fp32)."""

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_split_bf16_head_matches_fp32_linear(cuda_device):
    from pipelinerl_amd.lm_head import SplitBf16LmHead, split_bf16

    torch.manual_seed(0)
    T, H, V = 384, 256, 1536
    x = torch.randn(2, T // 2, H, device=cuda_device).to(torch.bfloat16)
    w = torch.randn(V, H, device=cuda_device) * 0.05
    parts = split_bf16(w, 2)
    assert (parts[0].float() + parts[1].float() - w).abs().max() <= 2.0 ** -15 * w.abs().max()

    head = SplitBf16LmHead(torch.nn.Parameter(w.clone()))
    xa = x.clone().requires_grad_(True)
    logits = head(xa)
    assert logits.dtype == torch.float32 and logits.shape == (2, T // 2, V)
    g = torch.randn_like(logits)
    logits.backward(g)

    w64 = w.double().requires_grad_(True)
    x64 = x.double().requires_grad_(True)
    ref = x64 @ w64.t()
    ref.backward(g.double())
    fp32 = x.float() @ w.t()
    scale = ref.abs().max().item()
    err_split = (logits.double() - ref).abs().max().item()
    err_fp32 = (fp32.double() - ref).abs().max().item()
    assert err_split <= 4e-6 * scale and err_split <= 4 * err_fp32 + 1e-7  # fp32-GEMM class accuracy
    assert (head.weight.grad.double() - w64.grad).abs().max() <= 2e-4 * w64.grad.abs().max()
    # d hidden is returned in bf16 (the dtype of the hidden states): bf16 rounding of the fp64 result
    assert (xa.grad.double() - x64.grad).abs().max() <= 2.0 ** -7 * x64.grad.abs().max()

    # the split follows in-place parameter updates
    with torch.no_grad():
        head.weight.add_(0.01)
    again = head(x)
    assert (again.double() - x.double() @ (w.double() + 0.01).t()).abs().max() <= 4e-6 * scale + 1e-5


def test_split_head_feeds_the_loss_kernel(libprl, cuda_device):
    """hidden -> SplitBf16LmHead -> rl_step's fused loss kernel -> gradients reach hidden and the head."""
    import types

    from pipelinerl_amd.finetune.rl import RLConfig, rl_step
    from pipelinerl_amd.hotpath import HotPathStep, dense_micro_batches
    from pipelinerl_amd.lm_head import SplitBf16LmHead
    from pipelinerl_amd.synthetic import make_ragged

    V, H = 640, 64
    rag_h, _ = make_ragged(2, attempts=2, seq_length=32, vocab=V, seed=5, prompt_min=3, prompt_max=6)
    rag = rag_h.to(cuda_device)
    cfg = RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.0, final_kl_coef=0.0, batch_size=4, divide_advantage_by_std=False)
    batches = HotPathStep(cfg, 2, 0, 10).preprocess(rag, dense_micro_batches(rag_h, 200))
    torch.manual_seed(3)
    emb = torch.nn.Embedding(V, H).to(cuda_device).to(torch.bfloat16)
    w = torch.nn.Parameter(torch.randn(V, H, device=cuda_device) * 0.1)

    def run(head_fn):
        for p in (emb.weight, w):
            p.grad = None
        model = lambda **kw: types.SimpleNamespace(logits=head_fn(emb(kw["input_ids"])))  # noqa: E731
        loss, stats = rl_step(model, batches[0], 0, 10, cfg)
        loss.backward()
        return loss.item(), emb.weight.grad.float().clone(), w.grad.clone()

    head = SplitBf16LmHead(w)
    l_split, ge_split, gw_split = run(head)
    l_fp32, ge_fp32, gw_fp32 = run(lambda h: h.float() @ w.t())
    assert abs(l_split - l_fp32) <= 1e-5 * max(1.0, abs(l_fp32))
    assert (gw_split - gw_fp32).abs().max() <= 1e-3 * gw_fp32.abs().max() + 1e-9
    assert (ge_split - ge_fp32).abs().max() <= 2.0 ** -6 * ge_fp32.abs().max() + 1e-9


@pytest.mark.parametrize("n", [4096, 4099, 1 << 22, 3])
def test_split_bf16_kernel_matches_torch(libprl, cuda_device, n):
    """prl_split_bf16 (one pass) == the three-pass torch formulation, bit for bit."""
    from pipelinerl_amd.lm_head import split_bf16

    torch.manual_seed(n)
    t = torch.randn(n + 4, device=cuda_device)[:n] * torch.logspace(-3, 3, n, device=cuda_device)
    t = t.contiguous()
    if n >= 8:
        t[:8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.0e-30, 65504.0], device=cuda_device)
    hi, lo = split_bf16(t, 2)
    want_hi = t.to(torch.bfloat16)
    want_lo = (t - want_hi.float()).to(torch.bfloat16)
    assert torch.equal(hi.view(torch.int16), want_hi.view(torch.int16))
    assert torch.equal(lo.view(torch.int16), want_lo.view(torch.int16))
    finite = torch.isfinite(hi.float() + lo.float())
    assert ((hi.float() + lo.float() - t).abs()[finite] <= 2.0 ** -15 * t.abs()[finite] + 1e-38).all()


@pytest.mark.parametrize("tied", [False, True])
def test_apply_fp32_lm_head_drop_in(cuda_device, tied):
    """`apply_fp32_lm_head(model)` (reference checkpoints.py:44-103): fp32-precision logits from the
    model's own lm_head module - fp32 weight (2-term split) or a bf16 weight tied to the embedding (exact)."""
    from pipelinerl_amd.lm_head import apply_fp32_lm_head

    class Tiny(torch.nn.Module):
        def __init__(self, V=768, H=128):
            super().__init__()
            self.embed = torch.nn.Embedding(V, H)
            self.lm_head = torch.nn.Linear(H, V, bias=False)

        def forward(self, ids):
            return self.lm_head(self.embed(ids))

    torch.manual_seed(4)
    m = Tiny().to(cuda_device).to(torch.bfloat16)
    if tied:
        m.lm_head.weight = m.embed.weight
    else:
        m.lm_head = m.lm_head.float()
    apply_fp32_lm_head(m)
    ids = torch.randint(0, 768, (2, 96), device=cuda_device)
    logits = m(ids)
    assert logits.dtype == torch.float32
    h = m.embed(ids)
    want = h.double() @ m.lm_head.weight.double().t()
    assert (logits.double() - want).abs().max() <= 4e-6 * want.abs().max()
    g = torch.randn_like(logits)
    logits.backward(g)
    w = m.lm_head.weight
    assert w.grad is not None and w.grad.dtype == w.dtype
    if not tied:
        want_dw = g.reshape(-1, 768).double().t() @ h.reshape(-1, 128).double()
        assert (w.grad.double() - want_dw).abs().max() <= 2e-4 * want_dw.abs().max()
    assert m.lm_head.weight.data_ptr() == (m.embed.weight.data_ptr() if tied else m.lm_head.weight.data_ptr())
