"""An fp32 lm_head evaluated on the bf16 matrix cores (first step).

The reference keeps the output projection in fp32 (pipelinerl/finetune/checkpoints.py:87-103, vLLM side
vllm_quantization.py:240-278) while the hidden states arriving at it are bf16.  On MI355X an fp32 GEMM
runs at ~150 TFLOP/s and a bf16 MFMA GEMM at ~1450 TFLOP/s, so the three lm_head GEMMs of a 7B micro-batch
(8192 x 3584 x 152064) cost ~176 ms in fp32 next to 1.8 ms for this package's loss kernel.

bf16 products are exact in fp32 (8 + 8 mantissa bits), so splitting the fp32 operand into a sum of bf16 terms
and accumulating the partial GEMMs in fp32 reproduces the fp32 result to the accuracy of the split:

    W = W_hi + W_lo (+ ...),  W_hi = bf16(W),  W_lo = bf16(W - W_hi)
    x @ W^T  =  x @ W_hi^T + x @ W_lo^T                      (x is bf16, exact)

Measured on one MI355X (profiles/r01x_split_bf16_lm_head_probe.txt): two terms are as close to the fp64
product as the fp32 GEMM itself (1.7e-5 vs 2.1e-5 max abs error at |logits| <= 7.8) at 3.5x its speed.
The GEMMs are plain hipBLASLt library calls (`torch.mm(..., out_dtype=float32)`); what this module adds is
the operand splitting and the backward that keeps every GEMM on the bf16 cores.
"""

from __future__ import annotations

import torch


def split_bf16(t: torch.Tensor, terms: int = 2) -> list[torch.Tensor]:
    """fp32 tensor -> `terms` bf16 tensors whose fp32 sum approximates it to ~2^(-8 terms) relative.
    Two terms of a contiguous fp32 device tensor come out of ONE pass of the `prl_split_bf16` kernel."""
    if terms == 2 and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.data_ptr() % 16 == 0:
        from . import _lib

        if t.numel() % 4 == 0:  # hi plane, then lo plane, in one buffer (the lo plane stays 8-byte aligned)
            planes = torch.empty((2,) + tuple(t.shape), dtype=torch.bfloat16, device=t.device)
            hi, lo = planes[0], planes[1]
        else:
            hi, lo = torch.empty_like(t, dtype=torch.bfloat16), torch.empty_like(t, dtype=torch.bfloat16)
        with torch.cuda.device(t.device):
            _lib.check(_lib.load().prl_split_bf16(t.numel(), t.data_ptr(), hi.data_ptr(), lo.data_ptr(), _lib.current_stream_ptr(t.device)))
        return [hi, lo]
    parts, rest = [], t.float()
    for k in range(terms):
        p = rest.to(torch.bfloat16)
        parts.append(p)
        if k + 1 < terms:
            rest = rest - p.float()
    return parts


_ADDMM_OUT_DTYPE: bool | None = None  # does torch.addmm(fp32, bf16, bf16, out_dtype=fp32) work on this build?


def _mm_acc(a_parts, b_parts, pairs) -> torch.Tensor:
    """sum over (i, j) in pairs of a_parts[i] @ b_parts[j], accumulated in fp32 - inside the GEMM
    epilogue (beta = 1) where the library supports it, otherwise with a separate add pass."""
    global _ADDMM_OUT_DTYPE
    acc = None
    for i, j in pairs:
        if acc is None:
            acc = torch.mm(a_parts[i], b_parts[j], out_dtype=torch.float32)
            continue
        if _ADDMM_OUT_DTYPE is not False:
            try:
                acc = torch.addmm(acc, a_parts[i], b_parts[j], out_dtype=torch.float32)
                _ADDMM_OUT_DTYPE = True
                continue
            except (RuntimeError, TypeError):
                if _ADDMM_OUT_DTYPE:  # it worked before: a real error
                    raise
                _ADDMM_OUT_DTYPE = False
        acc.add_(torch.mm(a_parts[i], b_parts[j], out_dtype=torch.float32))
    return acc


class _SplitBf16Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: torch.Tensor, weight: torch.Tensor, w_parts: tuple[torch.Tensor, ...], w_cat: torch.Tensor | None, dx_terms: int = 3):  # type: ignore[override]
        if x.dtype != torch.bfloat16:
            raise TypeError("split_bf16_linear expects bf16 hidden states (they are exact bf16 operands)")
        x2 = x.reshape(-1, x.shape[-1])
        if w_cat is not None:
            # sum_k x @ W_k^T as ONE GEMM over the concatenated inner dimension: [x | x | ..] @ [W_0 | W_1 | ..]^T.
            # The partial products meet in the MFMA accumulators instead of a 5 GB read-modify-write pass.
            out = torch.mm(torch.cat([x2] * len(w_parts), dim=1), w_cat.t(), out_dtype=torch.float32)
        else:
            out = _mm_acc([x2], [p.t() for p in w_parts], [(0, j) for j in range(len(w_parts))])
        ctx.save_for_backward(x2, *w_parts)
        ctx.x_shape = x.shape
        ctx.dx_terms = dx_terms
        ctx.needs_w = weight.requires_grad
        return out.reshape(*x.shape[:-1], w_parts[0].shape[0])

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):  # type: ignore[override]
        x2, *w_parts = ctx.saved_tensors
        g = grad_out.reshape(-1, grad_out.shape[-1])
        g_parts = split_bf16(g, 2) if g.dtype == torch.float32 else [g.to(torch.bfloat16)]
        n_g, n_w = len(g_parts), len(w_parts)
        # d x = G W: keep the terms down to the second order of the splits (hi*hi, hi*lo, lo*hi)
        pairs = [(i, j) for i in range(n_g) for j in range(n_w) if i + j < max(n_g, n_w)][: ctx.dx_terms]
        dx = _mm_acc(g_parts, list(w_parts), pairs).to(torch.bfloat16).reshape(ctx.x_shape)
        dw = None
        if ctx.needs_w:  # d W = G^T x, x exact (returned in fp32; autograd casts it to a bf16 parameter's dtype)
            stacked = (n_g == 2 and g_parts[0].is_contiguous() and g_parts[1].is_contiguous()
                       and g_parts[1].data_ptr() == g_parts[0].data_ptr() + g_parts[0].numel() * 2)
            if stacked:  # the two planes are one [2T, V] buffer: G_hi^T x + G_lo^T x as ONE GEMM over 2T
                g_cat = torch.as_strided(g_parts[0], (2 * g.shape[0], g.shape[1]), (g.shape[1], 1))
                dw = torch.mm(g_cat.t(), torch.cat([x2, x2], dim=0), out_dtype=torch.float32)
            else:
                dw = _mm_acc([p.t() for p in g_parts], [x2], [(i, 0) for i in range(n_g)])
        return dx, dw, None, None, None


def _weight_key(w: torch.Tensor) -> tuple:
    """What the cached split is valid for: in-place autograd-visible updates bump `_version`;
    `p.data = other` (ColocatedSender.rehome, FSDP unshard) changes `data_ptr`; `.to(device)` the
    device.  Writes through `p.data.copy_()` change none of them - see `invalidate()`."""
    return (w._version, w.data_ptr(), w.device, tuple(w.shape))


class SplitBf16LmHead(torch.nn.Module):
    """Drop-in for an fp32 `nn.Linear(hidden, vocab, bias=False)` output head: `weight` stays an fp32
    parameter (optimizer, checkpoints and the weight broadcast see fp32), the forward / backward GEMMs run
    as bf16 MFMA GEMMs with fp32 accumulation.  The bf16 split of the weight is refreshed whenever the
    parameter changes: in-place optimizer steps bump its version counter, a re-homed storage changes
    its address; writers that go through `.data.copy_()` must call `invalidate()` (or use
    `attach_optimizer`)."""

    def __init__(self, weight: torch.Tensor, terms: int = 2, hidden_grad_terms: int = 3):
        """`hidden_grad_terms`: how many of the partial products G_hi W_hi, G_hi W_lo, G_lo W_hi enter d hidden.
        3 reproduces the fp32 product before it is rounded to the bf16 hidden dtype; 1 keeps only the leading
        term, whose error (2^-9 relative) is of the order of that rounding - 2 GEMMs cheaper."""
        super().__init__()
        self.hidden_grad_terms = hidden_grad_terms
        self.weight = weight if isinstance(weight, torch.nn.Parameter) else torch.nn.Parameter(weight.float())
        self.terms = terms
        self._parts: tuple[torch.Tensor, ...] | None = None
        self._cat: torch.Tensor | None = None
        self._parts_key = None

    @classmethod
    def from_linear(cls, linear: torch.nn.Linear, terms: int = 2, hidden_grad_terms: int = 3) -> "SplitBf16LmHead":
        if linear.bias is not None:
            raise ValueError("lm_head with a bias is not supported")
        return cls(linear.weight if linear.weight.dtype == torch.float32 else torch.nn.Parameter(linear.weight.float()), terms, hidden_grad_terms)

    def invalidate(self) -> None:
        """Drop the cached bf16 split.  REQUIRED after the weight was changed through a path that
        neither bumps the parameter's version counter nor moves its storage: `p.data.copy_()` (ZeRO /
        DeepSpeed, many checkpoint loaders), FSDP reshard into the same storage."""
        self._parts_key = None

    def attach_optimizer(self, optimizer: torch.optim.Optimizer) -> None:
        """Re-split after every `optimizer.step()` whatever the optimizer does to the storage."""
        optimizer.register_step_post_hook(lambda *_: self.invalidate())

    def _split(self) -> tuple[torch.Tensor, ...]:
        key = _weight_key(self.weight)
        if self._parts is None or self._parts_key != key:
            with torch.no_grad():
                parts = split_bf16(self.weight.detach(), self.terms)
                # [V, terms * H]: the terms side by side along the inner dimension (forward), and the
                # same storage viewed per term (backward) - once per optimizer step
                self._cat = torch.cat(parts, dim=1)
                h = self.weight.shape[1]
                self._parts = tuple(self._cat[:, k * h : (k + 1) * h] for k in range(self.terms))
            self._parts_key = key
        return self._parts

    def forward(self, hidden: torch.Tensor) -> torch.Tensor:
        parts = self._split()
        return _SplitBf16Linear.apply(hidden.to(torch.bfloat16) if hidden.dtype != torch.bfloat16 else hidden, self.weight, parts, self._cat, self.hidden_grad_terms)


def apply_fp32_lm_head(model: torch.nn.Module, layer_prefix: str = "lm_head", hidden_grad_terms: int = 3) -> torch.nn.Module:
    """Drop-in for the reference's `apply_fp32_lm_head(model, layer_prefix)` (finetune/checkpoints.py:44-103):
    the output projection computes in fp32 precision whatever the dtype of its inputs - here on the bf16
    matrix cores.  As in the reference the module and its parameter stay where they are (a tied weight keeps
    its storage and dtype); only `forward` is replaced.

      * fp32 weight (untied head upcast by the trainer): 2-term bf16 split, refreshed when the weight changes;
      * bf16 weight (tied to the embedding): the weight is already an exact bf16 operand, ONE GEMM;
      * a bias is added in fp32.
    """
    head = model
    for part in layer_prefix.split("."):
        head = getattr(head, part)
    if not isinstance(head, torch.nn.Linear):
        raise TypeError(f"{layer_prefix} is {type(head).__name__}, expected nn.Linear")
    state = {"key": None, "parts": None, "cat": None}

    def operands():
        w = head.weight
        key = _weight_key(w)
        if state["key"] != key:
            with torch.no_grad():
                if w.dtype == torch.bfloat16:
                    state["parts"], state["cat"] = (w.detach(),), None
                else:
                    parts = split_bf16(w.detach().float(), 2)
                    cat = torch.cat(parts, dim=1)
                    hdim = w.shape[1]
                    state["parts"], state["cat"] = tuple(cat[:, k * hdim : (k + 1) * hdim] for k in range(2)), cat
            state["key"] = key
        return state["parts"], state["cat"]

    def fp32_forward(x: torch.Tensor) -> torch.Tensor:
        parts, cat = operands()
        y = _SplitBf16Linear.apply(x if x.dtype == torch.bfloat16 else x.to(torch.bfloat16), head.weight, parts, cat, hidden_grad_terms)
        if head.bias is not None:
            y = y + head.bias.float()
        return y

    head.forward = fp32_forward
    # same contract as SplitBf16LmHead.invalidate(): call after `.data.copy_()`-style updates
    head.invalidate_split = lambda: state.update(key=None)
    return model
