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
        # with w_cat: sum_k x @ W_k^T as ONE GEMM over the concatenated inner dimension, [x | x | ..] @ [W_0 | W_1 | ..]^T -
        # the partial products meet in the MFMA accumulators instead of a 5 GB read-modify-write pass
        out = _split_logits(x2, w_parts, w_cat)
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
        dx, dw = _head_grads(g_parts, x2, w_parts, ctx.dx_terms, ctx.needs_w)
        return dx.to(torch.bfloat16).reshape(ctx.x_shape), dw, None, None, None


def _head_grads(g_parts, x2: torch.Tensor, w_parts, dx_terms: int, needs_w: bool):
    """fp32 (d hidden [T, H], d W [V, H] or None) from the bf16 planes of d logits [T, V], the bf16 hidden
    states and the bf16 planes of the weight - every GEMM on the bf16 cores, fp32 accumulation."""
    n_g, n_w = len(g_parts), len(w_parts)
    # d x = G W: keep the terms down to the second order of the splits (hi*hi, hi*lo, lo*hi)
    pairs = [(i, j) for i in range(n_g) for j in range(n_w) if i + j < max(n_g, n_w)][:dx_terms]
    dx = _mm_acc(g_parts, list(w_parts), pairs)
    dw = None
    if needs_w:  # d W = G^T x, x exact (returned in fp32; autograd casts it to a bf16 parameter's dtype)
        T, V = g_parts[0].shape
        stacked = (n_g == 2 and g_parts[0].is_contiguous() and g_parts[1].is_contiguous()
                   and g_parts[1].data_ptr() == g_parts[0].data_ptr() + g_parts[0].numel() * 2)
        if stacked:  # the two planes are one [2T, V] buffer: G_hi^T x + G_lo^T x as ONE GEMM over 2T
            g_cat = torch.as_strided(g_parts[0], (2 * T, V), (V, 1))
            dw = torch.mm(g_cat.t(), torch.cat([x2, x2], dim=0), out_dtype=torch.float32)
        else:
            dw = _mm_acc([p.t() for p in g_parts], [x2], [(i, 0) for i in range(n_g)])
    return dx, dw


def _split_logits(x2: torch.Tensor, w_parts, w_cat) -> torch.Tensor:
    if w_cat is not None:
        return torch.mm(torch.cat([x2] * len(w_parts), dim=1), w_cat.t(), out_dtype=torch.float32)
    return _mm_acc([x2], [p.t() for p in w_parts], [(0, j) for j in range(len(w_parts))])


class _SplitHeadLossFn(torch.autograd.Function):
    """(hidden [1, T, H] bf16, weight) -> (loss, stats) with the library head GEMMs: the fp32 logits exist only
    between the head GEMM and ONE fused pass over them that leaves log-probs / entropy and d logits as two bf16
    planes (`prl_fused_logits_loss_planes`) - the operand format of the backward GEMMs, so no fp32 gradient
    and no separate split pass (1.9 ms per 8192 x 152 064 micro-batch) exist."""

    @staticmethod
    def forward(ctx, hidden, weight, w_parts, w_cat, dx_terms, batch, cfg, temperature):  # type: ignore[override]
        import ctypes

        from . import _lib
        from .finetune.rl import grpo_loss_from_logprobs

        if hidden.dtype != torch.bfloat16:
            raise TypeError("the split head expects bf16 hidden states (they are exact bf16 operands)")
        if hidden.dim() != 3 or hidden.shape[0] != batch.input_ids.shape[0] or hidden.shape[1] != batch.input_ids.shape[1]:
            raise ValueError("hidden states [B, L, H] do not match the batch")
        _lib.require_device(hidden, weight, batch.input_ids)
        lib = _lib.load()
        B, L, _ = hidden.shape
        x2 = hidden.detach().reshape(B * L, hidden.shape[-1])
        dev = hidden.device
        V = w_parts[0].shape[0]
        nlp = torch.empty((B, L), dtype=torch.float32, device=dev)
        ent, lse2 = torch.empty_like(nlp), torch.empty_like(nlp)
        need_grad = hidden.requires_grad or weight.requires_grad
        ctx.sentinel = bool(batch.sentinel)
        idx = None
        if ctx.sentinel or not need_grad:
            planes = None
            if ctx.sentinel:
                nlp.zero_()
                ent.zero_()
            else:
                from .finetune.rl import logprob_entropy

                nlp, ent, _, _ = logprob_entropy(_split_logits(x2, w_parts, w_cat).reshape(B, L, V), batch.input_ids, temperature)
        else:
            from .fused_head import _MIN_SKIP_FRACTION, _labelled_rows

            # only the rows that predict a labelled token enter the loss (rl/__init__.py:238-250): gather them (and their
            # token columns) into a compact problem, as fused_head._FusedHeadLossFn does - GEMMs and the plane pass then
            # cost what the labelled tokens cost
            rows = _labelled_rows(batch.labels)
            if 0 < rows.numel() <= (1.0 - _MIN_SKIP_FRACTION) * B * L:
                idx = rows
                n = idx.numel()
                xc = torch.zeros((n + 1, x2.shape[1]), dtype=torch.bfloat16, device=dev)  # + a closing row that predicts nothing
                xc[:n] = x2.index_select(0, idx)

                def col(t, fill=0):  # token-aligned column of the compact problem: entry j + 1 belongs to compact row j
                    out = torch.full((1, n + 1), fill, dtype=t.dtype, device=dev)
                    out[0, 1:] = t.reshape(-1).index_select(0, idx + 1)
                    return out

                c_ids, c_lab = col(batch.input_ids), col(batch.labels, -100)
                cols = [col(t) for t in (batch.old_logprobs, batch.ref_logprobs, batch.advantages, batch.rewards, batch.group_tokens, batch.overflow)]
                rows_k, cols_k, x_used = 1, n + 1, xc
            else:
                cont = lambda t: t if t.is_contiguous() else t.contiguous()  # noqa: E731
                c_ids, c_lab = cont(batch.input_ids), cont(batch.labels)
                cols = [cont(t) for t in (batch.old_logprobs, batch.ref_logprobs, batch.advantages, batch.rewards, batch.group_tokens, batch.overflow)]
                rows_k, cols_k, x_used = B, L, x2
            logits = _split_logits(x_used, w_parts, w_cat)
            planes = torch.empty((2, rows_k * cols_k, V), dtype=torch.bfloat16, device=dev)
            k_nlp = torch.empty((rows_k, cols_k), dtype=torch.float32, device=dev)
            k_ent, k_lse = torch.empty_like(k_nlp), torch.empty_like(k_nlp)
            with torch.cuda.device(dev):
                _lib.check(lib.prl_fused_logits_loss_planes(
                    ctypes.byref(cfg), rows_k, cols_k, V, logits.data_ptr(), V, float(temperature), c_ids.data_ptr(), c_lab.data_ptr(),
                    *[c.data_ptr() for c in cols], k_nlp.data_ptr(), k_ent.data_ptr(), k_lse.data_ptr(), planes[0].data_ptr(),
                    planes[1].data_ptr(), V, _lib.current_stream_ptr(dev)))
            del logits
            if idx is not None:
                nlp.zero_()
                ent.zero_()
                nlp.view(-1).index_copy_(0, idx + 1, k_nlp[0, 1:])
                ent.view(-1).index_copy_(0, idx + 1, k_ent[0, 1:])
                x2 = xc
            else:
                nlp, ent = k_nlp, k_ent
        loss, stats, _, _ = grpo_loss_from_logprobs(cfg, batch, nlp, ent, want_grad=False)
        ctx.planes = planes
        ctx.rows = idx
        ctx.save_for_backward(x2, *w_parts)
        ctx.meta = (hidden.shape, weight.shape, weight.dtype, dx_terms)
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, grad_loss, _grad_stats):  # type: ignore[override]
        x2, *w_parts = ctx.saved_tensors
        h_shape, w_shape, w_dtype, dx_terms = ctx.meta
        want_h, want_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        planes, ctx.planes = ctx.planes, None
        if planes is None:  # sentinel batch: zero gradient, no GEMM
            dx = torch.zeros(h_shape, dtype=torch.bfloat16, device=x2.device) if want_h else None
            dw = torch.zeros(w_shape, dtype=w_dtype, device=x2.device) if want_w else None
            return dx, dw, None, None, None, None, None, None
        dx, dw = _head_grads([planes[0], planes[1]], x2, w_parts, dx_terms, want_w)
        # the upstream factor of the loss is applied to the small results ([T, H], [V, H]), never to [T, V]
        up = grad_loss.to(torch.float32)
        if want_h:
            dx = (dx * up).to(torch.bfloat16)
            if ctx.rows is not None:  # compact rows back to their places; rows without a label have no gradient
                full = torch.zeros((h_shape[0] * h_shape[1], h_shape[2]), dtype=torch.bfloat16, device=dx.device)
                full.index_copy_(0, ctx.rows, dx[:-1])
                dx = full
            dx = dx.reshape(h_shape)
        else:
            dx = None
        if dw is not None:
            dw = dw.mul_(up)
        return dx, dw, None, None, None, None, None, None


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

    def rl_loss(self, hidden: torch.Tensor, batch, config, current_step: int, max_step: int):
        """Hidden states -> (loss, stats), the `rl_step` contract (reference rl/__init__.py:136-143), without an
        fp32 d-logits tensor: head GEMM, one fused pass that leaves d logits as bf16 planes, backward GEMMs on
        those planes.  ppo / reinforce (the policies of the fused logits kernel)."""
        from .finetune.rl import STAT_INDEX, check_finite, make_loss_config, stats_to_dict

        if config.policy_loss == "gspo":
            raise NotImplementedError("the plane-emitting pass covers ppo / reinforce; gspo goes through rl_step")
        cfg, kl_coef, ent_coef = make_loss_config(config, current_step, max_step)
        parts = self._split()
        h = hidden if hidden.dtype == torch.bfloat16 else hidden.to(torch.bfloat16)
        loss, stats_dev = _SplitHeadLossFn.apply(h, self.weight, parts, self._cat, self.hidden_grad_terms, batch, cfg, config.temperature)
        stats = stats_dev.cpu().tolist()
        check_finite(stats)
        input_size = batch.input_ids.numel()
        if int(stats[STAT_INDEX["num_output_tokens_sum"]]) == 0:
            return loss, {"input_size": float(input_size)}
        return loss, stats_to_dict(stats, kl_coef, ent_coef, input_size)


def rl_step_split_head(model, batch, current_step: int, max_step: int, config, seq_parallel_group=None):
    """`rl_step` (same signature and return value) for a causal LM with `.model` (body) and an fp32 bias-free
    `.lm_head`, on the LIBRARY head GEMMs: body -> hidden states -> `SplitBf16LmHead.rl_loss`.  Against
    `rl_step` + `apply_fp32_lm_head` it saves the fp32 d-logits tensor and the pass that splits it; against
    `fused_head.rl_step_fused_head` it keeps the `[T, V]` logits (5 GB per 7B micro-batch) and in exchange
    needs no recomputation in backward."""
    body, lin = getattr(model, "model", None), getattr(model, "lm_head", None)
    if body is None or lin is None or getattr(lin, "bias", None) is not None:
        raise TypeError("rl_step_split_head needs model.model (body) and a bias-free model.lm_head")
    inputs = {"input_ids": batch.input_ids, "attention_mask": batch.attention_mask}
    if batch.is_packed:
        inputs["position_ids"] = batch.position_ids
    out = body(**inputs)
    hidden = out[0] if isinstance(out, (tuple, list)) else getattr(out, "last_hidden_state", out)
    w = lin.weight
    if w.dtype != torch.float32:
        raise TypeError("rl_step_split_head is for an fp32 head weight; a bf16 (tied) head needs no split: use rl_step")
    head = getattr(lin, "_prl_split_lm_head", None)  # lives on the module: one per head, gone with the model
    if head is None or head.weight is not w:
        head = SplitBf16LmHead(w)
        object.__setattr__(lin, "_prl_split_lm_head", head)
    return head.rl_loss(hidden, batch, config, current_step, max_step)


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
