"""Fused output head: hidden states -> new_logprobs / entropy -> GRPO loss, with the
[T, V] logits never materialised, forward or backward.

The reference runs `model(**inputs).logits` - an fp32 `lm_head` (finetune/checkpoints.py:87-103)
producing a [1, T, V] fp32 tensor, 4.98 GB per Qwen2.5-7B micro-batch - and then rl_step's K1
(rl/__init__.py:204-233).  Here `rl_step_fused_head` asks the model for its LAST HIDDEN STATES and
hands them, with the head's weight, to the MFMA kernels of csrc/prl_lmhead.hip:

    hidden [T, H] bf16, W [V, H] (fp32 split into two bf16 planes, or bf16 as is)
      --prl_lm_head_logprob_fwd-->  new_logprobs, entropy, lse2        (online softmax in the GEMM epilogue)
      --K2+K3-->                    loss, 32 stats, d loss / d new_logprobs, d loss / d entropy
      --prl_lm_head_logprob_bwd-->  d hidden, d W  (logits recomputed per row chunk, d logits live as bf16
                                    planes of ONE chunk only)

Memory: no d logits in fp32 and no autograd copies of the logits; the backward workspace holds the two bf16 d-logits
planes of one row chunk.  By default (`keep_logits`) the forward leaves the fp32 logits of the micro-batch behind for the
backward (4.98 GB from the head's forward to its backward, two plane products less); with `keep_logits=False` there are
no logits at all and the backward recomputes them chunk by chunk.  Numerics: bf16 x bf16 products are exact in fp32 and accumulate in fp32; the
two-plane split reproduces the fp32 head to ~2^-17 relative.
"""

from __future__ import annotations

import ctypes
import os
from typing import Any

import torch

from . import _lib
from .finetune.rl import RLConfig, _ValueLossFn, _with_advantages, grpo_loss_from_logprobs, gspo_segment_terms, host_stats, make_loss_config
from .finetune.types import PipelineBatchEncoding
from ._lib import STAT_INDEX


def _weight_key(w: torch.Tensor) -> tuple:
    return (w._version, w.data_ptr(), w.device, tuple(w.shape), w.dtype)


class FusedLmHead:
    """Prepared operands of one output-head weight [V, H]: bf16 planes (hi, lo) and their transposes.
    Refreshed when the weight changes (version counter / storage address); writers that go through
    `.data.copy_()` must call `invalidate()` - same contract as lm_head.SplitBf16LmHead."""

    def __init__(self, weight: torch.Tensor, backward: bool = True, chunk_rows: int = 8192, hidden_grad_terms: int = 3,
                 skip_unlabelled: bool = True, keep_logits: bool | None = None):
        """`chunk_rows`: logits rows whose d logits planes live in the workspace at a time (2 x chunk x V x 2 bytes: 5 GB for
        8192 rows of a 152 064-entry vocabulary; one chunk per 8192-token micro-batch halves the d W epilogues).
        `hidden_grad_terms`: 3 = d hidden from every bf16 product (fp32-GEMM accuracy before the final rounding),
        2 = (d logits_hi + d logits_lo) x W_hi: the fp32 weight rounded to bf16 for this product only, the d logits in full (2^-9 of
        each weight; one product less, on the forward's dual-plane core), 1 = leading product only (2^-9 relative in both operands,
        the size of the bf16 rounding of d hidden; two products less).
        `skip_unlabelled` (loss path only, `fused_head_loss`): rows whose next token carries no label (prompt and
        observation tokens, sequence starts, padding) enter neither the loss nor any statistic - the head runs,
        forward and backward, on the labelled rows only; see `_FusedHeadLossFn`.
        `keep_logits` (default on, PRL_LMHEAD_KEEP_LOGITS=0 turns it off): a forward that will be followed by a backward also
        writes its logits (fp32, rows x vocab x 4 bytes - they live from the head's forward to the end of its backward, where the
        d-logits planes of the recomputing form live anyway) and the backward forms d logits in one pass over them instead of
        recomputing both plane products: 5 products per micro-batch instead of 7.  Off: no logits anywhere, the backward
        recomputes them chunk by chunk (workspace only)."""
        if weight.dim() != 2:
            raise ValueError("lm_head weight must be [vocab, hidden]")
        if weight.dtype not in (torch.float32, torch.bfloat16):
            raise TypeError(f"lm_head weight must be float32 or bfloat16, got {weight.dtype}")
        self.weight = weight
        self.backward = backward
        self.chunk_rows = int(chunk_rows)
        self.hidden_grad_terms = int(hidden_grad_terms)
        if self.hidden_grad_terms not in (1, 2, 3):
            raise ValueError("hidden_grad_terms must be 1, 2 or 3")
        self.skip_unlabelled = bool(skip_unlabelled)
        self.keep_logits = bool(int(os.environ.get("PRL_LMHEAD_KEEP_LOGITS", "1"))) if keep_logits is None else bool(keep_logits)
        self._key = None
        self.w_hi = self.w_lo = self.wt_hi = self.wt_lo = None
        self._ws: dict[Any, torch.Tensor] = {}

    @property
    def vocab(self) -> int:
        return self.weight.shape[0]

    @property
    def hidden(self) -> int:
        return self.weight.shape[1]

    def invalidate(self) -> None:
        self._key = None

    def attach_optimizer(self, optimizer: torch.optim.Optimizer) -> None:
        optimizer.register_step_post_hook(lambda *_: self.invalidate())

    def refresh(self) -> None:
        w = self.weight
        key = _weight_key(w)
        if key == self._key:
            return
        _lib.require_device(w)
        lib = _lib.load()
        V, H = w.shape
        dev = w.device
        src = w.detach()
        if not src.is_contiguous():
            src = src.contiguous()

        def plane(old, *shape):
            """A plane buffer is REUSED across refreshes (same address: launches captured in a HIP graph keep reading the right
            memory after an optimizer step, and a step does not re-allocate 2-4 planes of V x H)."""
            if old is not None and tuple(old.shape) == shape and old.device == dev and old.dtype == torch.bfloat16 and old.data_ptr() != src.data_ptr():
                return old
            return torch.empty(shape, dtype=torch.bfloat16, device=dev)

        if w.dtype == torch.bfloat16:  # a bf16 weight (tied embedding) is its own exact plane
            self.w_hi, self.w_lo = src, None
            self.wt_hi, self.wt_lo = (plane(self.wt_hi, H, V) if self.backward else None), None
            outs = (None, None, self.wt_hi, None)
        else:
            self.w_hi, self.w_lo = plane(self.w_hi, V, H), plane(self.w_lo, V, H)
            self.wt_hi, self.wt_lo = (plane(self.wt_hi, H, V), plane(self.wt_lo, H, V)) if self.backward else (None, None)
            outs = (self.w_hi, self.w_lo, self.wt_hi, self.wt_lo)
        if any(o is not None for o in outs):
            with torch.cuda.device(dev):
                _lib.check(lib.prl_lm_head_prepare(V, H, src.data_ptr(), 0 if w.dtype == torch.float32 else 1,
                                                   *[_lib.ptr(o) for o in outs], _lib.current_stream_ptr(dev)))
        self._key = key

    def _workspace(self, kind: str, rows: int, cols: int, dev: torch.device, chunk_rows: int) -> torch.Tensor:
        fwd, bwd = ctypes.c_size_t(0), ctypes.c_size_t(0)
        _lib.check(_lib.load().prl_lm_head_workspace_bytes(rows, cols, self.hidden, self.vocab, chunk_rows, ctypes.byref(fwd), ctypes.byref(bwd)))
        need = fwd.value if kind == "fwd" else bwd.value
        key = (kind, dev, _lib.current_stream_ptr(dev))
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        return ws

    def can_keep_logits(self, rows: int, device: torch.device) -> bool:
        """Whether `rows` x vocab fp32 logits (+ the same again for the backward's two bf16 d-logits planes, which live
        while the kept logits still do) fit comfortably in what the device has free right now - the allocator's cached
        blocks included.  The kept form trades 4.98 GB per 8192 x 152 064 micro-batch for two plane products; on a GPU
        that is nearly full (longer micro-batches, a co-located inference engine) the recomputing form is the right one,
        and `_FusedHeadLossFn` falls back to it instead of running out of memory."""
        if not self.keep_logits or self.vocab % 8:
            return False
        need = 2 * rows * self.vocab * 4
        try:
            free, _ = torch.cuda.mem_get_info(device)
            cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        except Exception:  # noqa: BLE001 - no way to ask: keep the configured behaviour
            return True
        return need <= 0.9 * (free + max(cached, 0))

    # -- forward ------------------------------------------------------------------------------------
    def logprob_entropy(self, hidden: torch.Tensor, input_ids: torch.Tensor, temperature: float, keep: bool = False):
        """hidden [B, L, H] -> token-aligned (new_logprobs, entropy, lse2), each fp32 [B, L], and the bf16 hidden states the
        kernels read; no graph.  `keep`: a fifth value, the [B * L, V] fp32 logits in base-2 units for `backward_from_token_grads`."""
        lib = _lib.load()
        _lib.require_device(hidden, input_ids)
        self.refresh()
        B, L, H = hidden.shape
        if H != self.hidden:
            raise ValueError(f"hidden size {H} != weight's {self.hidden}")
        h = hidden.detach()
        if h.dtype != torch.bfloat16:
            h = h.to(torch.bfloat16)
        if not h.is_contiguous():
            h = h.contiguous()
        ids = input_ids if input_ids.is_contiguous() else input_ids.contiguous()
        dev = h.device
        nlp = torch.empty((B, L), dtype=torch.float32, device=dev)
        ent = torch.empty_like(nlp)
        lse2 = torch.empty_like(nlp)
        if keep and self.vocab % 8:
            raise ValueError("kept logits need a vocabulary that is a multiple of 8")
        kept = torch.empty((B * L, self.vocab), dtype=torch.float32, device=dev) if keep else None
        ws = self._workspace("fwd", B, L, dev, self.chunk_rows)
        head = (B, L, H, self.vocab, h.data_ptr(), self.w_hi.data_ptr(), _lib.ptr(self.w_lo), ids.data_ptr(), float(temperature),
                nlp.data_ptr(), ent.data_ptr(), lse2.data_ptr())
        tail = (ws.data_ptr(), ws.numel(), _lib.current_stream_ptr(dev))
        with torch.cuda.device(dev):
            if keep:
                _lib.check(lib.prl_lm_head_logprob_fwd_keep(*head, kept.data_ptr(), *tail))
            else:
                _lib.check(lib.prl_lm_head_logprob_fwd(*head, *tail))
        return (nlp, ent, lse2, h, kept) if keep else (nlp, ent, lse2, h)

    # -- backward -----------------------------------------------------------------------------------
    def backward_from_token_grads(self, h: torch.Tensor, input_ids: torch.Tensor, temperature: float, lse2: torch.Tensor,
                                  ent: torch.Tensor, g_nlp: torch.Tensor, g_ent: torch.Tensor | None, upstream: torch.Tensor | None,
                                  want_hidden: bool = True, grad_weight: torch.Tensor | None = None,
                                  grad_hidden_dtype: torch.dtype = torch.bfloat16, chunk_rows: int | None = None,
                                  overwrite_weight_grad: bool = False, kept_logits: torch.Tensor | None = None):
        """d hidden (returned) and d W (ACCUMULATED into `grad_weight`, fp32 [V, H]; with `overwrite_weight_grad`
        the buffer may be uninitialised and is overwritten) from the token-aligned gradients of new_logprobs / entropy.
        `kept_logits`: the fifth value of `logprob_entropy(..., keep=True)` for the same hidden states - no recompute."""
        if not self.backward:
            raise RuntimeError("this FusedLmHead was built with backward=False")
        lib = _lib.load()
        self.refresh()
        B, L, H = h.shape
        dev = h.device
        chunk = int(chunk_rows or self.chunk_rows)
        gh = torch.empty((B, L, H), dtype=grad_hidden_dtype, device=dev) if want_hidden else None
        if grad_weight is not None and (grad_weight.dtype != torch.float32 or not grad_weight.is_contiguous() or tuple(grad_weight.shape) != (self.vocab, H)):
            raise ValueError("grad_weight must be a contiguous float32 [vocab, hidden] tensor")
        ws = self._workspace("bwd", B, L, dev, chunk)
        ids = input_ids if input_ids.is_contiguous() else input_ids.contiguous()
        flags = ({1: _lib.PRL_LM_HEAD_DH_LEADING_TERM, 2: _lib.PRL_LM_HEAD_DH_NO_WEIGHT_LO}.get(self.hidden_grad_terms, 0)
                 | (_lib.PRL_LM_HEAD_DW_OVERWRITE if overwrite_weight_grad else 0))
        tail = (ids.data_ptr(), float(temperature), lse2.data_ptr(), ent.data_ptr(), g_nlp.data_ptr(), _lib.ptr(g_ent), _lib.ptr(upstream),
                _lib.ptr(gh), 0 if grad_hidden_dtype == torch.float32 else 1, _lib.ptr(grad_weight), chunk, flags, ws.data_ptr(), ws.numel(),
                _lib.current_stream_ptr(dev))
        with torch.cuda.device(dev):
            if kept_logits is not None:
                if kept_logits.dtype != torch.float32 or not kept_logits.is_contiguous() or tuple(kept_logits.shape) != (B * L, self.vocab):
                    raise ValueError("kept_logits must be the contiguous float32 [rows, vocab] tensor of the forward")
                _lib.check(lib.prl_lm_head_logprob_bwd_kept(B, L, H, self.vocab, h.data_ptr(), kept_logits.data_ptr(), self.wt_hi.data_ptr(),
                                                            _lib.ptr(self.wt_lo), *tail))
            else:
                _lib.check(lib.prl_lm_head_logprob_bwd(B, L, H, self.vocab, h.data_ptr(), self.w_hi.data_ptr(), _lib.ptr(self.w_lo),
                                                       self.wt_hi.data_ptr(), _lib.ptr(self.wt_lo), *tail))
        return gh


# below this fraction of unlabelled rows the gather / scatter around the compact problem is not worth it
_MIN_SKIP_FRACTION = 1.0 / 32.0


def _labelled_rows(labels: torch.Tensor) -> torch.Tensor:
    """Flat indices q = b * L + c of the logits rows that predict a labelled token (labels[b, c + 1] != -100)."""
    live = torch.zeros_like(labels, dtype=torch.bool)
    live[:, :-1] = labels[:, 1:] != -100
    return live.flatten().nonzero().squeeze(1)  # the one host synchronisation of this path


class _FusedHeadLossFn(torch.autograd.Function):
    """(hidden, weight) -> (loss, stats): K1 inside the head GEMM, K2+K3, hand-written backward.

    With `head.skip_unlabelled` the head sees a COMPACT problem: the hidden states of the rows that predict a labelled
    token, gathered into one row-major block (plus one closing row that predicts nothing), with their target ids next
    to them.  The reference computes log-probabilities for every position and masks afterwards
    (rl/__init__.py:207-233, 238-250: every term and statistic carries the `labels != -100` mask), so loss, statistics
    and gradients are unchanged - the unlabelled positions of `new_logprobs` / `entropy` are simply 0 instead of values
    nobody reads, d hidden of those rows is exactly 0 as before, and forward + backward cost what the LABELLED tokens
    cost.  The kernels are the same ones; only which rows they are handed changes (one `nonzero()` = one host sync)."""

    @staticmethod
    def forward(ctx, hidden, weight, head: FusedLmHead, batch, cfg, temperature, chunk_rows, sp_group=None):  # type: ignore[override]
        rows = None
        if head.skip_unlabelled and not batch.sentinel:
            _lib.require_device(hidden, batch.labels)
            # the loader thread may have found the rows on the host already (finetune_loop.annotate_host_batch): no sync here then
            rows = getattr(batch, "model_extra", {}).get("labelled_rows")
            if rows is None or rows.device != hidden.device:
                rows = _labelled_rows(batch.labels)
        ctx.sentinel = bool(batch.sentinel) or (rows is not None and rows.numel() == 0)
        if ctx.sentinel:
            # a sentinel batch (finetune/utils.py:17-78), or any batch without a labelled token: loss 0, statistics of an
            # empty batch, gradient 0 - no GEMM forward or backward, the wrappers still see a gradient for every input
            _lib.require_device(hidden, weight)
            B, L, _ = hidden.shape
            zeros = torch.zeros((B, L), dtype=torch.float32, device=hidden.device)
            loss, stats, _, _ = grpo_loss_from_logprobs(cfg, batch, zeros, zeros, want_grad=False)
            ctx.shapes = (hidden.shape, hidden.dtype, weight.shape, weight.dtype, hidden.device)
            ctx.mark_non_differentiable(stats)
            return loss, stats
        need_grad = hidden.requires_grad or weight.requires_grad
        B, L, H = hidden.shape
        idx = rows if (rows is not None and rows.numel() <= (1.0 - _MIN_SKIP_FRACTION) * B * L) else None
        n_rows = B * L if idx is None else idx.numel() + 1
        keep = need_grad and head.can_keep_logits(n_rows, hidden.device)
        kept = None

        def forward_head(h_in, ids_in):
            """The head's forward; a kept-logits allocation that does not fit after all falls back to the recomputing form."""
            if keep:
                try:
                    return head.logprob_entropy(h_in, ids_in, temperature, keep=True)
                except torch.cuda.OutOfMemoryError:
                    torch.cuda.empty_cache()
            return head.logprob_entropy(h_in, ids_in, temperature, keep=False)

        if idx is None:
            nlp, ent, lse2, h, *rest = forward_head(hidden, batch.input_ids)
            kept = rest[0] if rest else None
            ids = batch.input_ids
        else:
            n = idx.numel()
            dev = hidden.device
            hc = torch.zeros((1, n + 1, H), dtype=torch.bfloat16, device=dev)  # + one closing row (predicts nothing)
            hc[0, :n] = hidden.detach().reshape(B * L, H).index_select(0, idx)
            ids = torch.zeros((1, n + 1), dtype=torch.int64, device=dev)
            ids[0, 1:] = batch.input_ids.reshape(-1).index_select(0, idx + 1)  # row j predicts ids[j + 1]
            nlp_c, ent_c, lse2, h, *rest = forward_head(hc, ids)
            kept = rest[0] if rest else None
            nlp = torch.zeros((B, L), dtype=torch.float32, device=dev)
            ent = torch.zeros_like(nlp)
            nlp.view(-1).index_copy_(0, idx + 1, nlp_c[0, 1:])  # token-aligned: the value for token u = q + 1
            ent.view(-1).index_copy_(0, idx + 1, ent_c[0, 1:])
        if cfg.policy_loss == _lib.PRL_POLICY_GSPO:
            # sequence-level term (rl/__init__.py:310-352) on the head's log-probs: per-segment sums -> clipped sequence ratio -> a
            # per-token gradient coefficient that joins the token kernel's own terms (KL, entropy) on the way into the head's backward
            seg_loss, ext_g, ext_c = gspo_segment_terms(cfg, batch, nlp, sp_group)
            _, stats, g_nlp, g_ent = grpo_loss_from_logprobs(cfg, batch, nlp, ent, want_grad=need_grad, ext_token_grad=ext_g, ext_clamp_indicator=ext_c)
            loss = seg_loss
            stats[STAT_INDEX["loss"]] = seg_loss.double()
        else:
            loss, stats, g_nlp, g_ent = grpo_loss_from_logprobs(cfg, batch, nlp, ent, want_grad=need_grad)
        if idx is not None:
            # The reference asserts isfinite(new_logprobs) over EVERY position (rl/__init__.py:213).  Rows that were not handed
            # to the head cannot show up in the kernel's counter, so the check moves to their inputs: a non-finite hidden state
            # is what makes a logits row non-finite.  One pass over [T, H] (58 MB for a 7B micro-batch), no host sync - the
            # flag joins the statistics vector that `check_finite` reads after the step's single device->host copy.
            stats[STAT_INDEX["nonfinite_new_logprobs"]] += (~torch.isfinite(hidden.detach())).any().to(stats.dtype)
        if need_grad:
            if idx is not None:  # the token gradients of the compact rows, token-aligned to the compact problem
                def compact(g):
                    out = torch.zeros((1, idx.numel() + 1), dtype=torch.float32, device=g.device)
                    out[0, 1:] = g.reshape(-1).index_select(0, idx + 1)
                    return out

                g_nlp, g_ent, ent = compact(g_nlp), (compact(g_ent) if g_ent is not None else None), ent_c
            ctx.save_for_backward(h, ids, lse2, ent, g_nlp, g_ent if g_ent is not None else torch.empty(0, device=h.device),
                                  idx if idx is not None else torch.empty(0, dtype=torch.int64, device=h.device),
                                  kept if kept is not None else torch.empty(0, device=h.device))
        ctx.has_g_ent = g_ent is not None
        ctx.compact = idx is not None
        ctx.hidden_shape = tuple(hidden.shape)
        ctx.head, ctx.temperature, ctx.chunk_rows = head, float(temperature), chunk_rows
        ctx.hidden_dtype, ctx.weight_dtype = hidden.dtype, weight.dtype
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, grad_loss, _grad_stats):  # type: ignore[override]
        if ctx.sentinel:
            hs, hd, ws_, wd, dev = ctx.shapes
            gh = torch.zeros(hs, dtype=hd, device=dev) if ctx.needs_input_grad[0] else None
            gw = torch.zeros(ws_, dtype=wd, device=dev) if ctx.needs_input_grad[1] else None
            return gh, gw, None, None, None, None, None, None
        h, ids, lse2, ent, g_nlp, g_ent, idx, kept = ctx.saved_tensors
        head: FusedLmHead = ctx.head
        want_h, want_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gw = torch.empty((head.vocab, head.hidden), dtype=torch.float32, device=h.device) if want_w else None
        up = grad_loss.to(torch.float32).contiguous()
        gh = head.backward_from_token_grads(h, ids, ctx.temperature, lse2, ent, g_nlp, g_ent if ctx.has_g_ent else None, up,
                                            want_hidden=want_h, grad_weight=gw, overwrite_weight_grad=True,
                                            grad_hidden_dtype=torch.float32 if ctx.hidden_dtype == torch.float32 else torch.bfloat16,
                                            chunk_rows=ctx.chunk_rows, kept_logits=kept if kept.numel() else None)
        del kept
        if gh is not None and ctx.compact:  # rows without a label have no gradient
            B, L, H = ctx.hidden_shape
            full = torch.zeros((B * L, H), dtype=gh.dtype, device=gh.device)
            full.index_copy_(0, idx, gh[0, :-1])
            gh = full.view(B, L, H)
        if gh is not None and gh.dtype != ctx.hidden_dtype:
            gh = gh.to(ctx.hidden_dtype)
        if gw is not None and gw.dtype != ctx.weight_dtype:
            gw = gw.to(ctx.weight_dtype)
        return gh, gw, None, None, None, None, None, None


def _host_stats(stats_dev: torch.Tensor, batch: PipelineBatchEncoding, kl_coef: float, ent_coef: float, value_loss_coef: float = 0.0):
    return host_stats(stats_dev, batch.input_ids.numel(), kl_coef, ent_coef, value_loss_coef)


def fused_head_loss(hidden: torch.Tensor, weight: torch.Tensor, head: FusedLmHead, batch: PipelineBatchEncoding,
                    config: RLConfig, current_step: int, max_step: int, chunk_rows: int | None = None, values: torch.Tensor | None = None,
                    seq_parallel_group: Any = None):
    """Loss + stats from last hidden states and the head weight; same return contract as `rl_step`.
    `values`: the value head's predictions [B, L] for an actor-critic model (rl/__init__.py:265-272, 367-381, 441-448).
    `seq_parallel_group`: GSPO on a sequence-parallel slice adds its per-segment sums over the group (rl/utils.py:194-206)."""
    cfg, kl_coef, ent_coef = make_loss_config(config, current_step, max_step)
    if values is not None:
        value_loss, value_advantages, vstats_dev = _ValueLossFn.apply(values, batch, cfg)
        batch = _with_advantages(batch, value_advantages)
    loss, stats_dev = _FusedHeadLossFn.apply(hidden, weight, head, batch, cfg, config.temperature, chunk_rows, seq_parallel_group)
    if values is not None:
        loss = loss + config.value_loss_coef * value_loss
        stats_dev = torch.cat([stats_dev, vstats_dev])
    return loss, _host_stats(stats_dev, batch, kl_coef, ent_coef, config.value_loss_coef)



def _logits_postprocessing(model: Any) -> str | None:
    """Why `model(...).logits` is NOT simply `lm_head(body(...))` for this model, or None when it is.  The fused head computes
    hidden @ W^T and nothing else: an architecture that post-processes its logits (Gemma-2's `final_logit_softcapping`, Cohere's
    `logit_scale`, Granite's `logits_scaling`), or whose head is wider than the vocabulary it samples from (a padded head whose
    extra columns the model's own forward cuts off), would get different log-probabilities - a wrong KL term - without a
    warning.  Judged from the model's `config`; a model without one is taken at its word."""
    cfg = getattr(model, "config", None)
    if cfg is None:
        return None
    if getattr(cfg, "final_logit_softcapping", None):
        return f"config.final_logit_softcapping = {cfg.final_logit_softcapping}"
    for key in ("logit_scale", "logits_scaling"):
        v = getattr(cfg, key, None)
        if v is not None and float(v) != 1.0:
            return f"config.{key} = {v}"
    head = getattr(model, "lm_head", None)
    rows = getattr(head, "out_features", None)
    vocab = getattr(cfg, "vocab_size", None)
    if rows is not None and vocab is not None and int(rows) != int(vocab):
        return f"lm_head has {rows} rows, config.vocab_size is {vocab}"
    return None


def _body_and_head(model: Any):
    body = getattr(model, "model", None)
    lm_head = getattr(model, "lm_head", None)
    if body is None or lm_head is None or getattr(lm_head, "bias", None) is not None:
        raise TypeError("the fused head needs model.model (body) and a bias-free model.lm_head")
    why = _logits_postprocessing(model)
    if why is not None:
        raise TypeError(f"the fused head computes hidden @ W^T only, but this model post-processes its logits ({why}); use rl_step on its .logits")
    return body, lm_head


def _lm_of(model: Any) -> Any:
    """The causal LM inside an actor-critic wrapper (`AutoModelForCausalLMWithValueHead.pretrained_model`, reference
    finetune/value_model.py:54-116); any other model is its own LM."""
    return getattr(model, "pretrained_model", model) if getattr(model, "value_head", None) is not None else model


def _hidden_states(body: Any, batch: PipelineBatchEncoding) -> torch.Tensor:
    inputs = {"input_ids": batch.input_ids, "attention_mask": batch.attention_mask}
    if batch.is_packed:
        inputs["position_ids"] = batch.position_ids
    out = body(**inputs)
    return out[0] if isinstance(out, (tuple, list)) else getattr(out, "last_hidden_state", out)


def _head_for(owner: Any, weight: torch.Tensor, chunk_rows: int, hidden_grad_terms: int = 3, keep_logits: bool | None = None) -> FusedLmHead:
    """The FusedLmHead of `owner` (the lm_head module).  It lives ON the module - one per head, gone with the model - and is
    re-bound when the module hands out a new tensor object for its weight (FSDP with use_orig_params=False and re-created
    models do, on every forward): the planes and workspaces are reused, only the split is redone."""
    head = getattr(owner, "_prl_fused_lm_head", None)
    stale = head is None or tuple(head.weight.shape) != tuple(weight.shape) or head.weight.dtype != weight.dtype \
        or head.weight.device != weight.device or head.hidden_grad_terms != hidden_grad_terms
    if stale or (keep_logits is not None and head.keep_logits != bool(keep_logits)) or head.chunk_rows != int(chunk_rows):
        head = FusedLmHead(weight, backward=True, chunk_rows=chunk_rows, hidden_grad_terms=hidden_grad_terms, keep_logits=keep_logits)
        object.__setattr__(owner, "_prl_fused_lm_head", head)
    elif head.weight is not weight:
        head.weight = weight
        head.invalidate()
    return head


def install_fused_head(model: Any, chunk_rows: int = 8192, hidden_grad_terms: int = 3, keep_logits: bool | None = None) -> Any:
    """Teach a causal LM (`.model` body + bias-free `.lm_head`, the Hugging Face layout) to compute the RL loss
    INSIDE its own forward: `model(rl_batch=batch, rl_config=config, current_step=s, max_step=m)` returns
    `(loss, stats_device)`; every other call is the model's original forward.  Call this BEFORE wrapping the
    model in DistributedDataParallel / FSDP / `accelerator.prepare`: those wrappers arm their gradient
    reduction in THEIR forward, so the loss has to be produced by a call that goes through them -
    `rl_step_fused_head(wrapped, ...)` does exactly that.  Parameter names are unchanged (the weight-update
    path keeps seeing `model.*` / `lm_head.weight`).

    The head reads `lm_head.weight` directly, `lm_head.forward` is never called.  DDP, FSDP (the weight is part of
    a unit that is unsharded while the wrapped forward runs) and ZeRO stages 1-2 are fine with that; ZeRO-3
    gathers a parameter in its module's pre-forward hook, which a bypassed module never fires - there the call
    has to sit inside `deepspeed.zero.GatheredParameters([model.lm_head.weight])` (a partitioned placeholder is
    refused by shape, not silently used)."""
    _body_and_head(_lm_of(model))
    if getattr(model, "_prl_fused_head", None) is not None:
        return model
    original = model.forward

    def forward(*args, rl_batch: PipelineBatchEncoding | None = None, rl_config: RLConfig | None = None,
                current_step: int = 0, max_step: int = 1, seq_parallel_group: Any = None, **kwargs):
        if rl_batch is None:
            return original(*args, **kwargs)
        body, lm_head = _body_and_head(_lm_of(model))
        hidden = _hidden_states(body, rl_batch)
        w = lm_head.weight
        cfg, _, _ = make_loss_config(rl_config, current_step, max_step)
        opts = model._prl_fused_head
        value_head = getattr(model, "value_head", None)
        if value_head is not None:  # actor-critic wrapper: advantages := rewards - V, value loss and its five statistics ride along
            value_loss, value_advantages, vstats_dev = _ValueLossFn.apply(value_head(hidden), rl_batch, cfg)
            rl_batch = _with_advantages(rl_batch, value_advantages)
        loss, stats_dev = _FusedHeadLossFn.apply(hidden, w, _head_for(lm_head, w, opts["chunk_rows"], opts["hidden_grad_terms"], opts["keep_logits"]),
                                                 rl_batch, cfg, rl_config.temperature, opts["chunk_rows"], seq_parallel_group)
        if value_head is not None:
            loss = loss + rl_config.value_loss_coef * value_loss
            stats_dev = torch.cat([stats_dev, vstats_dev])
        return loss, stats_dev

    # `keep_logits`: None = the default (on, unless PRL_LMHEAD_KEEP_LOGITS=0, and only while two copies of the micro-batch's logits
    # fit in free device memory); False = never (no logits anywhere, 2 more plane products in the backward); `chunk_rows`: rows of
    # d-logits planes in the backward workspace at a time (2 x chunk_rows x vocab x 2 bytes)
    model._prl_fused_head = {"chunk_rows": int(chunk_rows), "hidden_grad_terms": int(hidden_grad_terms), "keep_logits": keep_logits}
    model.forward = forward
    return model


def rl_step_fused_head(model: Any, batch: PipelineBatchEncoding, current_step: int, max_step: int, config: RLConfig,
                       seq_parallel_group=None, chunk_rows: int | None = None, keep_logits: bool | None = None):
    """`rl_step` (reference rl/__init__.py:136-143, same signature and return value) for a causal LM
    that exposes its body and head separately, as Hugging Face models do (`model.model`,
    `model.lm_head`): the body runs as usual, the head never produces logits.

    A bare model is driven directly.  A model prepared with `install_fused_head` - bare or inside
    DistributedDataParallel / FSDP / an accelerate wrapper (anything that exposes it as `.module`) - is driven
    through ITS forward, which is what data-parallel training needs."""
    inner = model
    while getattr(inner, "_prl_fused_head", None) is None and hasattr(inner, "module"):
        inner = inner.module
    if getattr(inner, "_prl_fused_head", None) is not None:
        _, kl_coef, ent_coef = make_loss_config(config, current_step, max_step)
        loss, stats_dev = model(rl_batch=batch, rl_config=config, current_step=current_step, max_step=max_step,
                                **({"seq_parallel_group": seq_parallel_group} if seq_parallel_group is not None else {}))
        return loss, _host_stats(stats_dev, batch, kl_coef, ent_coef, config.value_loss_coef)
    # an actor-critic wrapper (finetune/value_model.py:54-116): the LM is `.pretrained_model`, the critic reads the same hidden states
    value_head = getattr(model, "value_head", None)
    body, lm_head = _body_and_head(_lm_of(model))
    hidden = _hidden_states(body, batch)
    values = value_head(hidden) if value_head is not None else None
    w = lm_head.weight
    # a bare model: the two memory knobs come from the call or from the RLConfig (`fused_head_chunk_rows`, `fused_head_keep_logits`)
    chunk_rows = int(chunk_rows or getattr(config, "fused_head_chunk_rows", 8192) or 8192)
    keep = keep_logits if keep_logits is not None else getattr(config, "fused_head_keep_logits", None)
    return fused_head_loss(hidden, w, _head_for(lm_head, w, chunk_rows, keep_logits=keep), batch, config, current_step, max_step, chunk_rows,
                           values=values, seq_parallel_group=seq_parallel_group)


# -- reference log-probabilities ---------------------------------------------------------
def token_logprobs_from_hidden(head: FusedLmHead, hidden: torch.Tensor, input_ids: torch.Tensor, labels: torch.Tensor | None,
                               temperature: float = 1.0) -> torch.Tensor:
    """log p(token u | prefix) of the LABELLED tokens, 0 elsewhere, fp32 [B, L], from last hidden states - no graph, no
    `[T, V]` tensor: the no-grad forward of the fused head (1 plane product for a bf16 weight, 2 for an fp32 one).

    This is the dense contraction of the reference-policy forward: the reference asks a second inference server for
    `prompt_logprobs` of prompt + completion and keeps the completion tokens' values (preprocess.py:86-104,
    llm.py:606-648), later left-zero-padded to the sequence (rl/__init__.py:573-594).  `labels` marks those tokens
    (labels != -100); only the rows that PREDICT one are handed to the kernels (a compact problem, as in the loss
    path), so prompt tokens cost nothing.  `labels=None`: every token (the first of each row has no prediction: 0)."""
    _lib.require_device(hidden, input_ids)
    B, L, H = hidden.shape
    dev = hidden.device
    with torch.no_grad():
        if labels is None:
            return head.logprob_entropy(hidden, input_ids, temperature)[0]
        idx = _labelled_rows(labels)
        n = idx.numel()
        out = torch.zeros((B, L), dtype=torch.float32, device=dev)
        if n == 0:
            return out
        if n > (1.0 - _MIN_SKIP_FRACTION) * B * L:
            nlp = head.logprob_entropy(hidden, input_ids, temperature)[0]
            return torch.where(labels != -100, nlp, out)
        hc = torch.zeros((1, n + 1, H), dtype=torch.bfloat16, device=dev)  # + one closing row (predicts nothing)
        hc[0, :n] = hidden.detach().reshape(B * L, H).index_select(0, idx)
        ids = torch.zeros((1, n + 1), dtype=torch.int64, device=dev)
        ids[0, 1:] = input_ids.reshape(-1).index_select(0, idx + 1)  # row j predicts ids[j + 1]
        nlp_c = head.logprob_entropy(hc, ids, temperature)[0]
        out.view(-1).index_copy_(0, idx + 1, nlp_c[0, 1:])
        return out


def ref_head_for(ref_model: Any) -> tuple[Any, FusedLmHead] | None:
    """(body, no-grad FusedLmHead) of a frozen causal LM in the Hugging Face layout (`.model` + bias-free `.lm_head`),
    None for anything else (a bare callable that only returns logits).  The head lives on the lm_head module and
    follows its weight like the training head does (`_head_for`); it holds the row-major planes only (`backward=False`:
    no transposed copies, half the memory of a training head).
    None as well - i.e. the caller goes through the model's own `.logits` and K1 - for an architecture that post-processes its
    logits or pads its head (`_logits_postprocessing`).  The hidden states enter the product in bf16 (the head's operand type):
    exact for a bf16 reference policy, a 2^-9 rounding of each hidden value for an fp32 one, whose old path was an fp32
    `F.linear`; pass `fused_head=False` to `annotate_ref_logprobs` to keep that."""
    body = getattr(ref_model, "model", None)
    lm_head = getattr(ref_model, "lm_head", None)
    w = getattr(lm_head, "weight", None)
    if body is None or w is None or getattr(lm_head, "bias", None) is not None or not callable(body):
        return None
    if _logits_postprocessing(ref_model) is not None:
        return None
    if w.dim() != 2 or w.dtype not in (torch.float32, torch.bfloat16) or not w.is_cuda:
        return None
    head = getattr(lm_head, "_prl_ref_lm_head", None)
    if head is None or tuple(head.weight.shape) != tuple(w.shape) or head.weight.dtype != w.dtype or head.weight.device != w.device:
        head = FusedLmHead(w, backward=False, keep_logits=False)
        object.__setattr__(lm_head, "_prl_ref_lm_head", head)
    elif head.weight is not w:
        head.weight = w
        head.invalidate()
    return body, head


def annotate_ref_logprobs_fused(ref_model: Any, batch: PipelineBatchEncoding, temperature: float = 1.0) -> bool:
    """Fill `batch.ref_logprobs` from the reference model's HIDDEN STATES through the MFMA head.  Returns False (and
    leaves the batch alone) when `ref_model` does not expose body and head separately."""
    found = ref_head_for(ref_model)
    if found is None:
        return False
    body, head = found
    with torch.no_grad():
        hidden = _hidden_states(body, batch)
        batch.ref_logprobs = token_logprobs_from_hidden(head, hidden, batch.input_ids, batch.labels, temperature)
    return True
