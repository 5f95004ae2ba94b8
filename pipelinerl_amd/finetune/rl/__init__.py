"""GRPO / PPO / REINFORCE learner step and group-baseline advantages on MI355X.

Drop-in for reference pipelinerl/finetune/rl/__init__.py: same public names and signatures
(`RLConfig`, `rl_step`, `populate_rl_data`, `prepare_rl_fields`, `RL_DATA_COLUMNS`,
`linear_decay_coef`), same results; everything between the model's logits and the scalar
loss runs in the hand-written HIP kernels of libprl.so:

    logits --K1--> new_logprobs, entropy --K2+K3--> loss, 32 stats, d loss/d new_logprobs
           <-------------------- K1 backward: d loss / d logits ----------------------

There is one device->host copy per micro-batch (the 32-double stats vector; the reference
does ~31 `.item()` syncs; backward never touches the host) and no CPU fallback: CPU tensors raise.
"""

from __future__ import annotations

import ctypes
import logging
from dataclasses import dataclass
from typing import Any, Sequence

import numpy as np
import torch
from pydantic import BaseModel, Field

from ... import _lib
from ..._lib import STAT_INDEX, PrlLossConfig
from ...ragged import RaggedRollouts
from ..types import PipelineBatchEncoding

logger = logging.getLogger(__name__)

RL_DATA_COLUMNS = [
    "overflow",
    "group_tokens",
    "num_labels",
    "rewards",
    "advantages",
    "old_logprobs",
    "ref_logprobs",
]


class RLConfig(BaseModel):
    """Hyper-parameters of the RL loss (reference rl/__init__.py:43-105, same defaults)."""

    policy_loss: str = Field(default="ppo", description="ppo | reinforce | gspo")
    use_advantages: bool = Field(default=True, description="weight log-probs by advantages instead of rewards")
    epsilon_low: float = Field(default=0.2, description="lower clip of the importance ratio")
    epsilon_high: float = Field(default=0.2, description="upper clip of the importance ratio")
    batch_size: int = Field(default=0, description="samples per optimizer step; the loss normaliser")
    reward_minus_kl_coef: float = Field(default=0.0, description="declared by the reference, unused")
    kl_coef: float = Field(default=0.1, description="KL-to-reference penalty coefficient")
    final_kl_coef: float = Field(default=0.1, description="KL coefficient at the last step")
    entropy_bonus: float = Field(default=0.0, description="entropy bonus coefficient")
    final_entropy_bonus: float = Field(default=0.0, description="entropy bonus at the last step")
    relu_log_p_weights: bool = Field(default=False, description="clamp the log-prob weights at zero")
    clamp_log_ratio_ref_new_value: float = Field(default=10, description="clamp of log(ref/new)")
    divide_advantage_by_std: bool = Field(default=True, description="normalise advantages by the group std")
    overlong_filtering: bool = Field(default=False, description="zero-weight sequences that overflowed")
    group_normalization: bool = Field(default=False, description="weight tokens by 1/mean group tokens")
    temperature: float = Field(default=1.0, description="sampling temperature of the rollouts")
    filter_zero_advantage_groups: bool = Field(default=False, description="drop all-zero-advantage groups")
    value_loss_coef: float = Field(default=0.0, description="weight of the value loss for models with a value head")
    # --- MI355X extensions (absent from the reference; defaults keep its behaviour) ---
    fused_logits_grad: bool = Field(default=True, description="single-pass logits kernel (gradient computed in the forward launch) "
                                    "whenever the logits require a gradient; False = K1 forward, K2+K3, K1 backward as separate launches")
    inplace_logits_grad: bool = Field(default=False, description="write d loss/d logits over the logits buffer")
    expected_loss_scale: float = Field(default=1.0, description="factor the caller applies to the returned loss before backward "
                                       "(1/gradient_accumulation under accelerate, a loss scaler): the fused kernel folds it into d logits "
                                       "in the forward launch; any other factor is repaired on device in backward, without a host sync")
    skip_unlabelled_rows: bool = Field(default=False, description="fused logits kernel only: do not READ logits rows whose next token "
                                       "carries no label (prompt / observation tokens, sequence starts, padding).  Their log-prob, entropy "
                                       "and gradient are zeros and every labelled value is bit-identical, but the reference's assert over "
                                       "ALL positions (rl/__init__.py:213) can then no longer see a non-finite value in such a row - so the "
                                       "drop-in default is False; the native loops (HotPathStep, NativeLearnerStep, fused_head_loss) turn it on")
    fused_head_keep_logits: bool | None = Field(default=None, description="`rl_step_fused_head` on a bare model: None = keep the forward's fp32 "
                                                "logits for the backward while two copies of them fit in free device memory (2 plane products less), "
                                                "False = never (no logits anywhere), True = as None")
    fused_head_chunk_rows: int = Field(default=8192, description="`rl_step_fused_head` on a bare model: rows of d-logits planes the backward holds at a "
                                       "time (2 x rows x vocab x 2 bytes of workspace)")


def make_rl_data_callback(args: Any, current_dir: Any, rl_config: "RLConfig | None", model: Any):
    """`populate_rl_data` bound to a config, or None without one (reference rl/__init__.py:108-116)."""
    from functools import partial

    return partial(populate_rl_data, config=rl_config) if rl_config else None


def linear_decay_coef(current_step: int, max_step: int, initial_coef: float, final_coef: float) -> float:
    """initial -> final, linearly in current_step / max_step (reference :119-133)."""
    return initial_coef + (final_coef - initial_coef) * current_step / max_step


_POLICY = {"ppo": _lib.PRL_POLICY_PPO, "reinforce": _lib.PRL_POLICY_REINFORCE, "gspo": _lib.PRL_POLICY_GSPO}


def make_loss_config(config: RLConfig, current_step: int, max_step: int) -> tuple[PrlLossConfig, float, float]:
    """RLConfig -> the C struct.  Scalars are rounded to fp32 exactly where torch would do it:
    python doubles are combined in double first (1 - eps, linear decay), then cast."""
    if config.policy_loss not in _POLICY:
        raise ValueError(f"Unknown algorithm {config.policy_loss}")
    kl_coef = linear_decay_coef(current_step, max_step, config.kl_coef, config.final_kl_coef)
    ent_coef = linear_decay_coef(current_step, max_step, config.entropy_bonus, config.final_entropy_bonus)
    use_entropy = config.entropy_bonus != 0.0 or config.final_entropy_bonus != 0.0
    if config.group_normalization:
        token_weight = 0.0
    else:
        token_weight = float(np.float32(1.0) / np.float32(config.batch_size)) if config.batch_size else float("inf")
    c = PrlLossConfig(
        policy_loss=_POLICY[config.policy_loss],
        use_advantages=int(config.use_advantages),
        relu_log_p_weights=int(config.relu_log_p_weights),
        group_normalization=int(config.group_normalization),
        overlong_filtering=int(config.overlong_filtering),
        use_entropy_loss=int(use_entropy),
        token_weight=token_weight,
        clip_lo=1 - config.epsilon_low,
        clip_hi=1 + config.epsilon_high,
        kl_coef=kl_coef,
        entropy_coef=ent_coef,
        clamp_log_ratio_ref_new=config.clamp_log_ratio_ref_new_value,
    )
    return c, kl_coef, ent_coef


def _logits_dtype_code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return _lib.PRL_DTYPE_F32
    if t.dtype == torch.bfloat16:
        return _lib.PRL_DTYPE_BF16
    raise TypeError(f"logits must be float32 or bfloat16, got {t.dtype}")


_workspaces: dict[Any, torch.Tensor] = {}


def _loss_workspace(device: torch.device) -> torch.Tensor:
    """Scratch for the per-block partial records of K2+K3: one per (device, stream), because two
    streams may run the loss kernel concurrently and the finalize kernel reads what the partial
    kernel of the SAME launch wrote."""
    key = (device, _lib.current_stream_ptr(device))
    ws = _workspaces.get(key)
    if ws is None:
        need = ctypes.c_size_t(0)
        _lib.check(_lib.load().prl_grpo_loss_workspace_bytes(1, 1, ctypes.byref(need)))
        ws = torch.empty(need.value, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def grpo_loss_from_logprobs(
    cfg: PrlLossConfig,
    batch: PipelineBatchEncoding,
    new_logprobs: torch.Tensor,
    entropy: torch.Tensor,
    want_grad: bool = True,
    ext_token_grad: torch.Tensor | None = None,
    ext_clamp_indicator: torch.Tensor | None = None,
) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor | None, torch.Tensor | None]:
    """K2+K3 on token-aligned new_logprobs / entropy ([B, L], column 0 unused).
    Returns (loss fp32 scalar, stats double[32], d loss/d new_logprobs, d loss/d entropy).
    `ext_*`: GSPO only, the per-token view of the segment-level gradient / clip indicator."""
    lib = _lib.load()
    rows, cols = batch.labels.shape
    cols_tensors = [
        batch.labels, new_logprobs, entropy, batch.old_logprobs, batch.ref_logprobs, batch.advantages,
        batch.rewards, batch.group_tokens, batch.num_labels, batch.overflow,
    ]
    _lib.require_device(*cols_tensors)
    cols_tensors = [t if t.is_contiguous() else t.contiguous() for t in cols_tensors]
    labels, nlp, ent, old, ref, adv, rew, gt, nl, ovf = cols_tensors
    pos = None
    if batch.is_packed and rows == 1 and batch.position_ids is not None:
        pos = batch.position_ids if batch.position_ids.is_contiguous() else batch.position_ids.contiguous()
    dev = labels.device
    stats = torch.empty(_lib.PRL_NUM_STATS, dtype=torch.float64, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    g_nlp = torch.empty_like(nlp) if want_grad else None
    g_ent = torch.empty_like(nlp) if (want_grad and cfg.use_entropy_loss) else None
    ws = _loss_workspace(dev)
    with torch.cuda.device(dev):
        _lib.check(
            lib.prl_grpo_loss_fwd_bwd(
                ctypes.byref(cfg), rows, cols, _lib.ptr(labels), _lib.ptr(pos), _lib.ptr(nlp), _lib.ptr(ent),
                _lib.ptr(old), _lib.ptr(ref), _lib.ptr(adv), _lib.ptr(rew), _lib.ptr(gt), _lib.ptr(nl),
                _lib.ptr(ovf), _lib.ptr(ext_token_grad), _lib.ptr(ext_clamp_indicator), _lib.ptr(g_nlp), _lib.ptr(g_ent),
                _lib.ptr(loss), _lib.ptr(stats),
                _lib.ptr(ws), ws.numel(), _lib.current_stream_ptr(dev),
            )
        )
    return loss, stats, g_nlp, g_ent


def logprob_entropy(logits: torch.Tensor, input_ids: torch.Tensor, temperature: float):
    """K1 forward: token-aligned (new_logprobs, entropy, lse2), each float32 [B, L]."""
    lib = _lib.load()
    _lib.require_device(logits, input_ids)
    B, L, V = logits.shape
    if logits.stride(-1) != 1 or logits.stride(1) != logits.shape[-1] or (B > 1 and logits.stride(0) != L * V):
        logits = logits.contiguous()
    ids = input_ids if input_ids.is_contiguous() else input_ids.contiguous()
    nlp = torch.empty((B, L), dtype=torch.float32, device=logits.device)
    ent = torch.empty_like(nlp)
    lse2 = torch.empty_like(nlp)
    with torch.cuda.device(logits.device):
        _lib.check(
            lib.prl_logprob_entropy_fwd(
                B, L, V, _lib.ptr(logits), _logits_dtype_code(logits), V, _lib.ptr(ids), float(temperature),
                _lib.ptr(nlp), _lib.ptr(ent), _lib.ptr(lse2), _lib.current_stream_ptr(logits.device),
            )
        )
    return nlp, ent, lse2, logits


def segment_sums(segment_ids: torch.Tensor, labels: torch.Tensor, a: torch.Tensor, b: torch.Tensor, n_segments: int):
    """Masked per-segment sums of two token-aligned columns + token counts (float64 [n_segments] x 3)."""
    lib = _lib.load()
    dev = a.device
    out = [torch.empty(n_segments, dtype=torch.float64, device=dev) for _ in range(3)]
    cont = lambda t: t if t.is_contiguous() else t.contiguous()  # noqa: E731
    with torch.cuda.device(dev):
        _lib.check(lib.prl_segment_sums(a.shape[-1], n_segments, _lib.ptr(cont(segment_ids)), _lib.ptr(cont(labels)),
                                        _lib.ptr(cont(a)), _lib.ptr(cont(b)), _lib.ptr(out[0]), _lib.ptr(out[1]), _lib.ptr(out[2]),
                                        _lib.current_stream_ptr(dev)))
    return out


def _group_all_reduce(t: torch.Tensor, group: Any, op: Any) -> torch.Tensor:
    """All-reduce a small device tensor over `group`; a gloo group gets it through host memory."""
    import torch.distributed as dist

    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        return h.to(t.device)
    dist.all_reduce(t, op=op, group=group)
    return t


def gspo_segment_terms(cfg: PrlLossConfig, batch: PipelineBatchEncoding, new_logprobs: torch.Tensor, seq_parallel_group: Any = None):
    """Sequence-level (GSPO) policy term, reference rl/__init__.py:310-352 + rl/utils.py:106-208:
    per-segment masked means of log(new/old) and of the advantages (segment-sum kernel), clipped
    sequence ratio, loss = -sum_s min(r_s A_s, clip(r_s) A_s) * (sum of the segment's token weights).
    Returns (loss scalar, per-token d loss/d new_logprobs coefficient, per-token clip indicator).  Three launches: the four
    per-segment sums (`prl_gspo_segment_sums`), the O(#segments) arithmetic (`prl_gspo_segment_terms`), the way back to the
    tokens (`prl_gspo_expand`).

    With `seq_parallel_group` the batch is one `make_slices` slice of a packed sequence: the four
    per-segment sums are added over the group in ONE all-reduce (the reference issues one per
    column, rl/utils.py:194-206), every rank then holds the full loss, and - as with the reference's
    differentiable all-reduce, whose backward sums the identical gradients of all ranks - the token
    gradient carries a factor `group size`."""
    import torch.distributed as dist

    sp = seq_parallel_group is not None and dist.is_available() and dist.is_initialized()
    seg_ids = batch.segment_ids
    if seg_ids is None:
        raise ValueError("segment_ids must be provided for per-segment reductions")
    if batch.seq_boundaries is not None:
        n_seg = int(batch.seq_boundaries.shape[0]) - 1  # slices carry the boundaries of the whole sequence
    else:
        top = seg_ids.max().to(torch.int64).reshape(1) if seg_ids.numel() else torch.full((1,), -1, dtype=torch.int64, device=seg_ids.device)
        if sp:
            top = _group_all_reduce(top, seq_parallel_group, dist.ReduceOp.MAX)
        n_seg = int(top.item()) + 1
    f32 = torch.float32
    # the four per-segment sums in ONE launch: new - old and the token weight are formed in the kernel's loop, not as [1, T] tensors
    lib = _lib.load()
    dev = new_logprobs.device
    cont = lambda t: t if t.is_contiguous() else t.contiguous()  # noqa: E731
    sums = torch.empty((4, max(n_seg, 0)), dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        _lib.check(lib.prl_gspo_segment_sums(ctypes.byref(cfg), new_logprobs.shape[-1], n_seg, _lib.ptr(cont(seg_ids)), _lib.ptr(cont(batch.labels)),
                                             _lib.ptr(cont(new_logprobs)), _lib.ptr(cont(batch.old_logprobs)), _lib.ptr(cont(batch.advantages)),
                                             _lib.ptr(cont(batch.group_tokens)), _lib.ptr(cont(batch.overflow)), _lib.ptr(sums),
                                             _lib.current_stream_ptr(dev)))
    grad_scale = 1.0
    if sp:
        sums = _group_all_reduce(sums, seq_parallel_group, dist.ReduceOp.SUM)
        grad_scale = float(dist.get_world_size(seq_parallel_group))
    # the O(#segments) arithmetic - clipped sequence ratio, loss, per-segment gradient coefficient, clip indicator - in one small launch
    coef = torch.empty(max(n_seg, 0), dtype=f32, device=dev)
    indicator = torch.empty_like(coef)
    loss = torch.empty((), dtype=f32, device=dev)
    T = new_logprobs.shape[-1]
    with torch.cuda.device(dev):
        _lib.check(lib.prl_gspo_segment_terms(ctypes.byref(cfg), n_seg, _lib.ptr(sums.contiguous()), float(grad_scale), 1 if (batch.sentinel or T <= 1) else 0,
                                              _lib.ptr(coef), _lib.ptr(indicator), _lib.ptr(loss), _lib.current_stream_ptr(dev)))
    # back to the tokens in one launch: the segment's coefficient, and the clip indicator of the j-th sequence STARTING OR CONTINUING in
    # this slice (the reference zips local segments with per-segment values, rl/__init__.py:347-350; identical to indexing by segment
    # id when the slice starts at segment 0)
    ext_g = torch.empty(new_logprobs.shape, dtype=f32, device=dev)
    ext_c = torch.empty_like(ext_g)
    if n_seg > 0 and ext_g.numel():
        with torch.cuda.device(dev):
            _lib.check(lib.prl_gspo_expand(ext_g.shape[-1], n_seg, _lib.ptr(cont(seg_ids)), _lib.ptr(coef), _lib.ptr(indicator),
                                           _lib.ptr(ext_g), _lib.ptr(ext_c), _lib.current_stream_ptr(dev)))
    else:
        ext_g.zero_()
        ext_c.zero_()
    return loss, ext_g, ext_c


def annotate_ref_logprobs(ref_model: Any, batch: PipelineBatchEncoding, temperature: float = 1.0, fused_head: bool = True) -> PipelineBatchEncoding:
    """Fill `batch.ref_logprobs` with a no-grad forward of a frozen reference model that lives on
    the learner GPU.  The reference pipeline instead asks a second vLLM server
    for `prompt_logprobs` over HTTP per chunk (preprocess.py:86-104, llm.py:606-648) and stores
    the completion-token logprobs left-zero-padded:
    ref_logprobs[u] = log p_ref(token u | prefix) on labelled tokens, 0 elsewhere.

    A model in the Hugging Face layout (`.model` body + bias-free `.lm_head`) is asked for its last HIDDEN STATES and
    the output head runs on the MFMA kernels of csrc/prl_lmhead.hip (`fused_head.token_logprobs_from_hidden`): the
    `[T, V]` reference logits - 4.98 GB fp32 per 8192-token micro-batch at V = 152 064 - are never written, and only the
    rows that predict a labelled token enter the product.  Anything else (a callable that only returns `.logits`), or
    `fused_head=False`, goes through the logits and the K1 kernel."""
    if fused_head:
        from ...fused_head import annotate_ref_logprobs_fused

        if annotate_ref_logprobs_fused(ref_model, batch, temperature):
            return batch
    model_inputs = {"input_ids": batch.input_ids, "attention_mask": batch.attention_mask}
    if batch.is_packed:
        model_inputs["position_ids"] = batch.position_ids
    with torch.no_grad():
        logits = ref_model(**model_inputs).logits
        nlp, _, _, _ = logprob_entropy(logits, batch.input_ids, temperature)
        batch.ref_logprobs = torch.where(batch.labels != -100, nlp, torch.zeros_like(nlp))
    return batch


class _GrpoLossFn(torch.autograd.Function):
    """logits -> (loss, stats) with a hand-written backward to the logits."""

    @staticmethod
    def forward(ctx, logits, batch, cfg, temperature, fused, inplace, sp_group=None, expected_scale=1.0, skip_unlabelled=False):  # type: ignore[override]
        lib = _lib.load()
        B, L, V = logits.shape
        dev = logits.device
        ids = batch.input_ids if batch.input_ids.is_contiguous() else batch.input_ids.contiguous()
        if fused:
            lg = logits if logits.is_contiguous() else logits.contiguous()
            nlp = torch.empty((B, L), dtype=torch.float32, device=dev)
            ent = torch.empty_like(nlp)
            if batch.sentinel:
                # a sentinel batch has no labelled token (finetune/utils.py:17-78): loss 0, gradient 0.
                # Nothing of the [T, V] logits needs to be read; backward hands out zeros.
                nlp.zero_()
                ent.zero_()
                grad = None
            else:
                grad = lg if inplace else torch.empty_like(lg)
                lse2 = torch.empty_like(nlp)
                kcfg = type(cfg).from_buffer_copy(cfg)
                kcfg.upstream_scale = float(expected_scale)
                # opt-in (RLConfig.skip_unlabelled_rows): rows that predict an unlabelled token reach neither the loss nor a
                # statistic and need not be read - but then their finiteness is not checked either (reference :213 checks it)
                kcfg.skip_unlabelled = 1 if skip_unlabelled else 0
                cont = lambda t: t if t.is_contiguous() else t.contiguous()  # noqa: E731
                with torch.cuda.device(dev):
                    _lib.check(
                        lib.prl_fused_logits_loss(
                            ctypes.byref(kcfg), B, L, V, _lib.ptr(lg), _logits_dtype_code(lg), V, float(temperature),
                            _lib.ptr(ids), _lib.ptr(cont(batch.labels)), _lib.ptr(cont(batch.old_logprobs)),
                            _lib.ptr(cont(batch.ref_logprobs)), _lib.ptr(cont(batch.advantages)),
                            _lib.ptr(cont(batch.rewards)), _lib.ptr(cont(batch.group_tokens)),
                            _lib.ptr(cont(batch.overflow)), _lib.ptr(nlp), _lib.ptr(ent), _lib.ptr(lse2),
                            _lib.ptr(grad), _lib.current_stream_ptr(dev),
                        )
                    )
            loss, stats, _, _ = grpo_loss_from_logprobs(cfg, batch, nlp, ent, want_grad=False)
            ctx.fused = True
            ctx.grad_logits = grad
            ctx.zero_like = lg if grad is None else None
            ctx.inplace = inplace
            ctx.expected_scale = float(expected_scale)
        else:
            nlp, ent, lse2, lg = logprob_entropy(logits, ids, temperature)
            if cfg.policy_loss == _lib.PRL_POLICY_GSPO:
                seg_loss, ext_g, ext_c = gspo_segment_terms(cfg, batch, nlp, sp_group)
                _, stats, g_nlp, g_ent = grpo_loss_from_logprobs(cfg, batch, nlp, ent, want_grad=True,
                                                                  ext_token_grad=ext_g, ext_clamp_indicator=ext_c)
                loss = seg_loss
                stats[STAT_INDEX["loss"]] = seg_loss.double()
            else:
                loss, stats, g_nlp, g_ent = grpo_loss_from_logprobs(cfg, batch, nlp, ent, want_grad=True)
            ctx.fused = False
            ctx.save_for_backward(lg, ids, lse2, ent, g_nlp, g_ent if g_ent is not None else torch.empty(0, device=dev))
            ctx.has_g_ent = g_ent is not None
            ctx.inplace = inplace
        ctx.temperature = float(temperature)
        ctx.mark_non_differentiable(stats)
        return loss, stats

    @staticmethod
    def backward(ctx, grad_loss, _grad_stats):  # type: ignore[override]
        if ctx.fused:
            grad = ctx.grad_logits
            ctx.grad_logits = None
            if grad is None:  # sentinel batch
                lg = ctx.zero_like
                ctx.zero_like = None
                grad = lg.zero_() if ctx.inplace else torch.zeros_like(lg)
                return grad, None, None, None, None, None, None, None, None
            # d logits already carries `expected_scale`; any other upstream factor is applied by a
            # kernel that returns after one scalar load when the guess was right (no host sync).
            up = grad_loss.to(torch.float32).contiguous()
            with torch.cuda.device(grad.device):
                _lib.check(_lib.load().prl_scale_unless(_lib.ptr(grad), grad.numel(), _logits_dtype_code(grad), _lib.ptr(up),
                                                        ctx.expected_scale, _lib.current_stream_ptr(grad.device)))
            return grad, None, None, None, None, None, None, None, None
        lib = _lib.load()
        lg, ids, lse2, ent, g_nlp, g_ent = ctx.saved_tensors
        B, L, V = lg.shape
        dev = lg.device
        grad = lg if ctx.inplace else torch.empty_like(lg)
        up = grad_loss.to(torch.float32).contiguous()
        with torch.cuda.device(dev):
            _lib.check(
                lib.prl_logprob_entropy_bwd(
                    B, L, V, _lib.ptr(lg), _logits_dtype_code(lg), V, _lib.ptr(ids), ctx.temperature,
                    _lib.ptr(lse2), _lib.ptr(ent), _lib.ptr(g_nlp), _lib.ptr(g_ent) if ctx.has_g_ent else None,
                    _lib.ptr(up), _lib.ptr(grad), _lib.current_stream_ptr(dev),
                )
            )
        return grad, None, None, None, None, None, None, None, None


_STAT_KEYS_IN_ORDER = [
    "reward", "max_reward", "min_reward", "entropy", "old_logprobs", "new_logprobs", "ref_logprobs",
    "advantage", "max_advantage", "min_advantage", "kl", "kl_new_old", "mean_abs_log_ratio_new_old",
    "max_kl", "min_kl", "ratio_new_old", "ratio_new_old_sum", "ratio_new_old_squared_sum", "ratio_ref_new",
    "ratio_ref_old", "clamp_log_ratio_ref_new_indicator", "clamp_log_ratio_new_old_indicator",
    "token_weight", "max_token_weight", "min_token_weight",
]


def stats_to_dict(stats: Sequence[float], kl_coef: float, ent_coef: float, input_size: int) -> dict[str, float]:
    """Device stats vector -> the reference's 32-key dict (rl/__init__.py:398-439), same key order."""
    s = stats
    loss = float(np.float32(s[STAT_INDEX["loss"]]))
    out: dict[str, float] = {"loss": loss, "max_loss": loss, "min_loss": loss}
    for k in _STAT_KEYS_IN_ORDER:
        out[k] = float(np.float32(s[STAT_INDEX[k]]))
    n_seq = int(s[STAT_INDEX["num_sequences"]])
    out["kl_coef"] = n_seq * kl_coef
    out["entropy_bonus_coef"] = n_seq * ent_coef
    out["num_output_tokens_sum"] = int(s[STAT_INDEX["num_output_tokens_sum"]])
    out["input_size"] = input_size
    return out


VALUE_STAT_KEYS = ("value_mean", "value_max", "value_min", "value_loss", "value_mse")  # rl/__init__.py:441-448, in this order


def value_head_terms(cfg: PrlLossConfig, batch: PipelineBatchEncoding, values: torch.Tensor, want_grad: bool = True):
    """The value-head branch of rl_step in one launch (csrc/prl_value.hip; reference rl/__init__.py:265-272, 367-381,
    441-448).  `values`: outputs.value [B, L], float32 or bfloat16.  Returns (value_loss fp32 scalar, advantages fp32
    [B, L] unshifted like batch.advantages = rewards - V one column to the left, stats double[5] in VALUE_STAT_KEYS order,
    d value_loss / d values fp32 [B, L] or None)."""
    lib = _lib.load()
    rows, cols = batch.labels.shape
    if tuple(values.shape) != (rows, cols):
        raise ValueError(f"Values shape {tuple(values.shape)} does not match the batch shape {(rows, cols)}")
    tensors = [batch.labels, values.detach(), batch.rewards, batch.group_tokens, batch.num_labels, batch.overflow]
    _lib.require_device(*tensors)
    labels, val, rew, gt, nl, ovf = [t if t.is_contiguous() else t.contiguous() for t in tensors]
    dev = labels.device
    adv = torch.empty((rows, cols), dtype=torch.float32, device=dev)
    g_val = torch.empty((rows, cols), dtype=torch.float32, device=dev) if want_grad else None
    vstats = torch.empty(_lib.PRL_NUM_VALUE_STATS, dtype=torch.float64, device=dev)
    vloss = torch.empty((), dtype=torch.float32, device=dev)
    ws = _loss_workspace(dev)  # 64 KB of the K2+K3 scratch: the two launches are ordered on the stream
    with torch.cuda.device(dev):
        _lib.check(lib.prl_value_head_fwd_bwd(
            ctypes.byref(cfg), rows, cols, _lib.ptr(labels), _lib.ptr(val), _logits_dtype_code(val), _lib.ptr(rew), _lib.ptr(gt),
            _lib.ptr(nl), _lib.ptr(ovf), _lib.ptr(adv), _lib.ptr(g_val), _lib.ptr(vloss), _lib.ptr(vstats),
            _lib.ptr(ws), ws.numel(), _lib.current_stream_ptr(dev)))
    return vloss, adv, vstats, g_val


class _ValueLossFn(torch.autograd.Function):
    """value_loss as a node of the model's graph: forward = value_head_terms, backward = its closed-form gradient."""

    @staticmethod
    def forward(ctx, values: torch.Tensor, batch: PipelineBatchEncoding, cfg: PrlLossConfig):
        want = ctx.needs_input_grad[0]
        vloss, adv, vstats, g_val = value_head_terms(cfg, batch, values, want_grad=want)
        if want:
            ctx.save_for_backward(g_val)
        ctx.values_dtype = values.dtype
        ctx.mark_non_differentiable(adv, vstats)
        return vloss, adv, vstats

    @staticmethod
    def backward(ctx, grad_loss, _grad_adv, _grad_stats):
        (g_val,) = ctx.saved_tensors
        return (g_val * grad_loss).to(ctx.values_dtype), None, None


def _with_advantages(batch: PipelineBatchEncoding, advantages: torch.Tensor) -> PipelineBatchEncoding:
    """A shallow copy of the batch whose `advantages` column is `advantages` (the caller's batch is left as it was)."""
    import copy

    b = copy.copy(batch)
    object.__setattr__(b, "advantages", advantages)
    return b


def host_stats(stats_dev: torch.Tensor, input_size: int, kl_coef: float, ent_coef: float, value_loss_coef: float = 0.0) -> dict[str, float]:
    """The step's device statistics -> the reference's dict, with its runtime asserts: ONE device -> host copy.  `stats_dev`:
    double[32] of K2+K3, optionally followed by the value head's double[5] (then the reported and asserted loss is the
    combined one, rl/__init__.py:381-386, 399-401, and the five value keys close the dict, :441-448)."""
    stats = stats_dev.cpu().tolist()  # the single device->host sync of the step
    vstats = stats[_lib.PRL_NUM_STATS:]
    if vstats:
        combined = np.float32(stats[STAT_INDEX["loss"]]) + np.float32(value_loss_coef * np.float32(vstats[VALUE_STAT_KEYS.index("value_loss")]))
        stats[STAT_INDEX["loss"]] = float(combined)
    check_finite(stats)
    if int(stats[STAT_INDEX["num_output_tokens_sum"]]) == 0:
        return {"input_size": float(input_size)}
    out = stats_to_dict(stats, kl_coef, ent_coef, input_size)
    out.update({k: float(np.float32(v)) for k, v in zip(VALUE_STAT_KEYS, vstats)})
    return out


def check_finite(stats: Sequence[float]) -> None:
    """The reference's runtime asserts (rl/__init__.py:213,247,262,291,386) from device counters."""
    assert stats[STAT_INDEX["nonfinite_new_logprobs"]] == 0, "new_logprobs is not finite"
    assert stats[STAT_INDEX["bad_group_tokens"]] == 0, "group_tokens must be greater than zero for group normalization"
    assert stats[STAT_INDEX["nonfinite_log_ratio_ref_new"]] == 0, "log_ratio_ref_new is not finite"
    assert stats[STAT_INDEX["nonfinite_kl"]] == 0, "approx_kl is not finite"
    assert np.isfinite(stats[STAT_INDEX["loss"]]), f"Non-finite loss detected: {stats[STAT_INDEX['loss']]}"


def rl_step(
    model: Any,
    batch: PipelineBatchEncoding,
    current_step: int,
    max_step: int,
    config: RLConfig,
    seq_parallel_group=None,
) -> tuple[torch.Tensor, dict[str, float]]:
    """One RL micro-batch: model forward + fused loss.  Signature and return value as in
    reference rl/__init__.py:136-143: (scalar loss attached to the model's graph, stats dict)."""
    if config.policy_loss == "gspo":
        if not batch.is_packed:
            raise ValueError("GSPO loss requires packed sequences with segments")
    has_value_head = hasattr(model, "value_head")  # finetune/value_model.py; reference rl/__init__.py:162
    cfg, kl_coef, ent_coef = make_loss_config(config, current_step, max_step)

    model_inputs = {
        "input_ids": batch.input_ids,
        "attention_mask": batch.attention_mask,
        "labels": batch.labels,
    }
    if batch.is_packed:
        model_inputs["position_ids"] = batch.position_ids
    if getattr(batch, "pixel_values", None) is not None:
        model_inputs["pixel_values"] = batch.pixel_values
    if getattr(batch, "image_grid_thw", None) is not None:
        model_inputs["image_grid_thw"] = batch.image_grid_thw
    outputs = model(**model_inputs)
    logits = outputs.logits
    _lib.require_device(logits)
    value_loss = vstats_dev = None
    if has_value_head:
        # advantages := rewards - V (detached, :272) for the policy loss and its statistics; the value loss joins below
        value_loss, value_advantages, vstats_dev = _ValueLossFn.apply(outputs.value, batch, cfg)
        batch = _with_advantages(batch, value_advantages)

    loss, stats_dev = _GrpoLossFn.apply(
        logits, batch, cfg, config.temperature,
        bool(config.fused_logits_grad) and config.policy_loss != "gspo" and logits.requires_grad and torch.is_grad_enabled(),
        bool(config.inplace_logits_grad), seq_parallel_group if config.policy_loss == "gspo" else None,
        float(config.expected_loss_scale) or 1.0, bool(config.skip_unlabelled_rows),
    )
    if has_value_head:
        loss = loss + config.value_loss_coef * value_loss  # (:381)
        stats_dev = torch.cat([stats_dev, vstats_dev])
    return loss, host_stats(stats_dev, batch.input_ids.numel(), kl_coef, ent_coef, config.value_loss_coef)


# ---------------------------------------------------------------------------------------------
# Preprocess: group-baseline advantages (K5)
# ---------------------------------------------------------------------------------------------


@dataclass
class PreparedRollouts:
    """Rollouts + the per-sequence scalars that populate_rl_data attaches (all on device)."""

    rollouts: RaggedRollouts
    reward32: torch.Tensor       # fp32 [S]   -> `rewards` column
    advantage: torch.Tensor      # fp32 [S]
    group_tokens: torch.Tensor   # fp32 [S]
    num_labels: torch.Tensor     # fp32 [S]
    overflow: torch.Tensor       # fp32 [S]
    advantage64: torch.Tensor    # fp64 [S]   (what the reference's python lists hold)
    group_tokens64: torch.Tensor  # fp64 [S]
    k5_out32: torch.Tensor | None = None  # fp32 [4, S]: the allocation behind num_labels / overflow / advantage / group_tokens (one copy takes all four to the host)


def _csr(keys: np.ndarray) -> tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(offsets, members in dataset order, key id per member) for equal-key runs of `keys`."""
    order = np.argsort(keys, kind="stable").astype(np.int32)
    sorted_keys = keys[order]
    if len(keys) == 0:
        return np.zeros(1, dtype=np.int32), order, np.zeros(0, dtype=np.int64)
    starts = np.flatnonzero(np.r_[True, sorted_keys[1:] != sorted_keys[:-1]])
    off = np.r_[starts, len(keys)].astype(np.int32)
    return off, order, sorted_keys[starts]


_NP_TORCH = {"int64": torch.int64, "int32": torch.int32, "float32": torch.float32, "float64": torch.float64, "uint8": torch.uint8}


def upload_packed(arrays: Sequence[np.ndarray], device) -> list[torch.Tensor]:
    """Several small host arrays -> device tensors of the same dtype and shape, views into ONE allocation filled by ONE copy
    (8-byte aligned; a planning table of a few KB costs a copy's fixed latency, not its bandwidth)."""
    host = [np.ascontiguousarray(a) for a in arrays]
    offs, total = [], 0
    for a in host:
        total += (-total) % 8
        offs.append(total)
        total += a.nbytes
    packed = np.empty(total, dtype=np.uint8)
    for a, o in zip(host, offs):
        packed[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    on_dev = torch.from_numpy(packed).to(device, non_blocking=True)
    return [on_dev[o:o + a.nbytes].view(_NP_TORCH[a.dtype.name]).view(a.shape) for a, o in zip(host, offs)]


def plan_groups(group_index: np.ndarray, step_index: np.ndarray, rollout_index: np.ndarray):
    """Host-side O(S) planning for K5: CSR membership of (group, step) keys and of groups, plus
    the number of distinct rollouts per group (reference groupby keys, rl/__init__.py:464-486)."""
    g = group_index.astype(np.int64)
    n_steps = int(step_index.max()) + 1 if len(step_index) else 1
    group_off, group_members, group_keys = _csr(g)
    if n_steps == 1:  # single-step rollouts (every BASELINE config): the (group, step) keys ARE the groups
        key_off, key_members = group_off, group_members
    else:
        key_off, key_members, _ = _csr(g * n_steps + step_index.astype(np.int64))
    n_roll = int(rollout_index.max()) + 1 if len(rollout_index) else 1
    pairs = np.unique(g * n_roll + rollout_index.astype(np.int64))
    pair_group = pairs // n_roll
    # groups are sorted ascending in both `group_keys` and `pair_group`
    counts = np.searchsorted(pair_group, group_keys, side="right") - np.searchsorted(pair_group, group_keys, side="left")
    return key_off, key_members, group_off, group_members, counts.astype(np.int32)


def populate_rl_data_ragged(rollouts: RaggedRollouts, eos_token_id: int, config: RLConfig, plan: Sequence[torch.Tensor] | None = None,
                            timer: Any = None) -> PreparedRollouts:
    """K5 on device: num_labels / overflow per sequence, leave-one-out advantages per
    (group_id, step_index), mean rollout tokens per group (reference rl/__init__.py:453-570).
    `plan`: the five arrays of `plan_groups` already on the device (they rode along with the rollouts' upload,
    `RaggedRollouts.to(device, stager, extra=plan_groups(...))`); None = planned and uploaded here."""
    lib = _lib.load()
    r = rollouts
    _lib.require_device(r.tokens)
    dev = r.device
    S = r.n_seqs
    # the six per-sequence outputs live in one allocation (8-byte columns first)
    out64 = torch.empty((2, S), dtype=torch.float64, device=dev)
    out32 = torch.empty((4, S), dtype=torch.float32, device=dev)
    adv64, gt64 = out64[0], out64[1]
    num_labels, overflow, adv32, gt32 = out32[0], out32[1], out32[2], out32[3]
    import contextlib

    # `timer` (bench.py's EventTimer): HIP events around each launch alone, next to the all-in figure of the caller
    with torch.cuda.device(dev):
        stream = _lib.current_stream_ptr(dev)
        # the scan needs no plan: it goes first, and the O(S) host planning below + its ONE upload overlap with it
        with (timer.time("group_advantages_K5_kernels") if timer is not None else contextlib.nullcontext()):
            _lib.check(
                lib.prl_seq_scan(
                    S, _lib.ptr(r.tokens), _lib.ptr(r.labels), _lib.ptr(r.seq_off), _lib.ptr(r.finish_code),
                    _lib.ptr(r.finished), int(eos_token_id), _lib.ptr(num_labels), _lib.ptr(overflow), stream,
                )
            )
        if plan is None:
            plan = upload_packed(plan_groups(r.host_group_index, r.host_step_index, r.host_rollout_index), dev)  # five small arrays, one copy
        key_off, group_off = plan[0], plan[2]
        with (timer.time("group_advantages_K5_group_launch") if timer is not None else contextlib.nullcontext()):
            _lib.check(
                lib.prl_group_advantages(
                    S, len(key_off) - 1, len(group_off) - 1, *[_lib.ptr(p) for p in plan], _lib.ptr(r.reward),
                    _lib.ptr(r.seq_off), int(config.divide_advantage_by_std), _lib.ptr(adv64), _lib.ptr(gt64),
                    _lib.ptr(adv32), _lib.ptr(gt32), stream,
                )
            )
    return PreparedRollouts(
        rollouts=r, reward32=r.reward.to(torch.float32), advantage=adv32, group_tokens=gt32,
        num_labels=num_labels, overflow=overflow, advantage64=adv64, group_tokens64=gt64, k5_out32=out32,
    )


def populate_rl_data(dataset: list[dict[str, Any]], eos_token_id: int, config: RLConfig) -> list[dict[str, Any]]:
    """List-of-dicts front end with the reference's contract (rl/__init__.py:453): fills the
    per-token `advantages`, `group_tokens`, `overflow`, `num_labels` lists of every entry in place.
    Entries are what `preprocess_fn(..., is_rl=True)` produced (+ group_id, rollout_index,
    step_index).  The numbers come from the device kernels."""
    if not dataset:
        # the reference builds a DataFrame and selects its columns (rl/__init__.py:456-459): on an empty list pandas raises this
        raise KeyError("None of [Index(['group_id', 'rollout_index', 'step_index', 'rewards'], dtype='object')] are in the [columns]")
    entries = []
    for e in dataset:
        if len(e["rewards"]) == 0:
            raise IndexError("populate_rl_data: empty sequence (the reference raises on rewards[0])")
        if any(r != e["rewards"][0] for r in e["rewards"]):
            raise NotImplementedError("per-token varying rewards inside one sequence are not produced on the RL path")
        n_lp = len(e.get("logprobs", ()))
        entries.append({
            "input_ids": e["input_ids"], "labels": e["labels"], "logprobs": e.get("logprobs", [])[:n_lp],
            "reward": e["rewards"][0], "group_id": e["group_id"], "rollout_index": e["rollout_index"],
            "step_index": e["step_index"], "model_version": e.get("model_version", 0),
            "finished": e.get("finished"), "finish_reason": e.get("finish_reason"),
        })
    rag = RaggedRollouts.from_entries(entries).to(_default_device())
    prep = populate_rl_data_ragged(rag, eos_token_id, config)
    adv = prep.advantage64.cpu().numpy()
    gt = prep.group_tokens64.cpu().numpy()
    ovf = prep.overflow.cpu().numpy()
    nl = prep.num_labels.cpu().numpy()
    for i, e in enumerate(dataset):
        n = len(e["input_ids"])
        e["advantages"] = [float(adv[i])] * len(e["rewards"])
        e["group_tokens"] = [float(gt[i])] * n
        e["overflow"] = [float(ovf[i])] * len(e["overflow"])
        e["num_labels"] = [int(nl[i])] * n
    return dataset


def _default_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("pipelinerl_amd preprocess kernels need a HIP device; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def prepare_rl_fields(
    encoding: dict[str, Any],
    reward: float,
    old_logprobs: list[float],
    ref_logprobs: list[float],
) -> dict[str, Any]:
    """Per-token RL columns of one sample as python lists (reference rl/__init__.py:573-594).
    Host-side compatibility helper: the device path never materialises these lists (the pack
    kernel expands per-sequence scalars while writing the batch)."""
    labels = encoding["labels"]
    n = len(labels)
    n_targets = n - labels.count(-100)
    assert n_targets == len(old_logprobs), f"Target tokens: {n_targets}, old logprobs: {len(old_logprobs)}"
    zeros = [0] * n
    encoding.update(
        rewards=[reward] * n,
        advantages=[0.0] * n,
        old_logprobs=zeros[: n - len(old_logprobs)] + old_logprobs,
        ref_logprobs=zeros[: n - len(ref_logprobs)] + ref_logprobs,
        overflow=list(zeros),
        group_tokens=list(zeros),
        num_labels=[int(lab != -100) for lab in labels],
    )
    return encoding
