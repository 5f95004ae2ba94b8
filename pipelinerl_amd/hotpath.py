"""The learner hot path for one optimizer step, MI355X style:

    rollouts (ragged SoA, HBM) --K5--> advantages --K6 (ONE launch)--> all micro-batches of the step
    per micro-batch:  model -> logits --fused K1+grad+K1'--> d loss/d logits ; new_logprobs/entropy
                                                              appended to a step-long buffer
    per step:         ONE K2+K3 launch over every token of the step -> loss + 32 stats
                      ONE all-gather of the stats vector across data-parallel ranks

Compared with the reference loop (finetune_loop.py:647-957) there is no per-micro-batch host
sync (`.item()` x31), no per-micro-batch all_gather of the sample counter (the schedule is
static within a step) and the statistics kernel runs once over ~2 GB instead of 4096 times over
0.5 MB.  `rl_step` (finetune/rl) keeps the reference's per-micro-batch contract for drop-in use;
this class is what `LearnerStep` and `bench.py` drive.
"""

from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Any, Sequence

import torch

from . import _lib
from .finetune.data import pack_prepared
from .finetune.rl import (
    PreparedRollouts,
    RLConfig,
    _logits_dtype_code,
    check_finite,
    grpo_loss_from_logprobs,
    make_loss_config,
    populate_rl_data_ragged,
    stats_to_dict,
)
from .finetune.types import PipelineBatchEncoding
from .ragged import RaggedRollouts

# lanes of the stats vector that reduce with max / min across ranks (everything else adds)
_MAX_LANES = [_lib.STAT_INDEX[k] for k in ("max_reward", "max_advantage", "max_kl", "max_token_weight")]
_MIN_LANES = [_lib.STAT_INDEX[k] for k in ("min_reward", "min_advantage", "min_kl", "min_token_weight")]


@dataclass
class StepBuffers:
    """Step-long token-aligned buffers that the per-micro-batch logits kernel fills."""

    new_logprobs: torch.Tensor  # fp32 [1, T_step]
    entropy: torch.Tensor
    lse2: torch.Tensor


class HotPathStep:
    """Drives K5 -> K6 -> (per micro-batch logits kernel) -> one K2+K3 for a step's rollouts."""

    def __init__(self, config: RLConfig, eos_token_id: int, current_step: int = 0, max_step: int = 1,
                 group: Any = None, skip_unlabelled: bool = True):
        """`skip_unlabelled` (default): the logits kernel does not READ rows whose next token carries no label (prompt and
        observation tokens, sequence starts, padding) - they reach neither the loss nor a statistic (rl/__init__.py:238-250),
        so nothing changes numerically and ~3-30 % of the logits traffic goes away; the price is that a non-finite logit in
        such a row passes unnoticed, where the reference asserts `isfinite(new_logprobs)` over EVERY position
        (rl/__init__.py:213).  `False` keeps that assert: every row is read, `stats_dict()` raises for a NaN / inf anywhere."""
        if config.policy_loss == "gspo":
            # the sequence-level term needs every token's log-prob of a sequence before any token's gradient exists (rl/__init__.py:310-352):
            # it cannot live in the one-pass logits kernel this step is built around
            raise ValueError("HotPathStep / NativeLearnerStep fuse the per-token losses (ppo, reinforce); gspo is served by rl_step and "
                             "rl_step_fused_head (LearnerStep, StreamedLearnerStep)")
        self.config = config
        self.eos_token_id = eos_token_id
        self.cfg, self.kl_coef, self.ent_coef = make_loss_config(config, current_step, max_step)
        self.cfg.skip_unlabelled = 1 if skip_unlabelled else 0
        self.group = group
        self.batches: Any = []
        self.step_batch: PipelineBatchEncoding | None = None
        self.buffers: StepBuffers | None = None
        self.offsets: list[int] = []

    # -- preprocess ---------------------------------------------------------------------------
    def preprocess(self, rollouts: RaggedRollouts, micro_batches: Sequence[Sequence[int]], timer: Any = None):
        """K5 + one K6 launch.  The returned micro-batches are views into one flat step batch.
        `timer`: optional event timer (bench.py) for the two device parts alone."""
        import contextlib

        # (the planning tables go up as they are: at step scale - five + three small arrays - the page-locked ring of the
        # preprocessor loop, `staging.PinnedStager`, measured 10-50 us SLOWER than the plain copies; profiles/r04zy vs r04z)
        with (timer.time("group_advantages_K5") if timer is not None else contextlib.nullcontext()):
            prep: PreparedRollouts = populate_rl_data_ragged(rollouts, self.eos_token_id, self.config, timer=timer)
        self.batches = pack_prepared(prep, micro_batches, self.eos_token_id, timer=timer)
        self.offsets = [int(x) for x in self.batches.token_off]
        total = self.offsets[-1]
        dev = rollouts.device
        self.step_batch = self.batches.step_batch()
        z = lambda: torch.zeros((1, total), dtype=torch.float32, device=dev)  # noqa: E731
        self.buffers = StepBuffers(z(), z(), z())
        return self.batches

    # -- per micro-batch ----------------------------------------------------------------------
    def logits_backward(self, j: int, logits: torch.Tensor, grad_out: torch.Tensor | None = None, temperature: float | None = None) -> torch.Tensor:
        """Fused K1 + token gradient + K1 backward for micro-batch j.  `logits` [1, T_j, V] (fp32 or
        bf16, contiguous).  Returns d loss / d logits (written to `grad_out`, which may be `logits`)."""
        lib = _lib.load()
        b = self.batches[j]
        a, e = self.offsets[j], self.offsets[j + 1]
        B, L, V = logits.shape
        assert B == 1 and L == e - a, "logits do not match the micro-batch"
        grad = torch.empty_like(logits) if grad_out is None else grad_out
        buf = self.buffers
        temp = float(self.config.temperature if temperature is None else temperature)
        dev = logits.device
        with torch.cuda.device(dev):
            _lib.check(
                lib.prl_fused_logits_loss(
                    ctypes.byref(self.cfg), 1, L, V, logits.data_ptr(), _logits_dtype_code(logits), V, temp,
                    b.input_ids.data_ptr(), b.labels.data_ptr(), b.old_logprobs.data_ptr(), b.ref_logprobs.data_ptr(),
                    b.advantages.data_ptr(), b.rewards.data_ptr(), b.group_tokens.data_ptr(), b.overflow.data_ptr(),
                    buf.new_logprobs[:, a:e].data_ptr(), buf.entropy[:, a:e].data_ptr(), buf.lse2[:, a:e].data_ptr(),
                    grad.data_ptr(), _lib.current_stream_ptr(dev),
                )
            )
        return grad

    def annotate_ref_logprobs(self, j: int, ref_logits: torch.Tensor, temperature: float | None = None) -> None:
        """Fill micro-batch j's `ref_logprobs` column from the logits of a frozen reference model that
        lives on this GPU (K1 forward, no graph): log p_ref(token | prefix) on labelled tokens, 0
        elsewhere - the values the reference pipeline fetches from a second inference server over HTTP
        (preprocess.py:86-104, llm.py:606-648).  Writes in place: the step-level statistics launch
        (`finish`) reads the same storage."""
        from .finetune.rl import logprob_entropy

        b = self.batches[j]
        temp = float(self.config.temperature if temperature is None else temperature)
        with torch.no_grad():
            nlp, _, _, _ = logprob_entropy(ref_logits, b.input_ids, temp)
            b.ref_logprobs.copy_(torch.where(b.labels != -100, nlp, torch.zeros_like(nlp)))

    def annotate_ref_logprobs_from_hidden(self, j: int, head, ref_hidden: torch.Tensor, temperature: float | None = None) -> None:
        """The same column from the reference model's last HIDDEN STATES [1, T_j, H] through the MFMA head
        (`fused_head.FusedLmHead(weight, backward=False)`): no `[T, V]` reference logits, only the rows that predict a
        labelled token enter the product.  Writes in place like `annotate_ref_logprobs`."""
        from .fused_head import token_logprobs_from_hidden

        b = self.batches[j]
        temp = float(self.config.temperature if temperature is None else temperature)
        b.ref_logprobs.copy_(token_logprobs_from_hidden(head, ref_hidden, b.input_ids, b.labels, temp))

    def logits_two_pass(self, j: int, logits: torch.Tensor, grad_out: torch.Tensor | None = None) -> torch.Tensor:
        """Unfused alternative: K1 forward, K2+K3 on the micro-batch, K1 backward (three launches)."""
        lib = _lib.load()
        b = self.batches[j]
        a, e = self.offsets[j], self.offsets[j + 1]
        _, L, V = logits.shape
        buf = self.buffers
        dev = logits.device
        nlp, ent, lse2 = buf.new_logprobs[:, a:e], buf.entropy[:, a:e], buf.lse2[:, a:e]
        grad = torch.empty_like(logits) if grad_out is None else grad_out
        temp = float(self.config.temperature)
        with torch.cuda.device(dev):
            stream = _lib.current_stream_ptr(dev)
            _lib.check(lib.prl_logprob_entropy_fwd(1, L, V, logits.data_ptr(), _logits_dtype_code(logits), V,
                                                   b.input_ids.data_ptr(), temp, nlp.data_ptr(), ent.data_ptr(),
                                                   lse2.data_ptr(), stream))
            _, _, g_nlp, g_ent = grpo_loss_from_logprobs(self.cfg, b, nlp, ent, want_grad=True)
            _lib.check(lib.prl_logprob_entropy_bwd(1, L, V, logits.data_ptr(), _logits_dtype_code(logits), V,
                                                   b.input_ids.data_ptr(), temp, lse2.data_ptr(), ent.data_ptr(),
                                                   g_nlp.data_ptr(), _lib.ptr(g_ent), None, grad.data_ptr(), stream))
        return grad

    # -- per step -----------------------------------------------------------------------------
    def finish(self, sync: bool = True, timer: Any = None) -> tuple[torch.Tensor, torch.Tensor]:
        """One K2+K3 launch over the whole step (+ one all-gather across ranks).  Returns the
        device loss scalar and the reduced stats vector (double[32], on device).
        `timer` (bench.py's EventTimer): event pairs around the launch alone ("grpo_loss_step") and around the
        cross-rank reduction ("stats_allgather"), so that a collective's wait is not charged to the kernel."""
        import contextlib

        timed = (lambda name: timer.time(name)) if timer is not None else (lambda name: contextlib.nullcontext())
        # the first token of every micro-batch is column 0 of ITS batch (no prediction exists for
        # it): flat_micro_batches makes the kernel skip positions with position_ids == 0
        flat_cfg = type(self.cfg).from_buffer_copy(self.cfg)
        flat_cfg.flat_micro_batches = 1
        with timed("grpo_loss_step"):
            loss, stats, _, _ = grpo_loss_from_logprobs(flat_cfg, self.step_batch, self.buffers.new_logprobs,
                                                       self.buffers.entropy, want_grad=False)
        with timed("stats_allgather"):
            stats = self.reduce_stats(stats)
        return loss, stats

    def reduce_stats(self, stats: torch.Tensor) -> torch.Tensor:
        import torch.distributed as dist

        if self.group is None and not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return stats
        world = dist.get_world_size(self.group)
        dev = stats.device
        if dist.get_backend(self.group) == "gloo":  # CPU collectives (tests / single-GPU dry runs)
            stats = stats.cpu()
        gathered = torch.empty((world, stats.numel()), dtype=stats.dtype, device=stats.device)
        dist.all_gather_into_tensor(gathered, stats.unsqueeze(0).contiguous(), group=self.group)
        gathered = gathered.to(dev)
        out = gathered.sum(dim=0)
        out[_MAX_LANES] = gathered[:, _MAX_LANES].max(dim=0).values
        out[_MIN_LANES] = gathered[:, _MIN_LANES].min(dim=0).values
        return out

    def stats_dict(self, stats: torch.Tensor) -> dict[str, float]:
        s = stats.cpu().tolist()
        check_finite(s)
        return stats_to_dict(s, self.kl_coef, self.ent_coef, int(self.step_batch.input_ids.numel()))


def dense_micro_batches(rollouts: RaggedRollouts, seq_length: int) -> list[list[int]]:
    """Greedy first-fit packing in arrival order into `seq_length`-token budgets (the packing rule
    of reference preprocess.py:610-625 without the per-step quota, which the caller applies)."""
    lens = rollouts.seq_lengths()
    out: list[list[int]] = []
    cur: list[int] = []
    used = 0
    for i, n in enumerate(lens):
        n = int(n)
        if cur and used + n > seq_length:
            out.append(cur)
            cur, used = [], 0
        cur.append(i)
        used += n
    if cur:
        out.append(cur)
    return out
