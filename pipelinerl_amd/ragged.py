"""Ragged SoA layout of a set of rollouts: the input format of the on-device preprocess path.

The reference moves rollouts around as lists of dicts of Python lists (the `actor` stream
record = `TrainingText.model_dump()`, actor.py:648-652) and expands every
per-sequence scalar into a per-token list.  Here a chunk of rollouts is a handful of flat
arrays that upload to HBM with one copy each:

    tokens, labels      int32  [N]      all sequences back to back
    logprobs            fp32   [Nc]     completion-token logprobs only (right-aligned in a sequence)
    ref_logprobs        fp32   [Nc]     or None (== logprobs when the KL term is off, preprocess.py:160-161)
    seq_off, lp_off     int64  [S+1]    exclusive prefix sums of lengths
    reward              fp64   [S]      pandas computes the baseline in fp64 (rl/__init__.py:478-521)
    group_index, step_index, rollout_index  int32 [S]
    model_version       int64  [S]
    finished, finish_code                   uint8 [S]
"""

from __future__ import annotations

from dataclasses import dataclass, field, fields
from typing import Any, Sequence

import numpy as np
import torch

from ._lib import PRL_FINISH_LENGTH, PRL_FINISH_NONE, PRL_FINISH_STOP


def finish_reason_code(finish_reason: Any) -> int:
    """Normalise a `finish_reason` string like reference rl/__init__.py:543-549."""
    if isinstance(finish_reason, str):
        fr = finish_reason.strip().lower()
        if fr == "length":
            return PRL_FINISH_LENGTH
        if fr in ("stop", "content_filter"):
            return PRL_FINISH_STOP
    return PRL_FINISH_NONE


@dataclass
class RaggedRollouts:
    tokens: torch.Tensor
    labels: torch.Tensor
    logprobs: torch.Tensor
    ref_logprobs: torch.Tensor | None
    seq_off: torch.Tensor
    lp_off: torch.Tensor
    reward: torch.Tensor
    group_index: torch.Tensor
    step_index: torch.Tensor
    rollout_index: torch.Tensor
    model_version: torch.Tensor
    finished: torch.Tensor
    finish_code: torch.Tensor
    group_ids: list[str] = field(default_factory=list)  # dense index -> original group_id
    # host copies of the O(S) metadata (kept so that host-side planning never syncs the device)
    host_seq_off: np.ndarray | None = None
    host_lp_off: np.ndarray | None = None
    host_group_index: np.ndarray | None = None
    host_step_index: np.ndarray | None = None
    host_rollout_index: np.ndarray | None = None
    host_model_version: np.ndarray | None = None

    @property
    def n_seqs(self) -> int:
        return int(self.seq_off.shape[0]) - 1

    @property
    def n_tokens(self) -> int:
        return int(self.tokens.shape[0])

    @property
    def device(self) -> torch.device:
        return self.tokens.device

    def seq_lengths(self) -> np.ndarray:
        return np.diff(self.host_seq_off)

    def to(self, device: str | torch.device, stager=None, extra=None):
        """Copies are enqueued `non_blocking` on the current stream: with `pin_memory()`ed sources
        they are true asynchronous DMAs that overlap the previous step's kernels.

        `stager` (a `staging.PinnedStager` of that device): the 13 arrays - and the host arrays in `extra`, e.g. the K5
        plan - are laid out back to back in one page-locked buffer and travel in ONE copy; the device tensors are views
        into one allocation.  Returns `(rollouts, device tensors of extra)` in that form."""
        if stager is not None:
            names = [f.name for f in fields(self) if isinstance(getattr(self, f.name), torch.Tensor)]
            up = stager.upload([getattr(self, n) for n in names] + list(extra or ()))
            kw = {f.name: getattr(self, f.name) for f in fields(self)}
            kw.update(zip(names, up[:len(names)]))
            return RaggedRollouts(**kw), up[len(names):]
        kw = {}
        for f in fields(self):
            v = getattr(self, f.name)
            kw[f.name] = v.to(device, non_blocking=True) if isinstance(v, torch.Tensor) else v
        return RaggedRollouts(**kw)

    def select(self, seq_range: range) -> "RaggedRollouts":
        """A contiguous run of sequences [start, stop) as its own ragged set (views of the flat buffers,
        offsets rebased) - how a data-parallel rank takes its share of a step's rollouts (whole groups)."""
        a, b = seq_range.start, seq_range.stop
        assert seq_range.step == 1 and 0 <= a <= b <= self.n_seqs
        t0, t1 = int(self.host_seq_off[a]), int(self.host_seq_off[b])
        l0, l1 = int(self.host_lp_off[a]), int(self.host_lp_off[b])
        so = (self.host_seq_off[a:b + 1] - t0).astype(np.int64)
        lo = (self.host_lp_off[a:b + 1] - l0).astype(np.int64)
        dev = self.device
        sl = slice(a, b)
        return RaggedRollouts(
            tokens=self.tokens[t0:t1], labels=self.labels[t0:t1], logprobs=self.logprobs[l0:l1],
            ref_logprobs=None if self.ref_logprobs is None else self.ref_logprobs[l0:l1],
            seq_off=torch.from_numpy(so).to(dev), lp_off=torch.from_numpy(lo).to(dev), reward=self.reward[sl],
            group_index=self.group_index[sl], step_index=self.step_index[sl], rollout_index=self.rollout_index[sl],
            model_version=self.model_version[sl], finished=self.finished[sl], finish_code=self.finish_code[sl],
            group_ids=self.group_ids, host_seq_off=so, host_lp_off=lo, host_group_index=self.host_group_index[sl],
            host_step_index=self.host_step_index[sl], host_rollout_index=self.host_rollout_index[sl],
            host_model_version=self.host_model_version[sl],
        )

    def pin_memory(self) -> "RaggedRollouts":
        """Page-locked host copy (one memcpy per column) so that `.to(device)` runs at PCIe rate
        instead of through the driver's pageable staging path."""
        kw = {}
        for f in fields(self):
            v = getattr(self, f.name)
            kw[f.name] = v.pin_memory() if isinstance(v, torch.Tensor) and not v.is_cuda else v
        return RaggedRollouts(**kw)

    def to_entries(self, finish_reasons: Sequence[Any] | None = None) -> list[dict[str, Any]]:
        """The same rollouts as `actor`-stream dicts (`TrainingText.model_dump()` layout, reference
        actor.py:648-652) - what the reference's preprocessor reads and what the JSONL mirror of the
        binary stream stores for `debug.streams_from` replay.  `finish_reasons` overrides the
        strings derived from `finish_code` (None = key absent)."""
        tokens = self.tokens.cpu().numpy()
        labels = self.labels.cpu().numpy()
        lp = self.logprobs.cpu().numpy()
        ref = None if self.ref_logprobs is None else self.ref_logprobs.cpu().numpy()
        so, lo = self.host_seq_off, self.host_lp_off
        reward = self.reward.cpu().numpy()
        fin = self.finished.cpu().numpy()
        codes = self.finish_code.cpu().numpy()
        names = {PRL_FINISH_LENGTH: "length", PRL_FINISH_STOP: "stop"}
        out = []
        for i in range(self.n_seqs):
            n_out = int(lo[i + 1] - lo[i])
            e: dict[str, Any] = {
                "text": "",
                "n_predicted": n_out,
                "reward": float(reward[i]),
                "logprobs": lp[lo[i] : lo[i + 1]].tolist(),
                "ref_logprobs": [] if ref is None else ref[lo[i] : lo[i + 1]].tolist(),
                "input_ids": tokens[so[i] : so[i + 1]].tolist(),
                "labels": labels[so[i] : so[i + 1]].tolist(),
                "group_id": self.group_ids[int(self.host_group_index[i])] if self.group_ids else str(int(self.host_group_index[i])),
                "finished": bool(fin[i]),
                "prompt_tokens": int((so[i + 1] - so[i]) - n_out),
                "output_tokens": n_out,
                "visual_features": None,
                "metadata": {
                    "model_version": int(self.host_model_version[i]),
                    "rollout_index": int(self.host_rollout_index[i]),
                    "step_index": int(self.host_step_index[i]),
                },
            }
            reason = finish_reasons[i] if finish_reasons is not None else names.get(int(codes[i]))
            if reason is not None:
                e["finish_reason"] = reason
            out.append(e)
        return out

    @classmethod
    def from_numpy(
        cls,
        tokens: np.ndarray,
        labels: np.ndarray,
        logprobs: np.ndarray,
        ref_logprobs: np.ndarray | None,
        seq_off: np.ndarray,
        lp_off: np.ndarray,
        reward: np.ndarray,
        group_index: np.ndarray,
        step_index: np.ndarray,
        rollout_index: np.ndarray,
        model_version: np.ndarray,
        finished: np.ndarray,
        finish_code: np.ndarray,
        group_ids: Sequence[str] | None = None,
    ) -> "RaggedRollouts":
        t = torch.from_numpy
        seq_off = np.ascontiguousarray(seq_off, dtype=np.int64)
        lp_off = np.ascontiguousarray(lp_off, dtype=np.int64)
        gi = np.ascontiguousarray(group_index, dtype=np.int32)
        si = np.ascontiguousarray(step_index, dtype=np.int32)
        ri = np.ascontiguousarray(rollout_index, dtype=np.int32)
        mv = np.ascontiguousarray(model_version, dtype=np.int64)
        return cls(
            tokens=t(np.ascontiguousarray(tokens, dtype=np.int32)),
            labels=t(np.ascontiguousarray(labels, dtype=np.int32)),
            logprobs=t(np.ascontiguousarray(logprobs, dtype=np.float32)),
            ref_logprobs=None if ref_logprobs is None else t(np.ascontiguousarray(ref_logprobs, dtype=np.float32)),
            seq_off=t(seq_off),
            lp_off=t(lp_off),
            reward=t(np.ascontiguousarray(reward, dtype=np.float64)),
            group_index=t(gi),
            step_index=t(si),
            rollout_index=t(ri),
            model_version=t(mv),
            finished=t(np.ascontiguousarray(finished, dtype=np.uint8)),
            finish_code=t(np.ascontiguousarray(finish_code, dtype=np.uint8)),
            group_ids=list(group_ids) if group_ids is not None else [],
            host_seq_off=seq_off,
            host_lp_off=lp_off,
            host_group_index=gi,
            host_step_index=si,
            host_rollout_index=ri,
            host_model_version=mv,
        )

    @classmethod
    def from_entries(cls, entries: Sequence[dict[str, Any]]) -> "RaggedRollouts":
        """Build from `actor`-stream style dicts (input_ids, labels, logprobs, reward, group_id,
        metadata{model_version, rollout_index, step_index} or the flattened keys, optional
        ref_logprobs / finished / finish_reason)."""
        n = len(entries)
        lens = np.fromiter((len(e["input_ids"]) for e in entries), dtype=np.int64, count=n)
        lp_lens = np.fromiter((len(e["logprobs"]) for e in entries), dtype=np.int64, count=n)
        seq_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lens, out=seq_off[1:])
        lp_off = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(lp_lens, out=lp_off[1:])
        tokens = np.empty(seq_off[-1], dtype=np.int32)
        labels = np.empty(seq_off[-1], dtype=np.int32)
        logprobs = np.empty(lp_off[-1], dtype=np.float32)
        have_ref = all(("ref_logprobs" in e and e["ref_logprobs"] is not None and len(e["ref_logprobs"]) == len(e["logprobs"])) for e in entries) and n > 0
        ref = np.empty(lp_off[-1], dtype=np.float32) if have_ref else None
        reward = np.empty(n, dtype=np.float64)
        group_index = np.empty(n, dtype=np.int32)
        step_index = np.empty(n, dtype=np.int32)
        rollout_index = np.empty(n, dtype=np.int32)
        model_version = np.empty(n, dtype=np.int64)
        finished = np.zeros(n, dtype=np.uint8)
        finish_code = np.zeros(n, dtype=np.uint8)
        gid_to_idx: dict[Any, int] = {}
        for i, e in enumerate(entries):
            if len(e["labels"]) != lens[i]:
                raise ValueError(f"entry {i}: labels and input_ids differ in length")
            tokens[seq_off[i] : seq_off[i + 1]] = e["input_ids"]
            labels[seq_off[i] : seq_off[i + 1]] = e["labels"]
            logprobs[lp_off[i] : lp_off[i + 1]] = e["logprobs"]
            if ref is not None:
                ref[lp_off[i] : lp_off[i + 1]] = e["ref_logprobs"]
            reward[i] = e["reward"]
            meta = e.get("metadata") or {}
            model_version[i] = e["model_version"] if "model_version" in e else meta.get("model_version", 0)
            rollout_index[i] = e["rollout_index"] if "rollout_index" in e else meta.get("rollout_index", 0)
            step_index[i] = e["step_index"] if "step_index" in e else meta.get("step_index", 0)
            gid = e.get("group_id")
            group_index[i] = gid_to_idx.setdefault(gid, len(gid_to_idx))
            finished[i] = 1 if e.get("finished") else 0
            finish_code[i] = finish_reason_code(e.get("finish_reason"))
        return cls.from_numpy(
            tokens, labels, logprobs, ref, seq_off, lp_off, reward, group_index, step_index, rollout_index,
            model_version, finished, finish_code, group_ids=[str(g) for g in gid_to_idx],
        )


def _cat_ref(parts: Sequence["RaggedRollouts"]) -> torch.Tensor:
    ts = [p.ref_logprobs if p.ref_logprobs is not None else p.logprobs for p in parts]
    if all(t.device.type == "cpu" for t in ts):
        return torch.from_numpy(np.concatenate([t.numpy() for t in ts]))
    return torch.cat(ts)


def concat_ragged(parts: Sequence["RaggedRollouts"]) -> "RaggedRollouts":
    """Concatenate rollouts that live on the same device (offsets and group indices are rebased)."""
    if len(parts) == 1:
        return parts[0]
    dev = parts[0].device
    tok_base = np.cumsum([0] + [p.n_tokens for p in parts])
    lp_base = np.cumsum([0] + [int(p.logprobs.shape[0]) for p in parts])
    grp_base = np.cumsum([0] + [len(p.group_ids) if p.group_ids else (int(p.host_group_index.max()) + 1 if p.n_seqs else 0) for p in parts])
    def cat(name):
        ts = [getattr(p, name) for p in parts]
        if all(t.device.type == "cpu" for t in ts):
            # host parts (decoded stream records): a plain single-threaded memcpy.  torch.cat hands anything above 32 K elements
            # to its intra-op thread pool, whose wake-up under a CPU quota costs more than the copy (5-9 ms per 1.5 MB chunk measured)
            return torch.from_numpy(np.concatenate([t.numpy() for t in ts]))
        return torch.cat(ts)

    seq_off = np.concatenate([[0]] + [p.host_seq_off[1:] + tok_base[i] for i, p in enumerate(parts)]).astype(np.int64)
    lp_off = np.concatenate([[0]] + [p.host_lp_off[1:] + lp_base[i] for i, p in enumerate(parts)]).astype(np.int64)
    gi = np.concatenate([p.host_group_index + grp_base[i] for i, p in enumerate(parts)]).astype(np.int32)
    have_ref = all(p.ref_logprobs is not None for p in parts)
    return RaggedRollouts(
        tokens=cat("tokens"), labels=cat("labels"), logprobs=cat("logprobs"),
        ref_logprobs=_cat_ref(parts) if (have_ref or any(p.ref_logprobs is not None for p in parts)) else None,
        seq_off=torch.from_numpy(seq_off).to(dev), lp_off=torch.from_numpy(lp_off).to(dev), reward=cat("reward"),
        group_index=torch.from_numpy(gi).to(dev), step_index=cat("step_index"), rollout_index=cat("rollout_index"),
        model_version=cat("model_version"), finished=cat("finished"), finish_code=cat("finish_code"),
        group_ids=[g for p in parts for g in p.group_ids],
        host_seq_off=seq_off, host_lp_off=lp_off, host_group_index=gi,
        host_step_index=np.concatenate([p.host_step_index for p in parts]),
        host_rollout_index=np.concatenate([p.host_rollout_index for p in parts]),
        host_model_version=np.concatenate([p.host_model_version for p in parts]),
    )
