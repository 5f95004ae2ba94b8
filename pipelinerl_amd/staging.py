"""Batched host <-> device transfers for the preprocessor (SURVEY.md §8 a12 / a6 / a8 at the reference's chunk granularity).

The reference preprocesses `chunk_n_groups` = 2 groups at a time (preprocess.py:190-228, conf/base.yaml:25-44): 16
rollouts, ~131 k tokens.  At that size the kernels (K5 ~70 us, K6 ~5 us) are dwarfed by what surrounds them when every
array travels alone: a ragged chunk is 13 arrays, the K5 plan 5, the K6 plan 3, a packed micro-batch 12 columns - a
pageable `tensor.to(device)` or `.cpu()` each, every one a blocking driver call of 10-30 us.  Here a chunk costs ONE
host -> device copy (ragged arrays + K5 plan laid out back to back in a page-locked buffer, device tensors are views
into one allocation), one more for the K6 plan, and ONE device -> host copy of the whole packed block.

`PinnedStager` owns a small ring of page-locked buffers; a slot is reused only after the copy that read (or wrote) it
has completed (an event per slot), so uploads stay asynchronous.
"""

from __future__ import annotations

from typing import Sequence

import numpy as np
import torch

_ALIGN = 256  # device views start on 256-byte boundaries (every kernel of the path needs <= 16)

_NP_TO_TORCH = {np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64, np.dtype(np.float32): torch.float32,
                np.dtype(np.float64): torch.float64, np.dtype(np.uint8): torch.uint8}


def _layout(sizes: Sequence[int]) -> tuple[list[int], int]:
    offs, off = [], 0
    for n in sizes:
        off += (-off) % _ALIGN
        offs.append(off)
        off += n
    return offs, off


class PinnedStager:
    """A ring of page-locked host buffers for `upload` (many host arrays -> one H2D -> device views) and `download`
    (one device block -> one D2H -> a host tensor valid until the slot comes round again)."""

    def __init__(self, device: torch.device | str, slots: int = 4, min_bytes: int = 1 << 20):
        self.device = torch.device(device)
        self.min_bytes = int(min_bytes)
        self._bufs: list[torch.Tensor | None] = [None] * slots
        self._events: list[torch.cuda.Event | None] = [None] * slots
        self._next = 0
        self.uploads = self.downloads = 0
        self.bytes_up = self.bytes_down = 0

    def _slot(self, nbytes: int) -> tuple[int, torch.Tensor]:
        k = self._next
        self._next = (k + 1) % len(self._bufs)
        ev = self._events[k]
        if ev is not None:
            ev.synchronize()  # the copy that last used this slot is done
            self._events[k] = None
        buf = self._bufs[k]
        if buf is None or buf.numel() < nbytes:
            size = max(self.min_bytes, 1 << (max(nbytes, 1) - 1).bit_length())
            # (a CPU "device" - the layout logic under test without a GPU - takes ordinary memory)
            buf = self._bufs[k] = torch.empty(size, dtype=torch.uint8, pin_memory=self.device.type == "cuda")
        return k, buf

    def upload(self, arrays: Sequence[np.ndarray | torch.Tensor | None]) -> list[torch.Tensor | None]:
        """Host arrays (numpy or CPU tensors; None passes through) -> device tensors of the same dtype and shape, views
        into ONE device allocation filled by ONE asynchronous copy on the current stream."""
        arrs = [None if a is None else (a.numpy() if isinstance(a, torch.Tensor) else np.ascontiguousarray(a)) for a in arrays]
        live = [a for a in arrs if a is not None]
        offs, total = _layout([a.nbytes for a in live])
        if total == 0:
            return [None if a is None else torch.empty(a.shape, dtype=_NP_TO_TORCH[a.dtype], device=self.device) for a in arrs]
        k, buf = self._slot(total)
        host = buf.numpy()
        for a, o in zip(live, offs):
            if a.nbytes:
                host[o:o + a.nbytes] = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
        if self.device.type == "cuda":
            with torch.cuda.device(self.device):
                dev = torch.empty(total, dtype=torch.uint8, device=self.device)
                dev.copy_(buf[:total], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            self._events[k] = ev
        else:
            dev = buf[:total].clone()
        self.uploads += 1
        self.bytes_up += total
        out, it = [], iter(offs)
        for a in arrs:
            if a is None:
                out.append(None)
                continue
            o = next(it)
            out.append(dev[o:o + a.nbytes].view(_NP_TO_TORCH[a.dtype]).view(a.shape))
        return out

    def download(self, block: torch.Tensor) -> torch.Tensor:
        """A contiguous device tensor -> a host tensor of the same dtype / shape in page-locked memory (ONE copy, waited
        for).  The result is a VIEW of a ring slot: consume it before `slots` further transfers."""
        n = block.numel() * block.element_size()
        k, buf = self._slot(n)
        host = buf[:n].view(block.dtype).view(block.shape)
        if block.is_cuda:
            with torch.cuda.device(block.device):
                host.copy_(block, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
            ev.synchronize()
        else:
            host.copy_(block)
        self.downloads += 1
        self.bytes_down += n
        return host
