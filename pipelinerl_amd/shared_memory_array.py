"""`SharedMemoryQueue` with the reference's interface (pipelinerl/shared_memory_array.py:109-196)
on top of the native ring (csrc/prl_ring.cpp).

The reference keeps a slot array in a `SharedMemoryManager` segment plus two `multiprocessing.Queue`s
of slot indices (every put/get costs a pipe write + a feeder-thread hop on each side); here slot
claiming is a lock-free ticket in the same shared segment and blocked peers park on a futex.
Items are still pickled, so arbitrary Python objects (lists of rollout dicts, exception info)
pass through unchanged.
"""

from __future__ import annotations

import pickle
from typing import Any

from .ring import Ring


class SharedMemoryQueue:
    def __init__(self, smm: Any, max_size: int, max_entry_size: int):
        """`smm` (a SharedMemoryManager in the reference) is accepted for signature compatibility
        and ignored: the ring owns its POSIX shared-memory segment."""
        if max_size <= 0:
            raise ValueError("Number of entries must be positive")
        if max_entry_size <= 0:
            raise ValueError("Maximum entry size must be positive")
        self.max_size = max_size
        self.max_entry_size = max_entry_size
        self._ring = Ring(n_slots=max_size, slot_bytes=max_entry_size)

    def put(self, item: Any, block: bool = True, timeout: float | None = None) -> None:
        """Raises queue.Full when no slot frees up, ValueError when the pickle exceeds a slot.
        (The reference takes the slot before it checks the size and never returns it, so every
        oversize put permanently shrinks its queue by one slot — shared_memory_array.py:150-158,
        pinned in tests/golden/queue_trace.json; here the capacity is unchanged.)"""
        # default pickle protocol, as the reference: the oversize threshold is on the pickled size
        self._ring.put_bytes(pickle.dumps(item), block=block, timeout=timeout)

    def get(self, block: bool = True, timeout: float | None = None) -> Any:
        """Raises queue.Empty when nothing arrives in time."""
        return pickle.loads(self._ring.get_bytes(block=block, timeout=timeout))

    def get_memory_size(self) -> int:
        return self.max_size * (self.max_entry_size + 64)

    def full(self) -> bool:
        return self._ring.qsize() >= self.max_size

    def qsize(self) -> int:
        return self._ring.qsize()

    def max_actual_entry_size(self) -> int:
        return self._ring.max_record_bytes()

    def close(self) -> None:
        self._ring.close()
