"""`SharedMemoryQueue` with the reference's interface (pipelinerl/shared_memory_array.py:109-196)
on top of the native ring (csrc/prl_ring.cpp), and `SharedMemoryArray` (:9-106), the indexed slot array of pickled objects.

The reference keeps a slot array in a `SharedMemoryManager` segment plus two `multiprocessing.Queue`s
of slot indices (every put/get costs a pipe write + a feeder-thread hop on each side); here slot
claiming is a lock-free ticket in the same shared segment and blocked peers park on a futex.
Items are still pickled, so arbitrary Python objects (lists of rollout dicts, exception info)
pass through unchanged.
"""

from __future__ import annotations

import pickle
from typing import Any

import numpy as np

from .ring import Ring


class SharedMemoryArray:
    """A fixed number of slots of pickled Python objects in one shared-memory segment, addressed by index (reference :9-106; there it is
    the storage under `SharedMemoryQueue`, here the queue has its own ring and this class stands on its own).  Layout: a table of
    `num_entries` uint32 sizes, then the slots of `max_entry_size` bytes each (the reference interleaves size and payload; nothing reads
    one layout with the other).  The segment comes from `smm.SharedMemory` when a SharedMemoryManager is given - it then lives and dies
    with the manager, and the object can be sent to a child process like the reference's - otherwise from an unmanaged segment that
    `close()` unlinks.  No locking, as in the reference: one writer per slot at a time."""

    def __init__(self, smm: Any, num_entries: int, max_entry_size: int):
        if num_entries <= 0:
            raise ValueError("Number of entries must be positive")
        if max_entry_size <= 0:
            raise ValueError("Maximum entry size must be positive")
        self.num_entries, self.max_entry_size = int(num_entries), int(max_entry_size)
        self._table_bytes = 4 * self.num_entries
        nbytes = self._table_bytes + self.num_entries * self.max_entry_size
        if smm is not None and hasattr(smm, "SharedMemory"):
            self.shared_mem, self._owned = smm.SharedMemory(size=nbytes), False
        else:
            from multiprocessing import shared_memory

            self.shared_mem, self._owned = shared_memory.SharedMemory(create=True, size=nbytes), True
        self._sizes()[:] = 0
        self._max_actual_entry_size = 0

    def _sizes(self) -> np.ndarray:
        return np.ndarray((self.num_entries,), dtype=np.uint32, buffer=self.shared_mem.buf)

    def _slot(self, index: int) -> int:
        if not 0 <= index < self.num_entries:
            raise IndexError(f"Index {index} out of range (0-{self.num_entries - 1})")
        return self._table_bytes + index * self.max_entry_size

    def get_memory_size(self) -> int:
        return self.shared_mem.size

    def __len__(self) -> int:
        return self.num_entries

    def __getitem__(self, index: int) -> Any:
        at = self._slot(index)
        size = int(self._sizes()[index])
        self._max_actual_entry_size = max(self._max_actual_entry_size, size)
        if size == 0:
            return None
        return pickle.loads(self.shared_mem.buf[at: at + size])

    def __setitem__(self, index: int, value: Any) -> None:
        data = pickle.dumps(value)
        if len(data) > self.max_entry_size:
            raise ValueError(f"Serialized object size ({len(data)} bytes) exceeds maximum entry size ({self.max_entry_size} bytes)")
        at = self._slot(index)
        self.shared_mem.buf[at: at + len(data)] = data
        self._sizes()[index] = len(data)  # the size goes last: a reader never sees a size whose bytes are not there yet
        self._max_actual_entry_size = max(self._max_actual_entry_size, len(data))

    def max_actual_entry_size(self) -> int:
        """Largest entry this process has written or read."""
        return self._max_actual_entry_size

    def close(self) -> None:
        self.shared_mem.close()
        if self._owned:
            try:
                self.shared_mem.unlink()
            except FileNotFoundError:
                pass


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
