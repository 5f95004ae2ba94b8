"""Python handles of the shared-memory record ring (C ABI `prl_ring_*`, csrc/prl_ring.cpp) and of the
shared-memory record log (`prl_log_*`, csrc/prl_log.cpp)."""

from __future__ import annotations

import ctypes
import logging
import os
import queue
import time
import uuid

from . import _lib

logger = logging.getLogger(__name__)


class Ring:
    """Bounded multi-producer / multi-consumer queue of byte records in POSIX shared memory.
    Picklable: a child process (fork or spawn) re-attaches by name."""

    def __init__(self, name: str | None = None, n_slots: int = 0, slot_bytes: int = 0, create: bool = True):
        lib = _lib.load()
        self.name = name or f"prl_{os.getpid()}_{uuid.uuid4().hex[:12]}"
        h = ctypes.c_void_p()
        if create:
            _lib.check(lib.prl_ring_create(self.name.encode(), n_slots, slot_bytes, ctypes.byref(h)))
        else:
            _lib.check(lib.prl_ring_attach(self.name.encode(), ctypes.byref(h)))
        self._h = h
        self._owner_pid = os.getpid() if create else None
        n, sb = ctypes.c_uint32(), ctypes.c_uint64()
        _lib.check(lib.prl_ring_capacity(self._h, ctypes.byref(n), ctypes.byref(sb)))
        self.n_slots, self.slot_bytes = n.value, sb.value

    # -- pickling: attach by name in the other process ------------------------------------------
    def __getstate__(self):
        return {"name": self.name}

    def __setstate__(self, state):
        self.__init__(state["name"], create=False)

    @staticmethod
    def _timeout_ms(block: bool, timeout: float | None) -> int:
        if not block:
            return 0
        return -1 if timeout is None else max(0, int(timeout * 1000))

    def put_bytes(self, data: bytes | bytearray | memoryview, block: bool = True, timeout: float | None = None) -> None:
        lib = _lib.load()
        buf = (ctypes.c_char * len(data)).from_buffer_copy(data) if not isinstance(data, bytes) else data
        rc = lib.prl_ring_put(self._h, buf, len(data), self._timeout_ms(block, timeout))
        if rc in (_lib.PRL_EAGAIN, _lib.PRL_ETIMEDOUT):
            raise queue.Full()
        if rc == _lib.PRL_EMSGSIZE:
            raise ValueError(f"Serialized object size ({len(data)} bytes) exceeds maximum entry size ({self.slot_bytes} bytes)")
        _lib.check(rc)

    def get_bytes(self, block: bool = True, timeout: float | None = None) -> bytes:
        lib = _lib.load()
        p, n, ticket = ctypes.c_void_p(), ctypes.c_uint64(), ctypes.c_uint64()
        rc = lib.prl_ring_acquire(self._h, ctypes.byref(p), ctypes.byref(n), ctypes.byref(ticket), self._timeout_ms(block, timeout))
        if rc in (_lib.PRL_EAGAIN, _lib.PRL_ETIMEDOUT):
            raise queue.Empty()
        _lib.check(rc)
        try:
            return ctypes.string_at(p.value, n.value)
        finally:
            lib.prl_ring_release(self._h, ticket.value)

    def qsize(self) -> int:
        n = ctypes.c_uint64()
        _lib.check(_lib.load().prl_ring_size(self._h, ctypes.byref(n)))
        return n.value

    def max_record_bytes(self) -> int:
        n = ctypes.c_uint64()
        _lib.check(_lib.load().prl_ring_max_record_bytes(self._h, ctypes.byref(n)))
        return n.value

    def close(self, unlink: bool = True) -> None:
        """Detach.  The creator also removes the shared-memory name unless `unlink=False` (stream
        writers keep the segment for readers that attach later and unlink at process exit)."""
        if getattr(self, "_h", None) and self._h:
            lib = _lib.load()
            if self._owner_pid is not None and self._owner_pid != os.getpid():
                # a forked child inherits the creator's handle: it must not unlink the segment
                pass
            elif unlink:
                lib.prl_ring_close(self._h)
            else:
                lib.prl_ring_detach(self._h)
            self._h = ctypes.c_void_p()

    @staticmethod
    def unlink_name(name: str) -> None:
        _lib.load().prl_ring_unlink(name.encode())

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Log:
    """Append-only record log with per-reader cursors (csrc/prl_log.cpp): every reader handle sees
    every record from the first retained one, a writer never waits for readers and may be closed and
    reopened, readers park on a futex.  One handle is either used for `append` or for `read`."""

    def __init__(self, name: str, create: bool = False, truncate: bool = False, reader: bool = False, trim: bool = False,
                 segment_bytes: int = 64 << 20, wait: float | None = None, takeover_after: float | None = 5.0):
        """`wait`: seconds to keep retrying while the log does not exist yet (None = forever for readers,
        no retry for writers that create).
        `takeover_after` (writers that may create): a control block that exists but stays uninitialised this long
        belongs to a creator that died between `shm_open` and publishing the magic word - nobody can ever have
        read or written a record of it - so it is removed and created again instead of being waited for forever."""
        lib = _lib.load()
        self.name = name
        flags = (_lib.PRL_LOG_CREATE if create else 0) | (_lib.PRL_LOG_TRUNCATE if truncate else 0) \
            | (_lib.PRL_LOG_READER if reader else 0) | (_lib.PRL_LOG_TRIM if trim else 0)
        h = ctypes.c_void_p()
        deadline = None if wait is None else time.time() + wait
        stuck_since, stuck_ident, warned = None, None, 0.0
        while True:
            rc = lib.prl_log_open(name.encode(), segment_bytes, flags, ctypes.byref(h))
            if rc == _lib.PRL_OK:
                break
            # EAGAIN: another process is creating it right now; EFAULT: not there yet (readers wait)
            if rc == _lib.PRL_EAGAIN or (rc == _lib.PRL_EFAULT and not create):
                now = time.time()
                if deadline is not None and now > deadline:
                    _lib.check(rc)
                flags &= ~_lib.PRL_LOG_TRUNCATE  # never truncate twice
                if rc == _lib.PRL_EAGAIN and create and takeover_after is not None:
                    # the timer belongs to ONE object: when the control block disappears or is replaced (another waiter took
                    # it over, or a live creator finally re-created it) the wait starts again
                    ident = self._shm_identity(name)
                    if ident != stuck_ident:
                        stuck_since, stuck_ident = now, ident
                    if ident is not None and now - stuck_since > takeover_after:
                        if self._take_over(lib, name, ident, segment_bytes, flags):
                            logger.warning(f"shm log {name}: control block uninitialised for {now - stuck_since:.1f} s, its creator "
                                           "is gone - removed it, creating the log again")
                        stuck_since, stuck_ident = None, None
                    elif now - warned > 1.0:
                        logger.info(f"shm log {name} is being created by another process, waiting")
                        warned = now
                time.sleep(0.002)
                continue
            _lib.check(rc)
        self._h = h

    @staticmethod
    def _shm_identity(name: str):
        """(inode, creation-ish time) of the log's control block, None when it does not exist."""
        try:
            st = os.stat("/dev/shm/" + name.lstrip("/"))
            return (st.st_ino, st.st_ctime_ns)
        except OSError:
            return None

    @classmethod
    def _take_over(cls, lib, name: str, ident, segment_bytes: int, flags: int) -> bool:
        """Remove a control block that stayed uninitialised - serialised between the waiters by an exclusive flock on a
        sidecar, and only if, under that lock, the object is STILL the one this waiter timed (same inode) and STILL answers
        "being created".  Without the re-check a waiter whose timer expires a moment after another one already replaced
        the object would unlink the live replacement (round-3 advisor finding), and its creator would write into an
        orphan.  The sidecar is NEVER unlinked here: a waiter blocked in flock holds the old inode, a newcomer after an unlink
        would lock a fresh file, and the two would be in the critical section together (round-4 advisor finding) - it is a
        zero-byte file that `streams.clean_shm_streams` / `begin_run` remove with the experiment's other shm objects (same
        name prefix).  Returns True when this call removed the control block."""
        import fcntl

        lock_path = "/dev/shm/" + name.lstrip("/") + ".takeover"
        try:
            fd = os.open(lock_path, os.O_CREAT | os.O_RDWR, 0o600)
        except OSError:
            return False
        try:
            fcntl.flock(fd, fcntl.LOCK_EX)
            if cls._shm_identity(name) != ident:
                return False  # somebody else dealt with it while we waited for the lock
            h = ctypes.c_void_p()
            rc = lib.prl_log_open(name.encode(), segment_bytes, flags & ~_lib.PRL_LOG_TRUNCATE & ~_lib.PRL_LOG_CREATE, ctypes.byref(h))
            if rc == _lib.PRL_OK:  # it came up after all: a slow creator, not a dead one
                lib.prl_log_close(h)
                return False
            if rc != _lib.PRL_EAGAIN:
                return False
            lib.prl_log_unlink(name.encode())
            return True
        finally:
            os.close(fd)  # releases the flock; the sidecar stays (see above)

    def append(self, data: bytes | bytearray | memoryview) -> None:
        if isinstance(data, bytes):
            buf = data
        elif isinstance(data, bytearray):
            buf = (ctypes.c_char * len(data)).from_buffer(data) if len(data) else b""  # no copy: the library reads it in place
        else:
            buf = (ctypes.c_char * len(data)).from_buffer_copy(data)
        _lib.check(_lib.load().prl_log_append(self._h, buf, len(data)))

    def appendv(self, pieces: "list[tuple[int, int, int]]", nbytes: int, keep_alive=None) -> None:
        """One record of `nbytes` gathered from `pieces` = (source address, offset in the record, length), ascending and
        non-overlapping - copied straight into the shared-memory segment (`prl_log_appendv`), no record buffer in between.
        `keep_alive`: whatever owns the source memory (held until the call returns)."""
        iov = (_lib.PrlLogIov * len(pieces))(*pieces)
        _lib.check(_lib.load().prl_log_appendv(self._h, iov, len(pieces), nbytes))
        del keep_alive

    def read(self, block: bool = True, timeout: float | None = None) -> bytearray:
        """Next record of this handle's cursor: ONE copy out of the shared-memory segment into a `bytearray` the caller
        owns (the reader's mapping of a segment is released when it moves on to the next one, so a view into the
        segment could not outlive the call).  `batch_codec.decode` builds its tensors as views over that buffer.
        Raises `queue.Empty` at the tail when not blocking."""
        p, n = ctypes.c_void_p(), ctypes.c_uint64()
        rc = _lib.load().prl_log_read(self._h, ctypes.byref(p), ctypes.byref(n), Ring._timeout_ms(block, timeout))
        if rc in (_lib.PRL_EAGAIN, _lib.PRL_ETIMEDOUT):
            raise queue.Empty()
        _lib.check(rc)
        if n.value == 0:
            return bytearray()
        return bytearray((ctypes.c_char * n.value).from_address(p.value))

    def stats(self) -> dict[str, int]:
        v = [ctypes.c_uint64() for _ in range(4)]
        _lib.check(_lib.load().prl_log_stats(self._h, *[ctypes.byref(x) for x in v]))
        return dict(zip(("records", "bytes", "first_segment", "segments"), (x.value for x in v)))

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h:
            _lib.load().prl_log_close(self._h)
            self._h = ctypes.c_void_p()

    @staticmethod
    def unlink_name(name: str) -> None:
        _lib.load().prl_log_unlink(name.encode())

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
