"""Trainer -> inference-worker weight synchronisation over RCCL / xGMI.

Replaces (same roles, same message types):
  * `stateless_init_process_group`  reference pipelinerl/torch_utils.py:70-94 — a communicator
    that is independent of torch.distributed's default group; here an RCCL communicator created
    through the C ABI (`prl_wsync_*`), its unique id published through a `torch.distributed.TCPStore`.
  * `WeightUpdateManager.send_weight_update`  reference pipelinerl/finetune_loop.py:205-292 — one
    NCCL broadcast per parameter (339 for Qwen2.5-7B); here parameters are flattened into large
    byte buckets and each bucket moves with one collective (plain broadcast, or scatter +
    all-gather which uses every xGMI link of the mesh).
  * `WorkerExtension.receive_weight_update`  reference pipelinerl/vllm1.py:110-127 — the receiver
    unflattens each bucket into (name, tensor) views and hands them to a `load_weights` callback.

Control plane (HTTP trigger, `WeightUpdateRequest/Success` messages on the `weight_update_request`
stream) lives in `finetune_loop.py`; this module is the data plane.
"""

from __future__ import annotations

import ctypes
import datetime
from dataclasses import dataclass
from typing import Any, Callable, Iterable, Sequence
from urllib.parse import urlparse

import torch

from . import _lib

DEFAULT_BUCKET_BYTES = 1 << 30  # 1 GiB buckets: ~15 collectives for Qwen2.5-7B instead of 339


@dataclass
class ParamSpec:
    name: str
    shape: tuple[int, ...]
    dtype: torch.dtype

    @property
    def nbytes(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n * torch.empty((), dtype=self.dtype).element_size()


def _align(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


def plan_buckets(specs: Sequence[ParamSpec], bucket_bytes: int = DEFAULT_BUCKET_BYTES) -> list[list[tuple[ParamSpec, int]]]:
    """Greedy, order-preserving assignment of parameters to buckets; every parameter starts on a
    256-byte boundary inside its bucket.  Both sides derive the same plan from `parameters_info`
    (name / shape / dtype in the trainer's named_parameters order), so no layout is transmitted."""
    buckets: list[list[tuple[ParamSpec, int]]] = []
    cur: list[tuple[ParamSpec, int]] = []
    used = 0
    for sp in specs:
        need = _align(sp.nbytes)
        if cur and used + need > bucket_bytes:
            buckets.append(cur)
            cur, used = [], 0
        cur.append((sp, used))
        used += need
    if cur:
        buckets.append(cur)
    return buckets


def bucket_nbytes(bucket: Sequence[tuple[ParamSpec, int]]) -> int:
    sp, off = bucket[-1]
    return off + _align(sp.nbytes)


# `hipIpcOpenMemHandle` never returns for an allocation of 2 GiB or more on this stack (ROCm 7.0 / dmabuf IPC; measured with
# profiles/r05r_ipc_open_by_allocation_size.txt: 1.9 GiB opens in 0.2 ms, 2.0 GiB and above not within 40 s - the 7B-shaped
# pipeline, whose fp32 output head is ONE 2.18 GB tensor, stalled there).  No exported bucket may reach it.
IPC_MAX_ALLOCATION = (1 << 31) - (1 << 20)
_ROWS = "#rows"


def split_for_ipc(specs: Sequence[ParamSpec], piece_bytes: int = DEFAULT_BUCKET_BYTES, max_allocation: int = IPC_MAX_ALLOCATION) -> list[ParamSpec]:
    """The parameter list of the HIP-IPC transport: a tensor too large for ONE exportable allocation becomes consecutive row
    ranges, pseudo-parameters named `<name>#rows<a>:<b>` of at most `piece_bytes` each (every other tensor is itself).  Sender
    and receiver derive the same list from `parameters_info`, so nothing about it travels."""
    out: list[ParamSpec] = []
    for sp in specs:
        if _align(sp.nbytes) < max_allocation:
            out.append(sp)
            continue
        rows = sp.shape[0] if sp.shape else 0
        row_bytes = sp.nbytes // rows if rows else 0
        if rows < 2 or _align(row_bytes) >= max_allocation:
            raise ValueError(f"{sp.name}: {sp.nbytes} bytes cannot be exported over HIP IPC (allocations stay below {max_allocation} bytes) "
                             "and has no leading dimension to cut")
        per = max(1, min(piece_bytes, max_allocation - 256) // row_bytes)
        for a in range(0, rows, per):
            b = min(rows, a + per)
            out.append(ParamSpec(f"{sp.name}{_ROWS}{a}:{b}", (b - a,) + tuple(sp.shape[1:]), sp.dtype))
    return out


def rows_piece(name: str) -> tuple[str, int, int] | None:
    """`<name>#rows<a>:<b>` -> (name, a, b); any other name -> None."""
    base, sep, rng = name.rpartition(_ROWS)
    if not sep or ":" not in rng:
        return None
    a, _, b = rng.partition(":")
    return (base, int(a), int(b)) if a.isdigit() and b.isdigit() else None


def plan_shard_buckets(specs: Sequence[ParamSpec], shards: dict, tp_rank: int, tp_size: int,
                       bucket_bytes: int = DEFAULT_BUCKET_BYTES):
    """Bucket plan of the TP-aware update, derived identically by the trainer (for every TP rank) and by each worker
    (for its own): parameters are grouped by their FULL size into groups of `bucket_bytes * tp_size` - one
    gather of a sharded trainer and one bucket per TP rank each - and inside a group TP rank t's bucket holds
    its slice of every parameter (tp_shard.TpShard; replicated parameters whole), 256-byte aligned.
    Returns (groups of full specs, this rank's buckets of (slice spec, offset))."""
    groups = plan_buckets(specs, bucket_bytes * tp_size)
    mine = []
    for grp in groups:
        cur, used = [], 0
        for sp, _ in grp:
            ssp = ParamSpec(sp.name, shards[sp.name].shard_shape(sp.shape), sp.dtype)
            cur.append((ssp, used))
            used += _align(ssp.nbytes)
        mine.append(cur)
    return groups, mine


class WeightSyncGroup:
    """An RCCL communicator of `world_size` ranks: rank 0 is the trainer, ranks 1.. are the
    inference-worker GPUs (reference rank layout vllm1.py:71, world.py:192)."""

    def __init__(self, handle: int, rank: int, world_size: int, device: torch.device):
        self._h = ctypes.c_void_p(handle)
        self.rank = rank
        self.world_size = world_size
        self.device = device

    def comm_size(self) -> tuple[int, int]:
        """(ranks, this rank) as RCCL reports them for the communicator (ncclCommCount / ncclCommUserRank)."""
        n, r = ctypes.c_int32(), ctypes.c_int32()
        _lib.check(_lib.load().prl_wsync_comm_size(self._h, ctypes.byref(n), ctypes.byref(r)))
        return n.value, r.value

    # -- bootstrap ------------------------------------------------------------------------------
    @classmethod
    def _init(cls, uid: bytes, rank: int, world_size: int, device: torch.device) -> "WeightSyncGroup":
        lib = _lib.load()
        arr = (ctypes.c_uint8 * _lib.PRL_WSYNC_UID_BYTES).from_buffer_copy(uid)
        out = ctypes.c_void_p()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        _lib.check(lib.prl_wsync_init(arr, rank, world_size, idx, ctypes.byref(out)))
        return cls(out.value, rank, world_size, device)

    @staticmethod
    def _new_uid() -> bytes:
        lib = _lib.load()
        arr = (ctypes.c_uint8 * _lib.PRL_WSYNC_UID_BYTES)()
        _lib.check(lib.prl_wsync_unique_id(arr))
        return bytes(arr)

    @staticmethod
    def exchange_unique_id(init_method: str, rank: int, world_size: int, timeout_s: float = 300.0):
        """`tcp://host:port` rendezvous like the reference's `stateless_init_process_group`
        (torch_utils.py:70-94): rank 0 hosts a TCPStore and publishes the RCCL unique id, the others
        fetch it.  Returns (uid bytes, store); keep the store alive while peers may still join."""
        from torch.distributed import TCPStore

        u = urlparse(init_method)
        host, port = u.hostname or "127.0.0.1", u.port or 9000
        store = TCPStore(host, port, world_size, is_master=(rank == 0), timeout=datetime.timedelta(seconds=timeout_s),
                         wait_for_workers=False)
        if rank == 0:
            uid = WeightSyncGroup._new_uid()
            store.set("prl_wsync_uid", uid)
        else:
            uid = store.get("prl_wsync_uid")
        return bytes(uid), store

    @classmethod
    def from_init_method(cls, init_method: str, rank: int, world_size: int, device: torch.device,
                         timeout_s: float = 300.0) -> "WeightSyncGroup":
        uid, store = cls.exchange_unique_id(init_method, rank, world_size, timeout_s)
        grp = cls._init(uid, rank, world_size, device)
        grp._store = store  # keep the server alive for late joiners
        return grp

    @classmethod
    def tp_shard_groups(cls, init_method: str, rank: int, world_size: int, tp_size: int, device: torch.device,
                        timeout_s: float = 300.0) -> list["WeightSyncGroup"]:
        """Communicators of the TP-aware update (tp_shard.py): ONE per tensor-parallel rank t, made of the trainer
        (rank 0 in each) and the workers that hold TP rank t of their engine - global rank r >= 1 is TP rank
        (r - 1) % tp_size of engine (r - 1) // tp_size, the reference layout (vllm1.py:71).  The trainer gets all
        `tp_size` groups, a worker a one-element list.  Same `tcp://host:port` rendezvous as `from_init_method`:
        one store, one unique id per TP rank."""
        from torch.distributed import TCPStore

        if (world_size - 1) % tp_size:
            raise ValueError(f"{world_size - 1} workers do not form engines of {tp_size} TP ranks")
        n_engines = (world_size - 1) // tp_size
        u = urlparse(init_method)
        store = TCPStore(u.hostname or "127.0.0.1", u.port or 9000, world_size, is_master=(rank == 0),
                         timeout=datetime.timedelta(seconds=timeout_s), wait_for_workers=False)
        if rank == 0:
            uids = [cls._new_uid() for _ in range(tp_size)]
            for t, uid in enumerate(uids):
                store.set(f"prl_wsync_uid/tp{t}", uid)
            groups = [cls._init(uid, 0, 1 + n_engines, device) for uid in uids]  # group t completes when its workers joined
        else:
            t, engine = (rank - 1) % tp_size, (rank - 1) // tp_size
            groups = [cls._init(bytes(store.get(f"prl_wsync_uid/tp{t}")), 1 + engine, 1 + n_engines, device)]
        for g in groups:
            g._store = store
        return groups

    @classmethod
    def from_torch_distributed(cls, rank: int, world_size: int, device: torch.device, group: Any = None) -> "WeightSyncGroup":
        """Bootstrap over an existing torch.distributed group (used by bench.py and tests)."""
        import torch.distributed as dist

        holder = [cls._new_uid() if rank == 0 else None]
        dist.broadcast_object_list(holder, src=0, group=group)
        return cls._init(holder[0], rank, world_size, device)

    # -- data plane -----------------------------------------------------------------------------
    def broadcast_bucket(self, bucket: torch.Tensor, mode: str = "scatter_allgather", src: int = 0) -> None:
        """Move one contiguous byte bucket from the trainer to every worker, in place, on the
        current stream.  Every rank passes a buffer of the same size."""
        lib = _lib.load()
        assert bucket.is_cuda and bucket.is_contiguous()
        nbytes = bucket.numel() * bucket.element_size()
        stream = _lib.current_stream_ptr(bucket.device)
        if mode == "scatter_allgather" and src == 0:
            _lib.check(lib.prl_wsync_bcast_bucket_sag(self._h, bucket.data_ptr(), nbytes, stream))
        else:
            _lib.check(lib.prl_wsync_bcast_bucket(self._h, bucket.data_ptr(), nbytes, src, stream))

    def broadcast(self, tensor: torch.Tensor, src: int = 0, stream: Any = None) -> None:
        """Per-tensor broadcast with the reference communicator's signature
        (`actor_update_group.broadcast(parameter.data, src=0, stream=...)`, finetune_loop.py:238)."""
        lib = _lib.load()
        s = stream.cuda_stream if stream is not None else _lib.current_stream_ptr(tensor.device)
        t = tensor if tensor.is_contiguous() else tensor.contiguous()
        _lib.check(lib.prl_wsync_bcast_bucket(self._h, t.data_ptr(), t.numel() * t.element_size(), src, s))
        if t is not tensor:
            tensor.copy_(t)

    def close(self) -> None:
        if self._h:
            _lib.load().prl_wsync_destroy(self._h)
            self._h = ctypes.c_void_p()


class GlooWeightSyncGroup:
    """The weight-update group over gloo: the same rank layout (trainer 0, inference workers 1..) and the same `tcp://host:port`
    rendezvous as `WeightSyncGroup.from_init_method`, for hosts without a second GPU and for CPU tensors - `pipeline_run` uses it
    (`weight_transport="gloo"`) to run the N-learner x M-engine topology where RCCL cannot (RCCL refuses two ranks on one device).
    A STATELESS group like the reference's `stateless_init_process_group` (torch_utils.py:70-94): its own store and its own
    ProcessGroupGloo, independent of the process's default group (the learners' data-parallel group).  Device buffers are staged
    through host memory; this is a correctness transport, not a fast one."""

    def __init__(self, pg: Any, store: Any, rank: int, world_size: int, device: torch.device):
        self._pg, self._store = pg, store
        self.rank, self.world_size, self.device = rank, world_size, torch.device(device)
        self.bytes_moved = 0

    @classmethod
    def from_init_method(cls, init_method: str, rank: int, world_size: int, device: torch.device, timeout_s: float = 300.0) -> "GlooWeightSyncGroup":
        import torch.distributed as dist

        u = urlparse(init_method)
        timeout = datetime.timedelta(seconds=timeout_s)
        store = dist.TCPStore(u.hostname or "127.0.0.1", u.port or 9000, world_size, is_master=(rank == 0), timeout=timeout, wait_for_workers=False)
        pg = dist.ProcessGroupGloo(dist.PrefixStore("prl_wsync_gloo", store), rank, world_size, timeout)
        return cls(pg, store, rank, world_size, device)

    @classmethod
    def tp_shard_groups(cls, init_method: str, rank: int, world_size: int, tp_size: int, device: torch.device,
                        timeout_s: float = 300.0) -> list["GlooWeightSyncGroup"]:
        """The groups of the TP-aware update over gloo, same layout as `WeightSyncGroup.tp_shard_groups`: one group per tensor-parallel
        rank t = the trainer (rank 0 in each) + the workers that hold TP rank t of their engine (global rank r >= 1 is TP rank
        (r - 1) % tp_size of engine (r - 1) // tp_size, vllm1.py:71).  The trainer gets all `tp_size` groups, a worker a one-element list."""
        import torch.distributed as dist

        if (world_size - 1) % tp_size:
            raise ValueError(f"{world_size - 1} workers do not form engines of {tp_size} TP ranks")
        n_engines = (world_size - 1) // tp_size
        u = urlparse(init_method)
        timeout = datetime.timedelta(seconds=timeout_s)
        store = dist.TCPStore(u.hostname or "127.0.0.1", u.port or 9000, world_size, is_master=(rank == 0), timeout=timeout, wait_for_workers=False)

        def group(t: int, r: int) -> "GlooWeightSyncGroup":
            pg = dist.ProcessGroupGloo(dist.PrefixStore(f"prl_wsync_gloo/tp{t}", store), r, 1 + n_engines, timeout)
            return cls(pg, store, r, 1 + n_engines, device)

        if rank == 0:
            return [group(t, 0) for t in range(tp_size)]  # group t completes when its workers have joined
        t, engine = (rank - 1) % tp_size, (rank - 1) // tp_size
        return [group(t, 1 + engine)]

    def comm_size(self) -> tuple[int, int]:
        return self._pg.size(), self._pg.rank()

    def _bcast(self, t: torch.Tensor, src: int) -> None:
        import torch.distributed as dist

        opts = dist.BroadcastOptions()
        opts.rootRank, opts.rootTensor = src, 0
        if t.is_cuda:
            if self.rank == src:
                torch.cuda.current_stream(t.device).synchronize()  # the bucket was filled on this stream
                host = t.cpu()
            else:
                host = torch.empty(t.shape, dtype=t.dtype)  # nothing of the receiver's buffer needs to leave the device
            self._pg.broadcast([host], opts).wait()
            if self.rank != src:
                t.copy_(host)
        else:
            self._pg.broadcast([t], opts).wait()
        self.bytes_moved += t.numel() * t.element_size()

    def broadcast_bucket(self, bucket: torch.Tensor, mode: str = "scatter_allgather", src: int = 0) -> None:
        """`mode` is accepted for interface parity; gloo has one broadcast."""
        assert bucket.is_contiguous()
        self._bcast(bucket, src)

    def broadcast(self, tensor: torch.Tensor, src: int = 0, stream: Any = None) -> None:
        t = tensor if tensor.is_contiguous() else tensor.contiguous()
        self._bcast(t, src)
        if t is not tensor and self.rank != src:
            tensor.copy_(t)

    def close(self) -> None:
        self._pg = None


def weight_sync_group(backend: str, init_method: str, rank: int, world_size: int, device: torch.device, timeout_s: float = 300.0):
    """`rccl` -> `WeightSyncGroup` (RCCL over xGMI, one device per rank), `gloo` -> `GlooWeightSyncGroup`."""
    if backend == "rccl":
        return WeightSyncGroup.from_init_method(init_method, rank, world_size, device, timeout_s)
    if backend == "gloo":
        return GlooWeightSyncGroup.from_init_method(init_method, rank, world_size, device, timeout_s)
    raise ValueError(f"weight-update group backend {backend!r}: 'rccl' or 'gloo'")


def weight_sync_tp_groups(backend: str, init_method: str, rank: int, world_size: int, tp_size: int, device: torch.device, timeout_s: float = 300.0) -> list:
    """The per-TP-rank groups of the sharded update (`transport: sharded`) over `rccl` or `gloo`."""
    if backend == "rccl":
        return WeightSyncGroup.tp_shard_groups(init_method, rank, world_size, tp_size, device, timeout_s)
    if backend == "gloo":
        return GlooWeightSyncGroup.tp_shard_groups(init_method, rank, world_size, tp_size, device, timeout_s)
    raise ValueError(f"weight-update group backend {backend!r}: 'rccl' or 'gloo'")


_DTYPE_NAMES = {
    torch.bfloat16: ("bfloat16", "bf16"),
    torch.float32: ("float32", "fp32", "float"),
    torch.float16: ("float16", "fp16", "half"),
}
_DTYPES = {alias: dt for dt, aliases in _DTYPE_NAMES.items() for alias in aliases}


def string_to_dtype(name: str) -> torch.dtype:
    """'torch.bfloat16' -> torch.bfloat16: the dtype string travels in ParameterInfo.dtype
    (finetune_loop.py:226,273).  Accepts what the reference's receiver accepts
    (vllm_quantization.py:64-82: case-insensitive, optional `torch.` prefix, the three floating
    types and their short names) and raises ValueError for anything else."""
    key = name.lower().replace("torch.", "").strip()
    if key not in _DTYPES:
        raise ValueError(f"Unsupported dtype string: {name!r}. Supported values: {sorted(_DTYPES)}")
    return _DTYPES[key]


def _segment_table(bucket: Sequence[tuple[ParamSpec, int]], tensors: dict[str, torch.Tensor], writable: bool):
    """ctypes `prl_segment[]` for one bucket plus the contiguous temporaries made for it."""
    segs = (_lib.PrlSegment * len(bucket))()
    keep = []
    for i, (sp, off) in enumerate(bucket):
        t = tensors[sp.name]
        if t.dtype != sp.dtype or tuple(t.shape) != tuple(sp.shape):
            raise ValueError(f"{sp.name}: tensor is {t.dtype}{tuple(t.shape)}, the update announces {sp.dtype}{tuple(sp.shape)}")
        if not t.is_contiguous():
            if writable:
                raise ValueError(f"{sp.name}: destination tensor must be contiguous")
            t = t.contiguous()
            keep.append(t)
        segs[i].tensor, segs[i].bucket_offset, segs[i].nbytes = t.data_ptr(), off, sp.nbytes
    return segs, keep


def gather_into_bucket(buf: torch.Tensor, bucket: Sequence[tuple[ParamSpec, int]], tensors: dict[str, torch.Tensor]) -> None:
    """Flatten the bucket's parameters into `buf` (uint8): one `prl_bucket_gather` launch per 64
    parameters on the current stream.  Host tensors (the gloo protocol tests) use torch copies."""
    if not buf.is_cuda:
        for sp, off in bucket:
            buf[off : off + sp.nbytes].view(sp.dtype).view(sp.shape).copy_(tensors[sp.name])
        return
    segs, keep = _segment_table(bucket, tensors, writable=False)
    _lib.check(_lib.load().prl_bucket_gather(buf.data_ptr(), buf.numel(), segs, len(bucket), _lib.current_stream_ptr(buf.device)))
    for t in keep:  # temporaries made contiguous above: free only after the copy on this stream
        t.record_stream(torch.cuda.current_stream(buf.device))


def scatter_from_bucket(buf: torch.Tensor, bucket: Sequence[tuple[ParamSpec, int]], tensors: dict[str, torch.Tensor]) -> None:
    """Copy every slot of `buf` into the same-named destination tensor (the engine's own weights)
    with one `prl_bucket_scatter` launch per 64 parameters."""
    if not buf.is_cuda:
        for sp, off in bucket:
            tensors[sp.name].copy_(buf[off : off + sp.nbytes].view(sp.dtype).view(sp.shape))
        return
    segs, _ = _segment_table(bucket, tensors, writable=True)
    _lib.check(_lib.load().prl_bucket_scatter(buf.data_ptr(), buf.numel(), segs, len(bucket), _lib.current_stream_ptr(buf.device)))


def _deliver(buf: torch.Tensor, bucket: Sequence[tuple[ParamSpec, int]], load_weights: Callable | None,
             destinations: dict[str, torch.Tensor] | None) -> int:
    """Hand one received bucket to the engine: parameters with a registered same-dtype/shape
    destination are scattered by the copy kernel, the rest go through `load_weights` as views."""
    direct = []
    if destinations:
        direct = [(sp, off) for sp, off in bucket
                  if (d := destinations.get(sp.name)) is not None and d.dtype == sp.dtype and tuple(d.shape) == tuple(sp.shape) and d.is_contiguous()]
        if direct:
            scatter_from_bucket(buf, direct, destinations)
    if len(direct) < len(bucket):
        taken = {sp.name for sp, _ in direct}
        views = [(sp.name, buf[off : off + sp.nbytes].view(sp.dtype).view(sp.shape)) for sp, off in bucket if sp.name not in taken]
        if load_weights is None:
            raise ValueError(f"no destination registered for {[n for n, _ in views][:4]}... and no load_weights callback")
        load_weights(views)
    return len(bucket)


class _TwoStreamPipe:
    """Event plumbing shared by the bucketed sender and receiver on a GPU: the RCCL transfers run on
    their own HIP stream, the flatten / unflatten copies on the caller's stream, and two staging
    buffers alternate - so bucket k+1 is on the wire while bucket k is being copied.
    `wire(k, fn)` enqueues a transfer, `local(k, fn)` the copy work of bucket k."""

    def __init__(self, device: torch.device):
        self.enabled = torch.device(device).type == "cuda"
        if self.enabled:
            self.comm = torch.cuda.Stream(device)
            self.main = torch.cuda.current_stream(device)
            self.wire_done: dict[int, torch.cuda.Event] = {}
            self.local_done: dict[int, torch.cuda.Event] = {}
            start = torch.cuda.Event()
            start.record(self.main)
            self.comm.wait_event(start)  # everything queued before the update stays ahead of it

    def wire(self, k: int, fn: Callable[[], None], after_local: int | None) -> None:
        if not self.enabled:
            fn()
            return
        if after_local is not None and after_local in self.local_done:
            self.comm.wait_event(self.local_done[after_local])
        with torch.cuda.stream(self.comm):
            fn()
            ev = torch.cuda.Event()
            ev.record(self.comm)
        self.wire_done[k] = ev

    def local(self, k: int, fn: Callable[[], Any], after_wire: int | None) -> Any:
        if not self.enabled:
            return fn()
        if after_wire is not None and after_wire in self.wire_done:
            self.main.wait_event(self.wire_done[after_wire])
        out = fn()
        ev = torch.cuda.Event()
        ev.record(self.main)
        self.local_done[k] = ev
        return out

    def finish(self) -> None:
        """The caller's stream continues only after the last transfer."""
        if self.enabled and self.wire_done:
            self.main.wait_event(self.wire_done[max(self.wire_done)])


class BucketedSender:
    """Trainer side: flatten named parameters into reusable device buckets and broadcast them."""

    def __init__(self, group: WeightSyncGroup, bucket_bytes: int = DEFAULT_BUCKET_BYTES, mode: str = "scatter_allgather"):
        self.group = group
        self.bucket_bytes = bucket_bytes
        self.mode = mode
        self._staging: list[torch.Tensor] = []

    def send(self, named_parameters: Iterable[tuple[str, torch.Tensor]]) -> list[ParamSpec]:
        params = [(n, p.detach()) for n, p in named_parameters]
        specs = [ParamSpec(n, tuple(p.shape), p.dtype) for n, p in params]
        plan = plan_buckets(specs, self.bucket_bytes)
        cap = max(bucket_nbytes(b) for b in plan)
        if len(self._staging) < 2 or self._staging[0].numel() < cap:  # double buffer: flatten bucket k+1 while bucket k is on the wire
            self._staging = [torch.empty(cap, dtype=torch.uint8, device=self.group.device) for _ in range(2)]
        tensors = dict(params)
        pipe = _TwoStreamPipe(self.group.device)
        for k, bucket in enumerate(plan):
            buf = self._staging[k % 2][: bucket_nbytes(bucket)]
            # flatten k (needs staging[k % 2] back from transfer k - 2), then put it on the wire
            pipe.local(k, lambda buf=buf, bucket=bucket: gather_into_bucket(buf, bucket, tensors), after_wire=k - 2)
            pipe.wire(k, lambda buf=buf: self.group.broadcast_bucket(buf, mode=self.mode), after_local=k)
        pipe.finish()
        return specs


    def send_streamed(self, specs: Sequence[ParamSpec], fetch: Callable[[list[ParamSpec]], Any]) -> list[ParamSpec]:
        """As `send`, for trainers whose parameters are SHARDED (ZeRO-3, FSDP): the full tensors of a bucket
        exist only inside `with fetch(bucket_specs) as tensors:` (a collective gather among the trainer
        ranks), they are flattened into the staging buffer there and released again - at no point is more
        than one bucket of full parameters alive, where the reference gathers and sends parameter by
        parameter (finetune_loop.py:230-238)."""
        plan = plan_buckets(list(specs), self.bucket_bytes)
        cap = max(bucket_nbytes(b) for b in plan)
        if len(self._staging) < 2 or self._staging[0].numel() < cap:
            self._staging = [torch.empty(cap, dtype=torch.uint8, device=self.group.device) for _ in range(2)]
        pipe = _TwoStreamPipe(self.group.device)
        for k, bucket in enumerate(plan):
            buf = self._staging[k % 2][: bucket_nbytes(bucket)]
            with fetch([sp for sp, _ in bucket]) as tensors:
                pipe.local(k, lambda buf=buf, bucket=bucket, tensors=tensors: gather_into_bucket(buf, bucket, tensors), after_wire=k - 2)
            pipe.wire(k, lambda buf=buf: self.group.broadcast_bucket(buf, mode=self.mode), after_local=k)
        pipe.finish()
        return list(specs)


class ShardedSender:
    """Trainer side of the TP-aware update (the reference sends every full tensor to every TP rank,
    vllm1.py:110-127): `groups[t]` reaches the workers that hold TP rank t, and carries only their slices.
    Per group of parameters the full tensors are fetched ONCE (a collective for ZeRO-3 / FSDP trainers), each TP
    rank's slices are flattened into its own staging buffer, and the `tp_size` transfers run concurrently on their
    own streams - different peers, different xGMI links.  Bytes per worker: S / tp_size (+ replicated norms)
    instead of S."""

    def __init__(self, groups: Sequence[WeightSyncGroup], bucket_bytes: int = DEFAULT_BUCKET_BYTES, mode: str = "scatter_allgather"):
        self.groups = list(groups)
        self.tp_size = len(self.groups)
        self.bucket_bytes = bucket_bytes
        self.mode = mode
        self._staging: list[list[torch.Tensor]] = []
        self.bytes_sent = [0] * self.tp_size  # per TP rank, last update

    def send_streamed(self, specs: Sequence[ParamSpec], shards: dict, fetch: Callable[[list[ParamSpec]], Any]) -> list[ParamSpec]:
        from .tp_shard import shard_view

        tp, dev = self.tp_size, self.groups[0].device
        full_groups = None
        plans = []
        for t in range(tp):
            full_groups, mine = plan_shard_buckets(list(specs), shards, t, tp, self.bucket_bytes)
            plans.append(mine)
        caps = [max(bucket_nbytes(b) for b in plans[t]) for t in range(tp)]
        if not self._staging or any(self._staging[t][0].numel() < caps[t] for t in range(tp)):
            self._staging = [[torch.empty(caps[t], dtype=torch.uint8, device=dev) for _ in range(2)] for t in range(tp)]
        pipes = [_TwoStreamPipe(dev) for _ in range(tp)]
        self.bytes_sent = [0] * tp
        for k, grp in enumerate(full_groups):
            bufs = []
            with fetch([sp for sp, _ in grp]) as tensors:
                for t in range(tp):
                    bucket = plans[t][k]
                    buf = self._staging[t][k % 2][: bucket_nbytes(bucket)]
                    views = {sp.name: shard_view(tensors[sp.name], shards[sp.name], t, tp) for sp, _ in grp}
                    pipes[t].local(k, lambda buf=buf, bucket=bucket, views=views: gather_into_bucket(buf, bucket, views), after_wire=k - 2)
                    bufs.append(buf)
                    self.bytes_sent[t] += buf.numel()
            for t in range(tp):
                pipes[t].wire(k, lambda t=t, buf=bufs[t]: self.groups[t].broadcast_bucket(buf, mode=self.mode), after_local=k)
        for p in pipes:
            p.finish()
        return list(specs)

    def send(self, named_parameters: Iterable[tuple[str, torch.Tensor]], shards: dict) -> list[ParamSpec]:
        import contextlib

        params = {n: p.detach() for n, p in named_parameters}
        specs = [ParamSpec(n, tuple(p.shape), p.dtype) for n, p in params.items()]
        return self.send_streamed(specs, shards, lambda wanted: contextlib.nullcontext({sp.name: params[sp.name] for sp in wanted}))


class BucketedReceiver:
    """Worker side: receive the buckets implied by `parameters_info` and hand (name, tensor) views
    to `load_weights` (or scatter them into registered destinations), bucket by bucket; transfers
    run on their own stream, so bucket k + 1 arrives while bucket k is being unflattened."""

    def __init__(self, group: WeightSyncGroup, bucket_bytes: int = DEFAULT_BUCKET_BYTES, mode: str = "scatter_allgather"):
        self.group = group
        self.bucket_bytes = bucket_bytes
        self.mode = mode
        self._staging: list[torch.Tensor] = []

    def receive(self, parameters_info: Sequence[dict | ParamSpec], load_weights: Callable[[list[tuple[str, torch.Tensor]]], Any] | None,
                destinations: dict[str, torch.Tensor] | None = None) -> int:
        specs = [
            p if isinstance(p, ParamSpec) else ParamSpec(p["name"], tuple(p["shape"]), string_to_dtype(p["dtype"]))
            for p in parameters_info
        ]
        return self.receive_planned(plan_buckets(specs, self.bucket_bytes), load_weights, destinations)

    def receive_planned(self, plan: Sequence[Sequence[tuple[ParamSpec, int]]],
                        load_weights: Callable[[list[tuple[str, torch.Tensor]]], Any] | None,
                        destinations: dict[str, torch.Tensor] | None = None) -> int:
        """`receive` for a bucket plan made elsewhere (the TP-aware update: `plan_shard_buckets`)."""
        cap = max(bucket_nbytes(b) for b in plan)
        if not self._staging or self._staging[0].numel() < cap:
            self._staging = [torch.empty(cap, dtype=torch.uint8, device=self.group.device) for _ in range(2)]
        n = 0
        pipe = _TwoStreamPipe(self.group.device)
        for k, bucket in enumerate(plan):
            buf = self._staging[k % 2][: bucket_nbytes(bucket)]
            # receive k (staging[k % 2] is free once bucket k - 2 has been handed out), then unflatten it
            # on the caller's stream while bucket k + 1 is already arriving on the transfer stream
            pipe.wire(k, lambda buf=buf: self.group.broadcast_bucket(buf, mode=self.mode), after_local=k - 2)
            n += pipe.local(k, lambda buf=buf, bucket=bucket: _deliver(buf, bucket, load_weights, destinations), after_wire=k)
        pipe.finish()
        return n


# ---------------------------------------------------------------------------------------------
# Colocated hand-off over HIP IPC (trainer and inference worker share ONE GPU)
# ---------------------------------------------------------------------------------------------


class _RawDeviceMemory:
    """Exposes a raw device pointer through the CUDA array interface so torch can view it."""

    def __init__(self, ptr: int, nbytes: int, owner: Any):
        self._owner = owner  # keeps the allocation / mapping alive as long as a tensor views it
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class DeviceBucket:
    """A hipMalloc'ed byte bucket whose memory can be exported to another process on the same GPU."""

    def __init__(self, nbytes: int, device: torch.device):
        lib = _lib.load()
        self.nbytes = int(nbytes)
        self.device = device
        out = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.prl_ipc_alloc(self.nbytes, ctypes.byref(out)))
        self.ptr = out.value
        self._tensor = torch.as_tensor(_RawDeviceMemory(self.ptr, self.nbytes, self), device=device)

    def tensor(self) -> torch.Tensor:
        return self._tensor

    def handle(self) -> bytes:
        arr = (ctypes.c_uint8 * _lib.PRL_IPC_HANDLE_BYTES)()
        _lib.check(_lib.load().prl_ipc_export(self.ptr, arr))
        return bytes(arr)

    def free(self) -> None:
        if self.ptr:
            self._tensor = None
            _lib.load().prl_ipc_free(self.ptr)
            self.ptr = None


class MappedBucket:
    """The peer's view of a `DeviceBucket` (hipIpcOpenMemHandle)."""

    def __init__(self, handle: bytes, nbytes: int, device: torch.device):
        lib = _lib.load()
        arr = (ctypes.c_uint8 * _lib.PRL_IPC_HANDLE_BYTES).from_buffer_copy(handle)
        out = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(lib.prl_ipc_open(arr, ctypes.byref(out)))
        self.ptr, self.nbytes, self.device = out.value, int(nbytes), device
        self._tensor = torch.as_tensor(_RawDeviceMemory(self.ptr, self.nbytes, self), device=device)

    def tensor(self) -> torch.Tensor:
        return self._tensor

    def close(self) -> None:
        if self.ptr:
            self._tensor = None
            _lib.load().prl_ipc_close(self.ptr)
            self.ptr = None


class ColocatedSender:
    """Trainer side when the inference worker lives on the same GPU: flatten the parameters into
    IPC-exportable device buckets and describe them; nothing travels over any link."""

    def __init__(self, device: torch.device, bucket_bytes: int = DEFAULT_BUCKET_BYTES, max_allocation: int = IPC_MAX_ALLOCATION):
        self.device = device
        self.bucket_bytes = bucket_bytes
        self.max_allocation = int(max_allocation)
        self._buckets: list[DeviceBucket] = []

    def _pieces(self, params: list[tuple[str, torch.Tensor]]) -> list[tuple[str, torch.Tensor]]:
        """`params` with every tensor too large for one exportable allocation replaced by views of its row ranges (`split_for_ipc`)."""
        tensors = dict(params)
        dense: dict[str, torch.Tensor] = {}  # a non-contiguous base tensor is made contiguous ONCE, not once per piece
        out = []
        for sp in split_for_ipc([ParamSpec(n, tuple(p.shape), p.dtype) for n, p in params], self.bucket_bytes, self.max_allocation):
            piece = rows_piece(sp.name)
            if piece is not None and sp.name not in tensors:
                base, a, b = piece
                if base not in dense:
                    t = tensors[base]
                    dense[base] = t if t.is_contiguous() else t.contiguous()
                out.append((sp.name, dense[base][a:b]))
            else:
                out.append((sp.name, tensors[sp.name]))
        return out

    def _ensure_buckets(self, params: list[tuple[str, torch.Tensor]]):
        specs = [ParamSpec(n, tuple(p.shape), p.dtype) for n, p in params]
        plan = plan_buckets(specs, self.bucket_bytes)
        sizes = [bucket_nbytes(b) for b in plan]
        if any(n >= self.max_allocation for n in sizes):
            raise ValueError(f"an IPC bucket of {max(sizes)} bytes: allocations of {self.max_allocation} bytes or more cannot be opened by the peer")
        if [b.nbytes for b in self._buckets] != sizes:
            for b in self._buckets:
                b.free()
            self._buckets = [DeviceBucket(n, self.device) for n in sizes]
        return plan, sizes

    def rehome(self, named_parameters: Iterable[tuple[str, torch.nn.Parameter]]) -> None:
        """Move the parameters' storage INTO the exported buckets (`p.data` becomes a bucket view), so
        every later `publish` is zero-copy: the optimizer updates the shared memory in place.
        Call once after the model is built (before optimizer state captures `p.data` pointers)."""
        params = list(named_parameters)
        plan, _ = self._ensure_buckets(self._pieces([(n, p.data) for n, p in params]))
        by_name = dict(params)
        for bucket, dev_bucket in zip(plan, self._buckets):
            buf = dev_bucket.tensor()
            for sp, off in bucket:
                if sp.name not in by_name:
                    continue  # a row range of a tensor too large for one allocation: it stays where it is and is copied per update
                view = buf[off : off + sp.nbytes].view(sp.dtype).view(sp.shape)
                view.copy_(by_name[sp.name].data)
                by_name[sp.name].data = view
        torch.cuda.synchronize(self.device)

    def publish(self, named_parameters: Iterable[tuple[str, torch.Tensor]]) -> dict:
        """Returns {"ipc_handles": [hex...], "ipc_nbytes": [...]} for the update request; the buckets
        are complete (device-synchronised) when this returns.  Parameters that already live in their
        bucket slot (`rehome`) are not copied."""
        params = self._pieces([(n, p.detach()) for n, p in named_parameters])
        plan, sizes = self._ensure_buckets(params)
        tensors = dict(params)
        for bucket, dev_bucket in zip(plan, self._buckets):
            base = dev_bucket.ptr
            moving = [(sp, off) for sp, off in bucket
                      if not (tensors[sp.name].data_ptr() == base + off and tensors[sp.name].is_contiguous())]
            if moving:
                gather_into_bucket(dev_bucket.tensor(), moving, tensors)
        torch.cuda.synchronize(self.device)
        return {"ipc_handles": [b.handle().hex() for b in self._buckets], "ipc_nbytes": sizes, "ipc_max_allocation": self.max_allocation}

    def close(self) -> None:
        for b in self._buckets:
            b.free()
        self._buckets = []


class ColocatedReceiver:
    """Worker side: map the trainer's buckets (once per handle) and hand (name, view) pairs to
    `load_weights`, which copies them into the engine's own weights device-to-device."""

    def __init__(self, device: torch.device, bucket_bytes: int = DEFAULT_BUCKET_BYTES, max_allocation: int = IPC_MAX_ALLOCATION):
        self.device = device
        self.bucket_bytes = bucket_bytes
        self.max_allocation = int(max_allocation)
        self._mapped: dict[str, MappedBucket] = {}

    def receive(self, parameters_info: Sequence[dict | ParamSpec], ipc_handles: Sequence[str], ipc_nbytes: Sequence[int],
                load_weights: Callable[[list[tuple[str, torch.Tensor]]], Any] | None,
                destinations: dict[str, torch.Tensor] | None = None, max_allocation: int | None = None) -> int:
        """`max_allocation`: the SENDER's cap as announced in the request (`WeightUpdateRequest.ipc_max_allocation`); the piece
        list is derived with it so that both ends cut a large tensor into the same row ranges.  None (a request that predates
        the field) falls back to this receiver's own setting; a mismatch that survives is caught by the size check below."""
        specs = [
            p if isinstance(p, ParamSpec) else ParamSpec(p["name"], tuple(p["shape"]), string_to_dtype(p["dtype"]))
            for p in parameters_info
        ]
        full = {sp.name: sp for sp in specs}
        specs = split_for_ipc(specs, self.bucket_bytes, self.max_allocation if max_allocation is None else int(max_allocation))
        plan = plan_buckets(specs, self.bucket_bytes)
        if len(plan) != len(ipc_handles):
            raise ValueError(f"{len(ipc_handles)} IPC buckets announced, the parameter list implies {len(plan)}")
        implied = [bucket_nbytes(b) for b in plan]
        if list(ipc_nbytes) != implied:  # same bucket count, different cuts: the weights would be scattered to the wrong rows
            raise ValueError(f"IPC bucket sizes {list(ipc_nbytes)[:4]}... announced, the parameter list implies {implied[:4]}... "
                             "(sender and receiver disagree on bucket_bytes / ipc_max_allocation)")
        # row ranges of a tensor that travelled in pieces: straight into the rows of its destination when one is registered,
        # otherwise collected and handed to `load_weights` as ONE tensor once every piece is there
        pieces = {sp.name: rows_piece(sp.name) for sp in specs if sp.name not in full}
        dest = dict(destinations or {})
        for name, (base, a, b) in pieces.items():
            d = (destinations or {}).get(base)
            if d is not None and d.dtype == full[base].dtype and tuple(d.shape) == tuple(full[base].shape) and d.is_contiguous():
                dest[name] = d[a:b]
        parts: dict[str, list] = {}

        def deliver_views(views):
            whole = []
            for name, v in views:
                if name in pieces:
                    base, a, _ = pieces[name]
                    parts.setdefault(base, []).append((a, v))
                else:
                    whole.append((name, v))
            if whole:
                if load_weights is None:
                    raise ValueError(f"no destination registered for {[n for n, _ in whole][:4]}... and no load_weights callback")
                load_weights(whole)

        n = 0
        for bucket, hx, nb in zip(plan, ipc_handles, ipc_nbytes):
            mapped = self._mapped.get(hx)
            if mapped is None:
                mapped = self._mapped[hx] = MappedBucket(bytes.fromhex(hx), nb, self.device)
            n += _deliver(mapped.tensor(), bucket, deliver_views, dest)
        for base, got in parts.items():
            if load_weights is None:
                raise ValueError(f"no destination registered for {base} and no load_weights callback")
            load_weights([(base, torch.cat([v for _, v in sorted(got, key=lambda x: x[0])]))])
        torch.cuda.synchronize(self.device)  # the trainer may overwrite the buckets after the ack
        return n - len(pieces) + len({p[0] for p in pieces.values()})

    def close(self) -> None:
        for m in self._mapped.values():
            m.close()
        self._mapped = {}
