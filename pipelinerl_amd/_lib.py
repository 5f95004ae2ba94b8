"""ctypes binding of libprl.so (the C ABI declared in include/prl.h).

The product path has NO fallback: if the shared object is missing, or a device entry point is
called without a HIP device, this module raises.  Build with `python -m pipelinerl_amd.build`.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_size_t, c_uint8, c_uint32, c_uint64, c_void_p
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libprl.so"

PRL_ABI_VERSION = 13
PRL_OK = 0
PRL_EINVAL = -22
PRL_ENOMEM = -12
PRL_EFAULT = -14
PRL_EAGAIN = -11
PRL_ETIMEDOUT = -110
PRL_ENOSYS = -38
PRL_EMSGSIZE = -90

PRL_DTYPE_F32 = 0
PRL_DTYPE_BF16 = 1
PRL_POLICY_PPO = 0
PRL_POLICY_REINFORCE = 1
PRL_POLICY_GSPO = 2
PRL_FINISH_NONE = 0
PRL_FINISH_LENGTH = 1
PRL_FINISH_STOP = 2
PRL_NUM_STATS = 32
PRL_NUM_VALUE_STATS = 5
PRL_WSYNC_UID_BYTES = 128
PRL_IPC_HANDLE_BYTES = 64
PRL_LM_HEAD_DH_LEADING_TERM = 1
PRL_LM_HEAD_DW_OVERWRITE = 2
PRL_LM_HEAD_DH_NO_WEIGHT_LO = 4
PRL_LOG_CREATE, PRL_LOG_TRUNCATE, PRL_LOG_READER, PRL_LOG_TRIM = 1, 2, 4, 8

# index of every public statistic in the device stats vector (enum in include/prl.h)
STAT_INDEX = {
    "loss": 0,
    "num_output_tokens_sum": 1,
    "num_sequences": 2,
    "reward": 3,
    "max_reward": 4,
    "min_reward": 5,
    "entropy": 6,
    "old_logprobs": 7,
    "new_logprobs": 8,
    "ref_logprobs": 9,
    "advantage": 10,
    "max_advantage": 11,
    "min_advantage": 12,
    "kl": 13,
    "kl_new_old": 14,
    "mean_abs_log_ratio_new_old": 15,
    "max_kl": 16,
    "min_kl": 17,
    "ratio_new_old": 18,
    "ratio_new_old_sum": 19,
    "ratio_new_old_squared_sum": 20,
    "ratio_ref_new": 21,
    "ratio_ref_old": 22,
    "clamp_log_ratio_ref_new_indicator": 23,
    "clamp_log_ratio_new_old_indicator": 24,
    "token_weight": 25,
    "max_token_weight": 26,
    "min_token_weight": 27,
    "nonfinite_new_logprobs": 28,
    "nonfinite_log_ratio_ref_new": 29,
    "nonfinite_kl": 30,
    "bad_group_tokens": 31,
}


class PrlLogIov(ctypes.Structure):
    """`prl_log_iov` of include/prl.h: one source range of a gathered record."""

    _fields_ = [("ptr", ctypes.c_void_p), ("offset", ctypes.c_uint64), ("nbytes", ctypes.c_uint64)]


PRL_PUB_FROM_BLOCK, PRL_PUB_INLINE, PRL_PUB_FROM_HOST = 0, 1, 2  # prl_pub_piece.kind


class PrlPubPiece(ctypes.Structure):
    """`prl_pub_piece` of include/prl.h."""

    _fields_ = [("src", ctypes.c_uint64), ("offset", ctypes.c_uint64), ("nbytes", ctypes.c_uint64), ("kind", ctypes.c_uint32), ("_pad", ctypes.c_uint32)]


class PrlPubRecord(ctypes.Structure):
    """`prl_pub_record` of include/prl.h."""

    _fields_ = [("log", ctypes.c_void_p), ("nbytes", ctypes.c_uint64), ("first_piece", ctypes.c_uint32), ("n_pieces", ctypes.c_uint32)]


class PrlLossConfig(ctypes.Structure):
    """Mirror of `struct prl_loss_config` (include/prl.h)."""

    _fields_ = [
        ("policy_loss", c_int32),
        ("use_advantages", c_int32),
        ("relu_log_p_weights", c_int32),
        ("group_normalization", c_int32),
        ("overlong_filtering", c_int32),
        ("use_entropy_loss", c_int32),
        ("flat_micro_batches", c_int32),
        ("skip_unlabelled", c_int32),
        ("token_weight", c_float),
        ("clip_lo", c_float),
        ("clip_hi", c_float),
        ("kl_coef", c_float),
        ("entropy_coef", c_float),
        ("clamp_log_ratio_ref_new", c_float),
        ("upstream_scale", c_float),
    ]


class PrlSegment(ctypes.Structure):
    """Mirror of `struct prl_segment` (include/prl.h)."""

    _fields_ = [("tensor", c_void_p), ("bucket_offset", c_int64), ("nbytes", c_int64)]


class PrlError(RuntimeError):
    def __init__(self, code: int, message: str):
        super().__init__(f"libprl error {code}: {message}")
        self.code = code


_P = c_void_p  # every device / host buffer pointer crosses the ABI as void*

# name -> (restype, argtypes).  This table is also what tests/test_abi.py checks against
# the declarations in include/prl.h.
PROTOTYPES: dict[str, tuple] = {
    "prl_abi_version": (c_int32, []),
    "prl_set_tuning": (c_int32, [c_int32, c_int64]),
    "prl_get_tuning": (c_int32, [c_int32, POINTER(c_int64)]),
    "prl_last_error": (c_char_p, []),
    "prl_logprob_entropy_fwd": (c_int32, [c_int64, c_int64, c_int64, _P, c_int32, c_int64, _P, c_float, _P, _P, _P, _P]),
    "prl_logprob_entropy_bwd": (c_int32, [c_int64, c_int64, c_int64, _P, c_int32, c_int64, _P, c_float, _P, _P, _P, _P, _P, _P, _P]),
    "prl_grpo_loss_workspace_bytes": (c_int32, [c_int64, c_int64, POINTER(c_size_t)]),
    "prl_grpo_loss_fwd_bwd": (c_int32, [POINTER(PrlLossConfig), c_int64, c_int64] + [_P] * 13 + [_P, _P, _P, _P, _P, c_size_t, _P]),
    "prl_fused_logits_loss": (c_int32, [POINTER(PrlLossConfig), c_int64, c_int64, c_int64, _P, c_int32, c_int64, c_float] + [_P] * 8 + [_P, _P, _P, _P, _P]),
    "prl_last_fused_kernel": (c_char_p, []),
    "prl_scale_unless": (c_int32, [_P, c_int64, c_int32, _P, c_float, _P]),
    "prl_segment_sums": (c_int32, [c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P]),
    "prl_gspo_segment_sums": (c_int32, [_P, c_int64, c_int32, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "prl_gspo_segment_terms": (c_int32, [_P, c_int32, _P, c_float, c_int32, _P, _P, _P, _P]),
    "prl_gspo_expand": (c_int32, [c_int64, c_int32, _P, _P, _P, _P, _P, _P]),
    "prl_value_head_workspace_bytes": (c_int32, [c_int64, c_int64, POINTER(c_size_t)]),
    "prl_value_head_fwd_bwd": (c_int32, [POINTER(PrlLossConfig), c_int64, c_int64, _P, _P, c_int32] + [_P] * 8 + [_P, c_size_t, _P]),
    "prl_seq_scan": (c_int32, [c_int32, _P, _P, _P, _P, _P, c_int32, _P, _P, _P]),
    "prl_patch_oov": (c_int32, [c_int64, _P, _P, c_int32, c_int32, _P, _P]),
    "prl_group_advantages": (c_int32, [c_int32, c_int32, c_int32, _P, _P, _P, _P, _P, _P, _P, c_int32, _P, _P, _P, _P, _P]),
    "prl_pack_collate": (c_int32, [c_int32, c_int64] + [_P] * 14 + [c_int32, c_int32] + [_P] * 12 + [_P]),
    "prl_pad_collate": (c_int32, [c_int32, c_int64, c_int32] + [_P] * 12 + [c_int32] + [_P] * 10 + [_P]),
    "prl_ring_create": (c_int32, [c_char_p, c_uint32, c_uint64, POINTER(c_void_p)]),
    "prl_ring_attach": (c_int32, [c_char_p, POINTER(c_void_p)]),
    "prl_ring_put": (c_int32, [c_void_p, c_void_p, c_uint64, c_int64]),
    "prl_ring_get": (c_int32, [c_void_p, c_void_p, c_uint64, POINTER(c_uint64), c_int64]),
    "prl_ring_reserve": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_uint64), c_int64]),
    "prl_ring_commit": (c_int32, [c_void_p, c_uint64, c_uint64]),
    "prl_ring_acquire": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_uint64), POINTER(c_uint64), c_int64]),
    "prl_ring_release": (c_int32, [c_void_p, c_uint64]),
    "prl_ring_size": (c_int32, [c_void_p, POINTER(c_uint64)]),
    "prl_ring_capacity": (c_int32, [c_void_p, POINTER(c_uint32), POINTER(c_uint64)]),
    "prl_ring_max_record_bytes": (c_int32, [c_void_p, POINTER(c_uint64)]),
    "prl_ring_close": (c_int32, [c_void_p]),
    "prl_ring_detach": (c_int32, [c_void_p]),
    "prl_ring_unlink": (c_int32, [c_char_p]),
    "prl_log_open": (c_int32, [c_char_p, c_uint64, c_int32, POINTER(c_void_p)]),
    "prl_log_append": (c_int32, [c_void_p, c_void_p, c_uint64]),
    "prl_publisher_create": (c_int32, [c_int32, POINTER(c_void_p)]),
    "prl_publisher_submit": (c_int32, [c_void_p, c_void_p, c_uint64, c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_uint64, POINTER(c_uint64)]),
    "prl_publisher_completed": (c_int32, [c_void_p, POINTER(c_uint64)]),
    "prl_publisher_wait": (c_int32, [c_void_p, c_uint64, c_int64]),
    "prl_publisher_stats": (c_int32, [c_void_p, POINTER(c_uint64), POINTER(c_uint64)]),
    "prl_publisher_destroy": (c_int32, [c_void_p]),
    "prl_log_appendv": (c_int32, [c_void_p, c_void_p, c_int32, c_uint64]),
    "prl_log_read": (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_uint64), c_int64]),
    "prl_log_stats": (c_int32, [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    "prl_log_close": (c_int32, [c_void_p]),
    "prl_log_unlink": (c_int32, [c_char_p]),
    "prl_wsync_unique_id": (c_int32, [POINTER(c_uint8)]),
    "prl_wsync_init": (c_int32, [POINTER(c_uint8), c_int32, c_int32, c_int32, POINTER(c_void_p)]),
    "prl_wsync_bcast_bucket": (c_int32, [c_void_p, c_void_p, c_uint64, c_int32, c_void_p]),
    "prl_wsync_bcast_bucket_sag": (c_int32, [c_void_p, c_void_p, c_uint64, c_void_p]),
    "prl_wsync_comm_size": (c_int32, [c_void_p, POINTER(c_int32), POINTER(c_int32)]),
    "prl_wsync_destroy": (c_int32, [c_void_p]),
    "prl_ipc_alloc": (c_int32, [c_uint64, POINTER(c_void_p)]),
    "prl_ipc_free": (c_int32, [c_void_p]),
    "prl_ipc_export": (c_int32, [c_void_p, POINTER(c_uint8)]),
    "prl_ipc_open": (c_int32, [POINTER(c_uint8), POINTER(c_void_p)]),
    "prl_ipc_close": (c_int32, [c_void_p]),
    "prl_bucket_gather": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "prl_bucket_scatter": (c_int32, [c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "prl_lm_head_prepare": (c_int32, [c_int64, c_int64, _P, c_int32, _P, _P, _P, _P, _P]),
    "prl_lm_head_workspace_bytes": (c_int32, [c_int64, c_int64, c_int64, c_int64, c_int64, POINTER(c_size_t), POINTER(c_size_t)]),
    "prl_lm_head_logprob_fwd": (c_int32, [c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P, c_float, _P, _P, _P, _P, c_size_t, _P]),
    "prl_lm_head_logprob_bwd": (c_int32, [c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P,
                                          c_int32, _P, c_int64, c_int32, _P, c_size_t, _P]),
    "prl_lm_head_logprob_fwd_keep": (c_int32, [c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, c_size_t, _P]),
    "prl_lm_head_logprob_bwd_kept": (c_int32, [c_int64, c_int64, c_int64, c_int64, _P, _P, _P, _P, _P, c_float, _P, _P, _P, _P, _P, _P,
                                               c_int32, _P, c_int64, c_int32, _P, c_size_t, _P]),
}

_lib: ctypes.CDLL | None = None


def load() -> ctypes.CDLL:
    """Load libprl.so once, set prototypes, verify the ABI version.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("PRL_LIB", LIB_PATH))
    if not path.exists():
        raise ImportError(
            f"{path} not found: the HIP extension is not built. Run `python -m pipelinerl_amd.build` "
            "(needs hipcc). There is no CPU fallback for the pipelinerl_amd hot path."
        )
    # torch first: its bundled HIP runtime / RCCL must be the ones the process uses.
    import torch  # noqa: F401

    # ONE RCCL per process: the weight-sync entry points dlopen RCCL at first use (`PRL_RCCL_LIB`, then the loader's search path).
    # torch ships its own librccl.so next to libtorch; name it, so that a search-path hit on another copy (/opt/rocm/lib) cannot
    # put a second RCCL runtime beside the one torch.distributed's "nccl" backend already initialised.
    if "PRL_RCCL_LIB" not in os.environ:
        bundled = Path(torch.__file__).resolve().parent / "lib" / "librccl.so"
        if bundled.exists():
            os.environ["PRL_RCCL_LIB"] = str(bundled)

    lib = ctypes.CDLL(str(path), mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.prl_abi_version()
    if got != PRL_ABI_VERSION:
        raise ImportError(f"libprl.so ABI version {got} != expected {PRL_ABI_VERSION}; rebuild")
    for name in _ENV_TUNED_ENTRY_POINTS:
        setattr(lib, name, _with_env_tuning(lib, getattr(lib, name)))
    _lib = lib
    return lib


# ---- diagnostic launch overrides -----------------------------------------------------------------------
# The C library reads an integer table (prl_set_tuning), never the environment.  The PRL_* variables the
# measurement scripts and tests use are mapped onto it HERE, on the Python side of the entry points they
# influence: a dictionary lookup per call in the host language, nothing on the C launch path.
TUNE_KEYS = {"fused_variant": 0, "lmhead_tile": 1, "lmhead_nsplit": 2, "lmhead_ksplit": 3}
PRL_TUNE_UNSET = -(1 << 63)
_ENV_OF_KEY = {"fused_variant": "PRL_FUSED_VARIANT", "lmhead_tile": "PRL_LMHEAD_TILE", "lmhead_nsplit": "PRL_LMHEAD_NSPLIT",
               "lmhead_ksplit": "PRL_LMHEAD_KSPLIT"}
_ENV_TUNED_ENTRY_POINTS = ("prl_fused_logits_loss", "prl_lm_head_logprob_fwd", "prl_lm_head_logprob_bwd", "prl_lm_head_workspace_bytes",
                           "prl_lm_head_logprob_fwd_keep", "prl_lm_head_logprob_bwd_kept")
_env_seen: tuple | None = None


def set_tuning(key: str, value: int | None) -> None:
    """`value` None restores the library's own choice."""
    check(load().prl_set_tuning(TUNE_KEYS[key], PRL_TUNE_UNSET if value is None else int(value)))


def _parse_tuning(key: str, text: str) -> int:
    if key == "lmhead_tile":  # "128" | "256" | "256x256"
        return {"128": 128, "256": 256, "256x256": 512}.get(text.strip(), 0)
    return int(text)


def _sync_env_tuning(lib) -> None:
    """Map the PRL_* variables onto the library's tuning table - ONLY the keys whose variable changed since the last look
    (first look: only the variables that are set).  A key the environment does not mention is never written, so values
    placed with `set_tuning()` / `prl_set_tuning` survive (an A/B that calls `set_tuning` before its first launch used to
    be silently reset to the defaults); a variable that DISAPPEARS restores the library's own choice for its key."""
    global _env_seen
    now = tuple(os.environ.get(v) or None for v in _ENV_OF_KEY.values())
    if now == _env_seen:
        return
    before = _env_seen or (None,) * len(now)
    for (key, _), text, old in zip(_ENV_OF_KEY.items(), now, before):
        if text != old:
            lib.prl_set_tuning(TUNE_KEYS[key], PRL_TUNE_UNSET if text is None else _parse_tuning(key, text))
    _env_seen = now


def _with_env_tuning(lib, fn):
    def call(*args):
        _sync_env_tuning(lib)
        return fn(*args)

    call.__name__ = getattr(fn, "__name__", "prl_entry")
    call.raw = fn
    return call


def check(rc: int) -> None:
    if rc != PRL_OK:
        msg = load().prl_last_error()
        raise PrlError(rc, msg.decode("utf-8", "replace") if msg else "")


def ptr(t) -> int | None:
    """Device/host pointer of a torch tensor (None passes NULL)."""
    return None if t is None else t.data_ptr()


def current_stream_ptr(device=None) -> int:
    import torch

    return torch.cuda.current_stream(device).cuda_stream


def require_device(*tensors) -> None:
    """Fail loudly when a hot-path entry point is handed non-GPU tensors."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "pipelinerl_amd hot path needs tensors on a HIP device (got a CPU tensor); "
                "there is no CPU fallback"
            )
