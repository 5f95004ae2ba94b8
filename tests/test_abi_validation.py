"""Argument validation of the device entry points happens before any HIP call, so the error
behaviour of the C ABI can be checked on a machine without a GPU: every bad call returns
PRL_EINVAL / PRL_ENOMEM and leaves a message in prl_last_error()."""

import ctypes

import pytest

from pipelinerl_amd import _lib
from pipelinerl_amd._lib import PrlLossConfig

P = 0x1000  # any non-null "pointer": validation must reject the call before dereferencing it


def _cfg(**kw):
    c = PrlLossConfig(policy_loss=0, use_advantages=1, token_weight=1.0, clip_lo=0.8, clip_hi=1.2, clamp_log_ratio_ref_new=5.0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _err(lib):
    return lib.prl_last_error().decode()


def test_logprob_entry_points_reject_bad_geometry(libprl):
    lib = libprl
    assert lib.prl_logprob_entropy_fwd(0, 8, 16, P, 0, 16, P, 1.0, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_logprob_entropy_fwd(1, 8, 0, P, 0, 16, P, 1.0, P, P, P, None) == _lib.PRL_EINVAL and "vocab" in _err(lib)
    assert lib.prl_logprob_entropy_fwd(1, 8, 16, P, 7, 16, P, 1.0, P, P, P, None) == _lib.PRL_EINVAL and "dtype" in _err(lib)
    assert lib.prl_logprob_entropy_fwd(1, 8, 16, P, 0, 8, P, 1.0, P, P, P, None) == _lib.PRL_EINVAL and "stride" in _err(lib)
    assert lib.prl_logprob_entropy_fwd(1, 8, 16, None, 0, 16, P, 1.0, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_logprob_entropy_fwd(1, 8, 16, P, 0, 16, P, 0.0, P, P, P, None) == _lib.PRL_EINVAL and "temperature" in _err(lib)
    assert lib.prl_logprob_entropy_fwd(1, 8, 16, P, 0, 16, None, 1.0, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_logprob_entropy_bwd(1, 8, 16, P, 0, 16, P, 1.0, P, P, None, None, None, P, None) == _lib.PRL_EINVAL
    assert lib.prl_logprob_entropy_bwd(1 << 20, 1 << 20, 16, P, 0, 16, P, 1.0, P, P, P, None, None, P, None) == _lib.PRL_EINVAL


def test_loss_entry_point_rejects_bad_arguments(libprl):
    lib = libprl
    ok = _cfg()
    args = lambda cfg, rows=1, cols=8, ws=P, ws_bytes=1 << 30, labels=P, pos=P, ext=(None, None): (  # noqa: E731
        ctypes.byref(cfg), rows, cols, labels, pos, P, P, P, P, P, P, P, P, P, ext[0], ext[1], None, None, None, P, ws, ws_bytes, None)
    assert lib.prl_grpo_loss_fwd_bwd(*args(ok, rows=0)) == _lib.PRL_EINVAL
    assert lib.prl_grpo_loss_fwd_bwd(*args(_cfg(policy_loss=9))) == _lib.PRL_EINVAL and "policy_loss" in _err(lib)
    assert lib.prl_grpo_loss_fwd_bwd(*args(_cfg(policy_loss=2))) == _lib.PRL_EINVAL and "GSPO" in _err(lib)
    assert lib.prl_grpo_loss_fwd_bwd(*args(ok, labels=None)) == _lib.PRL_EINVAL
    assert lib.prl_grpo_loss_fwd_bwd(*args(ok, ws=None)) == _lib.PRL_EINVAL and "workspace" in _err(lib)
    assert lib.prl_grpo_loss_fwd_bwd(*args(ok, ws_bytes=16)) == _lib.PRL_ENOMEM and "too small" in _err(lib)
    assert lib.prl_grpo_loss_fwd_bwd(*args(_cfg(flat_micro_batches=1), pos=None)) == _lib.PRL_EINVAL
    assert lib.prl_grpo_loss_fwd_bwd(*args(_cfg(flat_micro_batches=1), rows=2)) == _lib.PRL_EINVAL
    assert lib.prl_grpo_loss_fwd_bwd(None, 1, 8, *([P] * 11), None, None, None, None, None, P, P, 1 << 30, None) == _lib.PRL_EINVAL


def test_fused_and_pack_entry_points_reject_bad_arguments(libprl):
    lib = libprl
    assert lib.prl_fused_logits_loss(ctypes.byref(_cfg(policy_loss=2)), 1, 8, 16, P, 0, 16, 1.0, *([P] * 8), P, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_fused_logits_loss(None, 1, 8, 16, P, 0, 16, 1.0, *([P] * 8), P, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_fused_logits_loss(ctypes.byref(_cfg()), 1, 8, 16, P, 0, 16, 1.0, *([P] * 8), P, P, P, None, None) == _lib.PRL_EINVAL
    assert lib.prl_pack_collate(-1, 8, *([P] * 14), 0, 2, *([P] * 12), None) == _lib.PRL_EINVAL
    assert lib.prl_pack_collate(1, 8, None, *([P] * 13), 0, 2, *([P] * 12), None) == _lib.PRL_EINVAL
    assert lib.prl_pack_collate(1, 8, *([P] * 14), 0, 2, None, *([P] * 11), None) == _lib.PRL_EINVAL and "output" in _err(lib)
    assert lib.prl_pack_collate(0, 0, *([None] * 14), 0, 2, *([None] * 12), None) == _lib.PRL_OK  # empty plan: nothing to do
    assert lib.prl_pad_collate(1, 16, 0, None, *([P] * 11), 0, *([P] * 10), None) == _lib.PRL_EINVAL
    assert lib.prl_seq_scan(-1, P, P, P, None, None, 2, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_seq_scan(0, None, None, None, None, None, 2, None, None, None) == _lib.PRL_OK
    assert lib.prl_seq_scan(3, None, P, P, None, None, 2, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_group_advantages(2, 1, 1, P, P, P, P, P, None, P, 1, P, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_segment_sums(-1, 1, P, P, P, P, P, P, P, None) == _lib.PRL_EINVAL
    # the three GSPO entry points: null config, negative shapes, null pointers - refused before any launch
    assert lib.prl_gspo_segment_sums(None, 8, 1, P, P, P, P, P, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_gspo_segment_sums(ctypes.byref(_cfg()), -1, 1, P, P, P, P, P, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_gspo_segment_sums(ctypes.byref(_cfg()), 8, 1, P, P, P, None, P, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_gspo_segment_terms(None, 1, P, 1.0, 0, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_gspo_segment_terms(ctypes.byref(_cfg()), 1, P, 1.0, 0, P, P, None, None) == _lib.PRL_EINVAL
    assert lib.prl_gspo_expand(8, 0, P, P, P, P, P, None) == _lib.PRL_EINVAL
    assert lib.prl_gspo_expand(8, 1, P, None, P, P, P, None) == _lib.PRL_EINVAL


def test_ring_and_wsync_reject_bad_arguments(libprl):
    lib = libprl
    h = ctypes.c_void_p()
    assert lib.prl_ring_create(b"prl_test_bad", 0, 64, ctypes.byref(h)) == _lib.PRL_EINVAL
    assert lib.prl_ring_create(b"prl_test_bad", 4, 0, ctypes.byref(h)) == _lib.PRL_EINVAL
    assert lib.prl_ring_attach(b"prl_does_not_exist_xyz", ctypes.byref(h)) == _lib.PRL_EFAULT
    assert lib.prl_ring_put(None, None, 0, 0) == _lib.PRL_EINVAL
    assert lib.prl_wsync_unique_id(None) == _lib.PRL_EINVAL
    uid = (ctypes.c_uint8 * _lib.PRL_WSYNC_UID_BYTES)()
    assert lib.prl_wsync_init(uid, 3, 2, 0, ctypes.byref(h)) == _lib.PRL_EINVAL and "rank" in _err(lib)
    assert lib.prl_wsync_destroy(None) == _lib.PRL_OK


def test_bucket_copy_rejects_bad_segments(libprl):
    lib = libprl
    segs = (_lib.PrlSegment * 2)()
    segs[0].tensor, segs[0].bucket_offset, segs[0].nbytes = P, 0, 64
    segs[1].tensor, segs[1].bucket_offset, segs[1].nbytes = P, 64, 65
    for fn in (lib.prl_bucket_gather, lib.prl_bucket_scatter):
        assert fn(P, 128, segs, 2, None) == _lib.PRL_EINVAL and "exceeds" in _err(lib)
        assert fn(P, 128, segs, -1, None) == _lib.PRL_EINVAL
        assert fn(None, 128, segs, 2, None) == _lib.PRL_EINVAL
        assert fn(P, 128, None, 2, None) == _lib.PRL_EINVAL
        assert fn(P, 128, segs, 0, None) == _lib.PRL_OK  # nothing to do, no launch
    segs[1].nbytes = 64
    segs[1].tensor = None
    assert lib.prl_bucket_gather(P, 128, segs, 2, None) == _lib.PRL_EINVAL and "null tensor" in _err(lib)
    segs[1].tensor, segs[1].bucket_offset = P, -8
    assert lib.prl_bucket_scatter(P, 128, segs, 2, None) == _lib.PRL_EINVAL and "negative" in _err(lib)
