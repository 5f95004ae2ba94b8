"""The compact `training_data` wire (batch_codec kind PRLCMP01) on the host: codec round trip, the publisher's gather recipe against
the generic encoder byte for byte (through the native publisher and a shared-memory log), the host-side facts the loader derives,
and the constructor's refusals.  The expansion on the learner's GPU is tests/test_gpu_compact_wire.py."""

import ctypes
import os

import numpy as np
import pytest
import torch

from pipelinerl_amd import _lib, batch_codec
from pipelinerl_amd.finetune.data import CompactBatch, compact_micro_batch
from pipelinerl_amd.preprocess import compact_sources, describe_compact
from pipelinerl_amd.ragged import RaggedRollouts


def _chunk(rng, n_seqs, with_ref, max_len=40):
    lens = rng.integers(2, max_len, size=n_seqs)
    out = np.array([int(rng.integers(0, l)) for l in lens])  # a sequence may have NO completion tokens
    seq_off = np.concatenate([[0], np.cumsum(lens)])
    lp_off = np.concatenate([[0], np.cumsum(out)])
    tokens = rng.integers(0, 1000, size=seq_off[-1]).astype(np.int32)
    labels = tokens.copy()
    for i in range(n_seqs):
        labels[seq_off[i]: seq_off[i + 1] - out[i]] = -100
    lp = rng.normal(size=lp_off[-1]).astype(np.float32)
    ref = rng.normal(size=lp_off[-1]).astype(np.float32) if with_ref else None
    z = np.zeros(n_seqs, dtype=np.int32)
    r = RaggedRollouts.from_numpy(tokens, labels, lp, ref, seq_off, lp_off, rng.normal(size=n_seqs), z, z, np.arange(n_seqs, dtype=np.int32),
                                  rng.integers(0, 5, size=n_seqs), np.ones(n_seqs, dtype=np.uint8), z.astype(np.uint8))
    k5 = rng.normal(size=(4, n_seqs)).astype(np.float32)
    return r, k5


def test_codec_round_trip_and_layout_agree():
    rng = np.random.default_rng(0)
    (r, k5) = _chunk(rng, 5, True)
    src = compact_sources(r, k5)
    cb = compact_micro_batch([r], [src["scalars"]], [(0, 3), (0, 0), (0, 4)], eos_token_id=7)
    rec = batch_codec.encode_compact(cb)
    head, base, where, total = batch_codec.compact_layout(cb.n_tokens, len(cb.logprobs), 3, True, cb.model_version, 0, 7)
    assert len(rec) == total and bytes(rec[:len(head)]) == head
    back = batch_codec.decode(rec)
    assert isinstance(back, CompactBatch) and back.model_version == cb.model_version == int(r.host_model_version[[3, 0, 4]].min())
    for k in ("tokens", "labels", "logprobs", "ref_logprobs", "seq_off", "lp_off", "seq_scalars"):
        np.testing.assert_array_equal(getattr(back, k), getattr(cb, k))
    # scalars: rewards rounded to fp32 + K5's rows in the order advantage, group_tokens, num_labels, overflow
    np.testing.assert_array_equal(cb.seq_scalars[0], r.reward.numpy().astype(np.float32)[[3, 0, 4]])
    np.testing.assert_array_equal(cb.seq_scalars[1:], k5[[2, 3, 0, 1]][:, [3, 0, 4]])
    # without reference log-probs the column is absent and the record shorter
    r2, k52 = _chunk(rng, 3, False)
    cb2 = compact_micro_batch([r2], [compact_sources(r2, k52)["scalars"]], [(0, 0), (0, 1)], eos_token_id=7)
    assert cb2.ref_logprobs is None and batch_codec.decode(batch_codec.encode_compact(cb2)).ref_logprobs is None


def test_host_facts_match_the_expanded_batch_rule():
    """labelled rows of the packed batch: row u predicts token u + 1; the first token of every sequence but the first carries no
    label (data.py:264-265) - the same indices `annotate_host_batch` finds on an expanded batch."""
    cb = CompactBatch(tokens=np.arange(9, dtype=np.int32), labels=np.array([-100, 1, 2, 3, 4, 5, -100, 7, 8], dtype=np.int32),
                      logprobs=np.zeros(6, np.float32), ref_logprobs=None, seq_off=np.array([0, 3, 6, 9]), lp_off=np.array([0, 2, 5, 6]),
                      seq_scalars=np.zeros((5, 3), np.float32))
    facts = cb.host_facts()
    # packed labels: [-100, 1, 2, -100(3: first of seq 1), 4, 5, -100, 7, 8] -> rows 0, 1, 3, 4, 6, 7
    assert facts["tokens"] == 9 and facts["labelled_rows"].tolist() == [0, 1, 3, 4, 6, 7]
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.finetune_loop import annotate_host_batch

    packed = torch.tensor([[-100, 1, 2, -100, 4, 5, -100, 7, 8]])
    f = torch.zeros(1, 9)
    b = annotate_host_batch(PipelineBatchEncoding(input_ids=torch.arange(9).unsqueeze(0), labels=packed, attention_mask=torch.ones(1, 9, dtype=torch.int64),
                                                  rewards=f, advantages=f, ref_logprobs=f, old_logprobs=f, group_tokens=f, num_labels=f, overflow=f, model_version=0))
    assert b.model_extra["labelled_rows"].tolist() == facts["labelled_rows"].tolist() and b.model_extra["tokens"] == 9


def test_publisher_recipe_is_the_generic_record_byte_for_byte():
    """Micro-batches spanning two chunks (one with, one without reference log-probs; sequences without completion tokens; a
    one-sequence micro-batch) through `describe_compact` -> native publisher -> shm log == `encode_compact(compact_micro_batch)`."""
    from pipelinerl_amd.ring import Log

    lib = _lib.load()
    rng = np.random.default_rng(1)
    chunks = [_chunk(rng, 6, True), _chunk(rng, 4, False), _chunk(rng, 3, False)]
    srcs = [compact_sources(r, k5) for r, k5 in chunks]
    micro_batches = [[(0, 5), (0, 0), (1, 2)], [(1, 0)], [(2, 0), (2, 1), (2, 2)], [(0, 1), (0, 2), (0, 3), (0, 4), (1, 1), (1, 3)]]
    name = f"prl_test_cmp_{os.getpid()}"
    log = Log(name, create=True, segment_bytes=1 << 20)
    pub = ctypes.c_void_p()
    _lib.check(lib.prl_publisher_create(0, ctypes.byref(pub)))
    try:
        inline, recs, pieces = bytearray(), [], []
        for mb in micro_batches:
            first = len(pieces)
            version = min(int(chunks[c][0].host_model_version[i]) for c, i in mb)
            nbytes = describe_compact([(srcs[c], i) for c, i in mb], version, 11, inline, pieces)
            recs.append((log._h.value, nbytes, first, len(pieces) - first))
        assert any(kind == _lib.PRL_PUB_FROM_HOST for *_, kind, _ in pieces)
        rec_arr = (_lib.PrlPubRecord * len(recs))(*recs)
        piece_arr = (_lib.PrlPubPiece * len(pieces))(*pieces)
        t = ctypes.c_uint64()
        _lib.check(lib.prl_publisher_submit(pub, None, 0, None, rec_arr, len(recs), piece_arr, len(pieces),
                                            (ctypes.c_char * len(inline)).from_buffer(inline), len(inline), ctypes.byref(t)))
        _lib.check(lib.prl_publisher_wait(pub, t.value, 5000))
        reader = Log(name, reader=True)
        for mb in micro_batches:
            want = batch_codec.encode_compact(compact_micro_batch([c[0] for c in chunks], [s["scalars"] for s in srcs], mb, eos_token_id=11))
            got = reader.read(timeout=1)
            assert bytes(got) == bytes(want)
            cb = batch_codec.decode(got)
            assert cb.n_seqs == len(mb) and cb.eos_token_id == 11 and cb.padding == 0
        reader.close()
        # a null host address is refused at submit
        bad = (_lib.PrlPubPiece * 1)((0, 0, 8, _lib.PRL_PUB_FROM_HOST, 0))
        rec = (_lib.PrlPubRecord * 1)((log._h.value, 8, 0, 1))
        assert lib.prl_publisher_submit(pub, None, 0, None, rec, 1, bad, 1, None, 0, ctypes.byref(t)) == _lib.PRL_EINVAL
    finally:
        lib.prl_publisher_destroy(pub)
        log.close()
        Log.unlink_name(name)


def test_sequence_parallel_and_reference_column_records():
    """The two extensions of the record: `slice` / `slices` + `padding` (every rank of an SP group gets the record and keeps its slice)
    and `ref_column` (the expanded batch's ref_logprobs when a reference policy ran in the preprocessor).  Round trip, header text =
    the generic encoder's, and records WITHOUT them are byte-identical to what they were (no new scalar, no new tensor)."""
    rng = np.random.default_rng(4)
    r, k5 = _chunk(rng, 4, False)
    src = compact_sources(r, k5)
    members = [(0, 2), (0, 0), (0, 3)]
    plain = compact_micro_batch([r], [src["scalars"]], members, eos_token_id=5)
    n = plain.n_tokens
    pad = (-n) % 4
    ref = rng.normal(size=n + pad).astype(np.float32)
    for k in range(4):
        cb = compact_micro_batch([r], [src["scalars"]], members, eos_token_id=5, padding=pad)
        cb.slice_index, cb.num_slices, cb.ref_column = k, 4, ref
        rec = batch_codec.encode_compact(cb)
        head, base, where, total = batch_codec.compact_layout(n, len(cb.logprobs), 3, False, cb.model_version, pad, 5, k, 4, n + pad)
        assert len(rec) == total and bytes(rec[:len(head)]) == head and where["ref_column"][1] == 4 * (n + pad)
        back = batch_codec.decode(rec)
        assert (back.slice_index, back.num_slices, back.padding) == (k, 4, pad)
        np.testing.assert_array_equal(back.ref_column, ref)
        np.testing.assert_array_equal(back.tokens, plain.tokens)
        facts = back.host_facts()
        assert facts["tokens"] == (n + pad) // 4
        # the slice's labelled rows are those of the packed labels cut to the slice (the slice's last row has no successor inside it)
        lab = plain.labels.copy()
        lab[plain.seq_off[1:-1]] = -100
        lab = np.concatenate([lab, np.full(pad, -100, dtype=lab.dtype)])[k * (n + pad) // 4: (k + 1) * (n + pad) // 4]
        assert facts["labelled_rows"].tolist() == np.flatnonzero(lab[1:] != -100).tolist()
    # the recipe of the preprocessor's publisher writes the same header and puts the column where the layout says
    inline, pieces = bytearray(), []
    nbytes = describe_compact([(src, i) for _, i in members], plain.model_version, 5, inline, pieces, padding=pad, slice_index=1, num_slices=4, ref_block=(128, 4 * (n + pad)))
    head, base, where, total = batch_codec.compact_layout(n, len(plain.logprobs), 3, False, plain.model_version, pad, 5, 1, 4, n + pad)
    assert nbytes == total and bytes(inline[:len(head)]) == head
    blocks = [p for p in pieces if p[3] == _lib.PRL_PUB_FROM_BLOCK]
    assert blocks == [(128, base + where["ref_column"][0], 4 * (n + pad), _lib.PRL_PUB_FROM_BLOCK, 0)]
    with pytest.raises(ValueError, match="ref column"):
        describe_compact([(src, i) for _, i in members], 0, 5, bytearray(), [], padding=pad, ref_block=(0, 4 * n + 4 * pad + 4))
    # unchanged records for the common case
    old_head = batch_codec.compact_layout(n, len(plain.logprobs), 3, False, plain.model_version, 0, 5)[0]
    assert b"slice" not in old_head and b"ref_column" not in old_head and bytes(batch_codec.encode_compact(plain)[:len(old_head)]) == old_head


def test_compact_wire_refusals():
    """The compact wire is a choice with preconditions, not a fallback: wrong combinations fail at construction; expansion
    without a HIP device fails loudly."""
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.preprocess import PreprocessorConfig, PreprocessorLoop

    def cfg(**kw):
        base = dict(exp_path="/tmp/x", num_trainers=1, train_batch_size=4, gradient_accumulation_passes=1, seq_length=64, attempts=2, rl=RLConfig(), eos_token_id=0)
        base.update(kw)
        return PreprocessorConfig(**base)

    with pytest.raises(ValueError, match="wire must be"):
        PreprocessorLoop(cfg(), "cpu", wire="tiny")
    with pytest.raises(ValueError, match="seq_packing"):
        PreprocessorLoop(cfg(seq_packing=False), "cpu", wire="compact")
    # sequence parallelism, a reference policy and the OOV patch are served by the compact wire (tests/test_gpu_compact_wire.py); it still needs a device
    for kw in (dict(), dict(seq_parallel=2, num_trainers=2)):
        with pytest.raises(RuntimeError, match="HIP device"):
            PreprocessorLoop(cfg(**kw), "cpu", wire="compact", ref_model=object() if kw else None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        PreprocessorLoop(cfg(), "cpu", wire="compact")
    cb = CompactBatch(tokens=np.zeros(2, np.int32), labels=np.zeros(2, np.int32), logprobs=np.zeros(1, np.float32), ref_logprobs=None,
                      seq_off=np.array([0, 2]), lp_off=np.array([0, 1]), seq_scalars=np.zeros((5, 1), np.float32))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        cb.to_batch("cpu")
