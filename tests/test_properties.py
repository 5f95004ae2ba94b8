"""Property tests (hypothesis) of the host-side planning code: the O(#sequences) plans that drive
the K5 / K6 launches, the micro-batch scheduler and the weight-sync bucket plan.  CPU only."""

import numpy as np
import torch
from hypothesis import given, settings
from hypothesis import strategies as st


@settings(max_examples=60, deadline=None)
@given(st.lists(st.integers(0, 30), min_size=0, max_size=40), st.data())
def test_plan_packing_matches_bruteforce(lens, data):
    from pipelinerl_amd.finetune.data import plan_packing

    lens = np.asarray(lens, dtype=np.int64)
    n = len(lens)
    order = data.draw(st.permutations(list(range(n)))) if n else []
    mbs, k = [], 0
    while k < n:
        m = data.draw(st.integers(0, 5))
        mbs.append(order[k:k + m])
        k += m
    pads = data.draw(st.one_of(st.none(), st.lists(st.integers(0, 7), min_size=len(mbs), max_size=len(mbs))))
    src, seg, dst, off = plan_packing(lens, mbs, pads)
    # brute force
    e_src, e_seg, e_len, e_off = [], [], [], [0]
    for j, mb in enumerate(mbs):
        for q, s in enumerate(mb):
            e_src.append(s); e_seg.append(q); e_len.append(int(lens[s]))
        cnt = len(mb)
        if pads is not None and pads[j] > 0:
            e_src.append(-1); e_seg.append(cnt); e_len.append(pads[j]); cnt += 1
        e_off.append(e_off[-1] + cnt)
    assert src.tolist() == e_src and seg.tolist() == e_seg and off.tolist() == e_off
    assert dst.tolist() == [0] + np.cumsum(e_len).tolist()


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.integers(0, 5), st.integers(0, 2), st.integers(0, 3)), min_size=1, max_size=40))
def test_plan_groups_matches_bruteforce(rows):
    from pipelinerl_amd.finetune.rl import plan_groups

    g = np.array([r[0] for r in rows], dtype=np.int32)
    s = np.array([r[1] for r in rows], dtype=np.int32)
    r = np.array([r[2] for r in rows], dtype=np.int32)
    key_off, key_members, group_off, group_members, n_roll = plan_groups(g, s, r)
    keys = sorted(set(zip(g.tolist(), s.tolist())))
    assert len(key_off) - 1 == len(keys)
    for k, (gg, ss) in enumerate(keys):
        members = key_members[key_off[k]:key_off[k + 1]].tolist()
        assert members == [i for i in range(len(rows)) if g[i] == gg and s[i] == ss]  # dataset order
    groups = sorted(set(g.tolist()))
    for k, gg in enumerate(groups):
        members = group_members[group_off[k]:group_off[k + 1]].tolist()
        assert members == [i for i in range(len(rows)) if g[i] == gg]
        assert n_roll[k] == len({int(r[i]) for i in members})


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 4), st.integers(1, 3), st.integers(1, 4), st.integers(8, 64), st.lists(st.integers(1, 8), min_size=1, max_size=80))
def test_scheduler_invariants(num_lead, tbs, passes_per_lead, seq_length, lens):
    """For any arrival sequence: samples leave in arrival order exactly once, every micro-batch fits
    the token budget, trainers are served round-robin, quotas are never exceeded, and at every step
    boundary all lead trainers hold the same number of samples."""
    from pipelinerl_amd.preprocess import MicroBatchScheduler

    lens = [min(x, seq_length) for x in lens]
    sched = MicroBatchScheduler(num_trainers=num_lead, train_batch_size=tbs, gradient_accumulation_passes=passes_per_lead * num_lead,
                                seq_length=seq_length, length_of=lambda e: e[1])
    sched.push(list(enumerate(lens)))
    emitted = []
    while sched.queue:
        mbs, done = sched.drain()
        emitted += mbs
        if done:
            counts = set(sched.samples_per_trainer.values())
            assert len(counts) == 1
        if not mbs and not done:
            break
    flat = [s[0] for mb in emitted for s in mb.samples]
    assert flat == list(range(len(flat)))
    assert [mb.trainer_id for mb in emitted] == [i % num_lead for i in range(len(emitted))]
    for mb in emitted:
        assert mb.sentinel == (len(mb.samples) == 0)
        assert sum(s[1] for s in mb.samples) <= seq_length
    assert all(v <= sched.target_samples_per_lead for v in sched.samples_per_trainer.values())
    assert sched.published_samples == len(flat)


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(st.integers(1, 5000), st.sampled_from([torch.bfloat16, torch.float32, torch.float16])), min_size=1, max_size=30),
       st.integers(256, 20000))
def test_bucket_plan_invariants(params, bucket_bytes):
    from pipelinerl_amd.weight_sync import ParamSpec, bucket_nbytes, plan_buckets

    specs = [ParamSpec(f"p{i}", (n,), dt) for i, (n, dt) in enumerate(params)]
    plan = plan_buckets(specs, bucket_bytes)
    assert [sp.name for b in plan for sp, _ in b] == [sp.name for sp in specs]  # order preserved, nothing lost
    for b in plan:
        end = 0
        for sp, off in b:
            assert off % 256 == 0 and off >= end
            end = off + sp.nbytes
        assert bucket_nbytes(b) >= end
        assert len(b) == 1 or bucket_nbytes(b) <= bucket_bytes  # only a single oversized tensor may exceed the budget


@settings(max_examples=50, deadline=None)
@given(st.lists(st.tuples(st.integers(1, 12), st.integers(0, 20), st.integers(0, 3), st.booleans(), st.sampled_from([None, "stop", "length", " Length ", "weird"])),
                min_size=1, max_size=25), st.booleans(), st.integers(0, 2**31 - 1))
def test_rollouts_round_trip_through_both_wire_formats(seqs, with_ref, seed):
    """RaggedRollouts -> binary record (shm backend) -> RaggedRollouts, and -> actor-stream dicts (JSONL
    mirror / files backend) -> RaggedRollouts: every column identical, for ragged shapes incl. empty completions."""
    import json

    from pipelinerl_amd import batch_codec
    from pipelinerl_amd.ragged import RaggedRollouts

    rng = np.random.default_rng(seed)
    entries = []
    for i, (p, c, g, fin, reason) in enumerate(seqs):
        ids = rng.integers(2, 1000, size=p + c).tolist()
        e = {"input_ids": ids, "labels": [-100] * p + ids[p:], "logprobs": (-rng.random(c)).astype(np.float32).tolist(),
             "ref_logprobs": (-rng.random(c)).astype(np.float32).tolist() if with_ref else [],
             "reward": float(np.float32(rng.normal())), "group_id": f"g{g}", "finished": fin,
             "metadata": {"model_version": int(rng.integers(0, 1000)), "rollout_index": i % 4, "step_index": int(rng.integers(0, 3))}}
        if reason is not None:
            e["finish_reason"] = reason
        entries.append(e)
    rag = RaggedRollouts.from_entries(entries)
    cols = ("tokens", "labels", "logprobs", "seq_off", "lp_off", "reward", "group_index", "step_index", "rollout_index", "model_version",
            "finished", "finish_code")

    def same(a, b):
        for name in cols:
            assert torch.equal(getattr(a, name), getattr(b, name)), name
        assert (a.ref_logprobs is None) == (b.ref_logprobs is None)
        if a.ref_logprobs is not None:
            assert torch.equal(a.ref_logprobs, b.ref_logprobs)
        assert [a.group_ids[i] for i in a.host_group_index] == [b.group_ids[i] for i in b.host_group_index]

    same(batch_codec.decode(batch_codec.encode_rollouts(rag)), rag)
    # the JSONL form survives a text round trip as well (fp32 values print exactly as python floats)
    same(RaggedRollouts.from_entries(json.loads(json.dumps(rag.to_entries()))), rag)


@settings(max_examples=40, deadline=None)
@given(st.lists(st.tuples(st.integers(1, 4), st.integers(1, 70), st.booleans()), min_size=1, max_size=8), st.sampled_from([256, 4096, 1 << 16]),
       st.integers(0, 2**31 - 1))
def test_batches_gathered_into_the_log_come_back_identical(shapes, segment_bytes, seed):
    """`batch_codec.append_batch` gathers a batch's tensors straight into the shared-memory segment (`prl_log_appendv`, no
    intermediate record) - for any mix of shapes, segment sizes small enough to force a rollover inside the sequence and records
    larger than a segment, every batch a reader gets back equals the one written, and equals the one-copy `encode_batch` form."""
    import time

    from pipelinerl_amd import batch_codec
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.ring import Log

    rng = np.random.default_rng(seed)
    batches = []
    for rows, cols, packed in shapes:
        rows = 1 if packed else rows
        f = lambda: torch.from_numpy(rng.standard_normal((rows, cols)).astype(np.float32))  # noqa: E731
        i = lambda hi: torch.from_numpy(rng.integers(-100, hi, size=(rows, cols), dtype=np.int64))  # noqa: E731
        kw = dict(input_ids=i(1000), attention_mask=torch.ones(rows, cols, dtype=torch.int64), labels=i(1000), rewards=f(), advantages=f(),
                  ref_logprobs=f(), old_logprobs=f(), group_tokens=f(), num_labels=f(), overflow=f(), model_version=int(rng.integers(0, 99)),
                  is_packed=packed)
        if packed:
            kw.update(position_ids=i(cols), segment_ids=i(4), seq_boundaries=torch.tensor([0, cols], dtype=torch.int32))
        batches.append(PipelineBatchEncoding(**kw))
    name = f"prl_prop_{time.time_ns()}"
    w = Log(name, create=True, segment_bytes=segment_bytes)
    try:
        r = Log(name, reader=True)
        for b in batches:
            batch_codec.append_batch(w, b)
        for b in batches:
            rec = r.read(block=False)
            assert bytes(rec) == bytes(batch_codec.encode_batch(b))  # the gathered record IS the one-copy record
            got = PipelineBatchEncoding(**batch_codec.decode(rec))  # a batch record decodes to the constructor's kwargs
            for k in ("input_ids", "attention_mask", "labels", "position_ids", "segment_ids", "seq_boundaries", "rewards", "advantages",
                      "ref_logprobs", "old_logprobs", "group_tokens", "num_labels", "overflow"):
                x, y = getattr(got, k, None), getattr(b, k, None)
                assert (x is None) == (y is None), k
                assert x is None or (x.dtype == y.dtype and torch.equal(x, y)), k
            assert got.model_version == b.model_version and bool(got.is_packed) == bool(b.is_packed)
        r.close()
    finally:
        w.close()
        Log.unlink_name(name)


@settings(max_examples=60, deadline=None)
@given(st.lists(st.one_of(st.none(), st.tuples(st.sampled_from(["int32", "int64", "float32", "float64", "uint8"]),
                                               st.lists(st.integers(0, 9), min_size=1, max_size=3))), min_size=1, max_size=12),
       st.integers(1, 4), st.integers(0, 2**31 - 1))
def test_stager_round_trips_any_mix_of_arrays(specs, slots, seed):
    """`PinnedStager.upload` (many host arrays -> one buffer -> views) and `.download`: any mix of dtypes, empty arrays and None
    entries, ring slots reused - every view has the array's dtype, shape and contents, starts on a 256-byte boundary of the one
    allocation, and earlier results survive later uploads (CPU device: the same layout code the GPU path runs)."""
    from pipelinerl_amd.staging import PinnedStager

    rng = np.random.default_rng(seed)
    st_ = PinnedStager("cpu", slots=slots, min_bytes=64)
    rounds = []
    for _ in range(slots + 2):  # more rounds than slots: every slot is reused at least once
        arrays = [None if sp is None else (rng.integers(0, 200, size=sp[1]).astype(sp[0])) for sp in specs]
        out = st_.upload(arrays)
        rounds.append((arrays, out))
    for arrays, out in rounds:
        bases = set()
        for a, t in zip(arrays, out):
            assert (a is None) == (t is None)
            if a is None:
                continue
            assert tuple(t.shape) == a.shape and np.array_equal(t.numpy(), a) and t.numpy().dtype == a.dtype
            if a.size:
                assert t.storage_offset() * t.element_size() % 256 == 0
                bases.add(t.untyped_storage().data_ptr())
        assert len(bases) <= 1
    block = torch.from_numpy(rng.standard_normal(37).astype(np.float32))
    assert torch.equal(st_.download(block), block) and st_.downloads == 1 and st_.uploads == sum(1 for arrays, _ in rounds if any(a is not None and a.size for a in arrays))
