"""The compact `training_data` wire end to end on a GPU: a micro-batch expanded on the consumer's device is the batch the
preprocessor's own pack launch writes - every column bit for bit - and a learner's loader fed by `PreprocessorLoop(wire="compact")`
sees the same batch sequence as one fed by the full wire (sentinels, model versions, boundaries included), with 4-5 x fewer
bytes in the log."""

import queue
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _tensors(batch):
    return {k: v.detach().cpu() for k, v in batch.tensors()}


@pytest.mark.parametrize("with_ref", [True, False])
def test_expansion_on_the_consumer_equals_the_preprocessors_pack(libprl, cuda_device, with_ref):
    from pipelinerl_amd.finetune.data import compact_micro_batch, pack_prepared
    from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data_ragged
    from pipelinerl_amd.preprocess import compact_sources
    from pipelinerl_amd.ragged import RaggedRollouts
    from pipelinerl_amd.staging import PinnedStager
    from pipelinerl_amd.synthetic import make_entries

    attempts = 4
    entries = make_entries(5, attempts=attempts, seq_length=200, vocab=500, seed=3, prompt_min=3, prompt_max=40)
    if with_ref:
        rng = np.random.default_rng(0)
        for e in entries:
            e["ref_logprobs"] = [float(x) for x in rng.normal(size=len(e["logprobs"])).astype(np.float32)]
    else:
        for e in entries:
            e.pop("ref_logprobs", None)
    host = RaggedRollouts.from_entries(entries)
    assert (host.ref_logprobs is not None) == with_ref
    prep = populate_rl_data_ragged(host.to(cuda_device), 2, RLConfig())
    plan = [[7, 0, 3], [1], [19, 18, 2, 4, 5, 6], [8, 9, 10, 11, 12, 13, 14, 15, 16, 17]]
    want = pack_prepared(prep, plan, 2)
    src = compact_sources(host, prep.k5_out32.cpu().numpy())
    stager = PinnedStager(cuda_device, slots=2)
    for j, members in enumerate(plan):
        cb = compact_micro_batch([host], [src["scalars"]], [(0, i) for i in members], eos_token_id=2)
        assert cb.n_tokens * 68 > 4 * (cb.tokens.nbytes + cb.labels.nbytes + cb.logprobs.nbytes)  # what the wire saves
        for st in (None, stager):
            got = cb.to_batch(cuda_device, st)
            a, b = _tensors(want[j]), _tensors(got)
            assert a.keys() == b.keys()
            for k in a:
                assert a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]), k
            assert got.model_version == want[j].model_version and got.is_packed and not got.sentinel and got.padding == 0
        # the loader's host-side facts are those of the expanded batch
        from pipelinerl_amd.finetune_loop import annotate_host_batch
        from pipelinerl_amd.finetune.types import PipelineBatchEncoding

        on_host = annotate_host_batch(PipelineBatchEncoding(**{k: v for k, v in a.items()}, model_version=0))
        facts = cb.host_facts()
        assert facts["tokens"] == on_host.model_extra["tokens"] and torch.equal(facts["labelled_rows"], on_host.model_extra["labelled_rows"])


@pytest.mark.parametrize("trainers", [1, 2])
def test_compact_wire_delivers_the_full_wires_batches(libprl, cuda_device, tmp_path, trainers):
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.synthetic import make_entries

    streams.reset_streams_backend()
    streams.set_streams_backend("shm", segment_bytes=1 << 20, owner=True, trim_topics=())
    try:
        attempts, n_groups = 4, 10
        raw = make_entries(n_groups, attempts=attempts, seq_length=96, vocab=300, seed=5, prompt_min=3, prompt_max=20)
        for i, e in enumerate(raw):
            e.setdefault("metadata", {})["model_version"] = i // 8  # versions move inside the stream
        cfg_kw = dict(num_trainers=trainers, train_batch_size=2, gradient_accumulation_passes=4, seq_length=256, attempts=attempts,
                      rl=RLConfig(), eos_token_id=2, chunk_n_groups=2)
        from pipelinerl_amd.ring import Log
        from pipelinerl_amd.streams import ring_name

        def counts(exp):
            return [Log(ring_name(streams.SingleStreamSpec(exp_path=exp, topic="training_data", partition=p)), reader=True).stats()["records"] for p in range(trainers)]

        # publish everything first, then read exactly as many records as each partition's log holds
        full_pub, full, full_bytes, _ = _run_loop_then_read(streams, tmp_path, cuda_device, "full", n_groups, attempts, raw, cfg_kw, counts)
        cmp_pub, cmp, cmp_bytes, loop = _run_loop_then_read(streams, tmp_path, cuda_device, "compact", n_groups, attempts, raw, cfg_kw, counts)
        assert full_pub == cmp_pub == n_groups * attempts
        n_real = 0
        for part in range(trainers):
            assert len(full[part]) == len(cmp[part]) > 0
            for a, b in zip(full[part], cmp[part]):
                assert a.sentinel == b.sentinel and a.model_version == b.model_version and a.is_packed == b.is_packed and a.padding == b.padding
                ta, tb = _tensors(a), _tensors(b)
                assert ta.keys() == tb.keys()
                for k in ta:
                    assert ta[k].dtype == tb[k].dtype and torch.equal(ta[k], tb[k]), (part, k)
                assert b.input_ids.is_cuda
                if not a.sentinel:
                    n_real += 1
                    assert a.model_extra["tokens"] == b.model_extra["tokens"]
                    assert torch.equal(a.model_extra["labelled_rows"].cpu(), b.model_extra["labelled_rows"].cpu())
        assert n_real >= 4
        assert cmp_bytes * 3 < full_bytes, (cmp_bytes, full_bytes)  # 12-16 bytes per token + headers against 68
        assert "k6_plan_launch" not in loop.prof and loop.prof.get("publish_submit", 0) > 0  # the preprocessor never packed
    finally:
        streams.reset_streams_backend()


def _run_loop_then_read(streams, tmp_path, cuda_device, wire, n_groups, attempts, raw, cfg_kw, counts):
    from pipelinerl_amd.preprocess import PreprocessorConfig, PreprocessorLoop
    from pipelinerl_amd.ragged import RaggedRollouts

    exp = tmp_path / wire
    exp.mkdir()
    with streams.write_to_streams(streams.SingleStreamSpec(exp_path=exp, topic="actor")) as w:
        for g in range(n_groups):
            w.write(RaggedRollouts.from_entries(raw[g * attempts:(g + 1) * attempts]))
    loop = PreprocessorLoop(PreprocessorConfig(exp_path=exp, **cfg_kw), cuda_device, wire=wire, profile=True)
    published = loop.run(max_published_samples=n_groups * attempts, idle_timeout=2.0)
    n = counts(exp)
    from pipelinerl_amd.finetune_loop import run_data_loader
    from pipelinerl_amd.ring import Log
    from pipelinerl_amd.streams import ring_name

    out, nbytes = [], 0
    for part, want in enumerate(n):
        spec = streams.SingleStreamSpec(exp_path=exp, topic="training_data", partition=part)
        nbytes += Log(ring_name(spec), reader=True).stats()["bytes"]
        q: queue.Queue = queue.Queue()
        stop = threading.Event()
        threading.Thread(target=run_data_loader, args=(spec, q, cuda_device, stop, True), daemon=True).start()
        got = []
        while len(got) < want:
            item = q.get(timeout=20)
            if isinstance(item, Exception):
                raise item
            got.append(item)
        stop.set()
        out.append(got)
    return published, out, nbytes, loop
