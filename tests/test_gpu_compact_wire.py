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


class _RefPolicy(torch.nn.Module):
    """A frozen reference policy that only exposes `.logits` (the K1 path of `annotate_ref_logprobs`)."""

    def __init__(self, vocab: int, dim: int = 32):
        super().__init__()
        torch.manual_seed(11)
        self.emb = torch.nn.Embedding(vocab, dim)
        self.out = torch.nn.Linear(dim, vocab)

    def forward(self, input_ids=None, **kw):
        import types

        return types.SimpleNamespace(logits=self.out(torch.tanh(self.emb(input_ids))).float())


@pytest.mark.parametrize("trainers,sp,kl,oov", [(1, 1, False, False), (2, 1, False, False), (2, 2, False, False), (1, 1, True, False), (2, 2, True, False),
                                                (4, 2, True, False), (2, 1, False, True), (2, 2, True, True)],
                         ids=["1_trainer", "2_trainers", "seq_parallel_2", "kl", "seq_parallel_2+kl", "2_leads_x_seq_parallel_2+kl", "oov_patch", "seq_parallel_2+kl+oov_patch"])
def test_compact_wire_delivers_the_full_wires_batches(libprl, cuda_device, tmp_path, trainers, sp, kl, oov):
    """Every combination the full wire serves: data-parallel partitions, sequence-parallel slices (types.py:145-180: every rank of the
    group expands the record and keeps its slice, filler included) and a reference policy in the preprocessor (preprocess.py:86-104's
    role: the `ref_logprobs` column comes from its forward, 4 bytes per token cross the bus and ride in the record)."""
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
                      rl=RLConfig(kl_coef=0.05 if kl else 0.0), eos_token_id=2, chunk_n_groups=2, seq_parallel=sp)
        ref_model = _RefPolicy(300).to(cuda_device).eval() if kl else None
        if oov:  # ids 280..299 are not in the tokenizer's vocabulary: replaced by "the" (id 5) on the device (preprocess.py:107-141), on both wires
            from pipelinerl_amd.preprocess import OovPatcher

            make_patcher = lambda: OovPatcher(range(280), 5, cuda_device)  # noqa: E731
        else:
            make_patcher = lambda: None  # noqa: E731
        from pipelinerl_amd.ring import Log
        from pipelinerl_amd.streams import ring_name

        def counts(exp):
            return [Log(ring_name(streams.SingleStreamSpec(exp_path=exp, topic="training_data", partition=p)), reader=True).stats()["records"] for p in range(trainers)]

        # publish everything first, then read exactly as many records as each partition's log holds
        full_pub, full, full_bytes, _ = _run_loop_then_read(streams, tmp_path, cuda_device, "full", n_groups, attempts, raw, cfg_kw, counts, ref_model, make_patcher())
        cmp_pub, cmp, cmp_bytes, loop = _run_loop_then_read(streams, tmp_path, cuda_device, "compact", n_groups, attempts, raw, cfg_kw, counts, ref_model, make_patcher())
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
        if oov:
            ids = torch.cat([b.input_ids.flatten().cpu() for part in cmp for b in part if not b.sentinel])
            assert int(ids.max()) < 280 and int((ids == 5).sum()) > 0, "out-of-vocabulary ids were patched before the records were gathered"
        if kl:
            assert any(bool((b.ref_logprobs != b.old_logprobs).any()) for part in cmp for b in part if not b.sentinel), "the reference policy's column arrived"
        if sp > 1:  # the slices of one micro-batch sit side by side in the partitions of its SP group and add up to the padded length
            for lead in range(0, trainers, sp):
                for group in zip(*[cmp[lead + k] for k in range(sp)]):
                    assert len({b.input_ids.shape[1] for b in group}) == 1 and len({b.sentinel for b in group}) == 1
                    assert all(torch.equal(group[0].seq_boundaries, b.seq_boundaries) for b in group)
        # bytes in the logs: 12-16 per token (+ 4 with the reference column) per COPY of the record against 68 per token of the expanded batch
        # spread over the slices - an SP group of 2 holds each compact record twice
        assert cmp_bytes * (3 if sp == 1 and not kl else 1.5) < full_bytes * (sp if sp > 1 else 1), (cmp_bytes, full_bytes)
        if kl:
            assert loop.prof.get("ref_logprobs", 0) > 0 and loop.prof.get("k6_plan_launch", 0) > 0  # K6 + the policy's forward ran here; only the column left the GPU
        else:
            assert "k6_plan_launch" not in loop.prof  # the preprocessor never packed
        assert loop.prof.get("publish_submit", 0) > 0
    finally:
        streams.reset_streams_backend()


def _run_loop_then_read(streams, tmp_path, cuda_device, wire, n_groups, attempts, raw, cfg_kw, counts, ref_model=None, oov_patcher=None):
    from pipelinerl_amd.preprocess import PreprocessorConfig, PreprocessorLoop
    from pipelinerl_amd.ragged import RaggedRollouts

    exp = tmp_path / wire
    exp.mkdir()
    with streams.write_to_streams(streams.SingleStreamSpec(exp_path=exp, topic="actor")) as w:
        for g in range(n_groups):
            w.write(RaggedRollouts.from_entries(raw[g * attempts:(g + 1) * attempts]))
    loop = PreprocessorLoop(PreprocessorConfig(exp_path=exp, **cfg_kw), cuda_device, wire=wire, profile=True, ref_model=ref_model, oov_patcher=oov_patcher)
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
