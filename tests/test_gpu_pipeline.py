"""End-to-end replay on the GPU: recorded `actor` stream -> PreprocessorLoop (K5, scheduler, K6)
-> `training_data` stream -> LearnerStep.step() with the HIP rl_step on a small torch model.
Mirrors the reference's stage-isolation test mode (`debug.mode=finetune+preprocessor` with
`debug.streams_from`, launch.py:554-564,692-693)."""

import queue
import threading
import types

import pytest
import torch

from oracle import preprocess as opre
from oracle import rl_loss as orl

pytestmark = pytest.mark.gpu


class TinyLM(torch.nn.Module):
    def __init__(self, vocab=64, dim=16):
        super().__init__()
        self.emb = torch.nn.Embedding(vocab, dim)
        self.head = torch.nn.Linear(dim, vocab)

    def forward(self, input_ids=None, **kw):
        return types.SimpleNamespace(logits=self.head(self.emb(input_ids)).float())


class TinyHFLM(torch.nn.Module):
    """The Hugging Face causal-LM layout: `.model` returns the last hidden states, `.lm_head` is a bias-free Linear.
    A reference policy of this shape goes through the MFMA head (`fused_head.annotate_ref_logprobs_fused`)."""

    class Body(torch.nn.Module):
        def __init__(self, vocab, dim):
            super().__init__()
            self.emb = torch.nn.Embedding(vocab, dim)
            self.calls = 0

        def forward(self, input_ids=None, **kw):
            self.calls += 1
            return types.SimpleNamespace(last_hidden_state=torch.tanh(self.emb(input_ids)).to(torch.bfloat16))

    def __init__(self, vocab=64, dim=64):
        super().__init__()
        self.model = TinyHFLM.Body(vocab, dim)
        self.lm_head = torch.nn.Linear(dim, vocab, bias=False)
        self.logits_calls = 0

    def forward(self, input_ids=None, **kw):
        self.logits_calls += 1
        return types.SimpleNamespace(logits=self.lm_head(self.model(input_ids=input_ids).last_hidden_state.float()))


def _oracle_ref_logprobs(ref_model, b, temperature=1.0):
    """log p_ref of the labelled tokens, 0 elsewhere, from `oracle.rl_loss.logprob_entropy` (the restatement of reference
    rl/__init__.py:207-213) fed the reference model's fp32 logits - the CPU oracle, not a second device computation."""
    import numpy as np

    from oracle import rl_loss as orl

    with torch.no_grad():
        if hasattr(ref_model, "lm_head"):  # fp64 product of the bf16 hidden states and the fp32 weight, rounded once
            h = ref_model.model(input_ids=b.input_ids).last_hidden_state
            logits = (h.double() @ ref_model.lm_head.weight.double().t()).float()
            ref_model.model.calls -= 1
        else:
            logits = ref_model(input_ids=b.input_ids, attention_mask=b.attention_mask, position_ids=b.position_ids).logits.float()
    nlp = orl.logprob_entropy(logits.cpu().numpy(), b.input_ids.cpu().numpy(), temperature)[0]
    want = np.zeros(tuple(b.input_ids.shape), dtype=np.float64)
    want[:, 1:] = nlp
    want[b.labels.cpu().numpy() == -100] = 0.0
    return torch.from_numpy(want).to(b.input_ids.device)


@pytest.mark.parametrize("backend", ["files", "shm"])
def test_replay_actor_stream_through_preprocessor_and_learner(libprl, cuda_device, tmp_path, backend):
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune_loop import TRAINER_TOPIC, LearnerStep, run_data_loader
    from pipelinerl_amd.preprocess import PreprocessorConfig, PreprocessorLoop
    from pipelinerl_amd.ragged import RaggedRollouts
    from pipelinerl_amd.state import TrainerState
    from pipelinerl_amd.synthetic import make_entries

    streams.reset_streams_backend()
    streams.set_streams_backend(backend, **({"segment_bytes": 1 << 20, "owner": True} if backend == "shm" else {}))
    try:
        attempts, V = 4, 64
        raw = make_entries(6, attempts=attempts, seq_length=48, vocab=V, seed=21, prompt_min=3, prompt_max=8)
        rl = RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.0, final_kl_coef=0.0,
                      divide_advantage_by_std=False, clamp_log_ratio_ref_new_value=5)
        cfg = PreprocessorConfig(exp_path=tmp_path, num_trainers=1, train_batch_size=1, gradient_accumulation_passes=8,
                                 seq_length=96, attempts=attempts, rl=rl, eos_token_id=2, chunk_n_groups=2)
        actor_spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="actor")
        published = {}

        errors = []

        def preprocessor():
            try:
                _preprocessor()
            except Exception as e:  # noqa: BLE001 - surfaced by the main thread
                import traceback

                errors.append(traceback.format_exc())
                raise e

        def _preprocessor():
            with streams.write_to_streams(actor_spec) as w:
                for g in range(6):
                    group = raw[g * attempts:(g + 1) * attempts]
                    # shm backend: rollouts travel as binary ragged SoA records; files: the text record
                    w.write(RaggedRollouts.from_entries(group) if backend == "shm" else group)
                loop = PreprocessorLoop(cfg, cuda_device)
                published["n"] = loop.run(max_published_samples=16, idle_timeout=3.0)

        t = threading.Thread(target=preprocessor, daemon=True)
        t.start()

        torch.manual_seed(0)
        model = TinyLM(V).to(cuda_device)
        before = [p.detach().clone() for p in model.parameters()]
        step = LearnerStep(model, torch.optim.SGD(model.parameters(), lr=0.5), rl, train_batch_size=1, gradient_accumulation_passes=8,
                           max_train_steps=10, send_weight_updates=False,
                           trainer_stream=streams.SingleStreamSpec(exp_path=tmp_path, topic=TRAINER_TOPIC))
        q: queue.Queue = queue.Queue(maxsize=2)
        data_spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="training_data", partition=0)
        threading.Thread(target=run_data_loader, args=(data_spec, q, cuda_device), daemon=True).start()
        steps, first = 0, None
        while steps < 2:
            while True:  # poll so that a dead preprocessor fails the test instead of hanging it
                try:
                    batch = q.get(timeout=0.5)
                    break
                except queue.Empty:
                    assert not errors, errors[0]
                    assert t.is_alive() or not q.empty(), "preprocessor exited without producing the expected batches"
            if isinstance(batch, Exception):
                raise batch
            assert batch.input_ids.is_cuda and batch.input_ids.shape[1] <= 96
            if first is None:
                # oracle check of the very first micro-batch with the model's own logits
                with torch.no_grad():
                    logits = model(input_ids=batch.input_ids).logits
                b = {k: v.cpu().numpy() for k, v in batch.tensors()}
                ocfg = rl.model_copy()
                ocfg.batch_size = 8
                first = orl.rl_step(logits.cpu().numpy(), b, ocfg, 0, 10, True)
            res = step.step(batch)
            if first is not None and "checked" not in first:
                assert abs(res["stats"]["loss"] - float(first["loss"])) <= 1e-4 * max(abs(float(first["loss"])), 1e-6)
                first["checked"] = True
            assert torch.isfinite(res["loss"]).item()
            steps += int(res["did_optimizer_step"])
        assert step.metrics.samples == 16 and step.metrics.completed_steps == 2
        assert any(not torch.equal(a, p.detach()) for a, p in zip(before, model.parameters()))
        step.finish()
        t.join(timeout=20)
        assert published["n"] == 16
        # both transports are logs: a follower that starts after the trainer finished still sees every message
        st = TrainerState(tmp_path)
        st.start_listening()
        assert st.wait_for_training_done(timeout=10) and st.samples_processed == 16
    finally:
        streams.reset_streams_backend()


def test_native_step_equals_per_micro_batch_rl_step(libprl, cuda_device):
    """NativeLearnerStep (one K6 launch, fused logits kernel per micro-batch, ONE stats launch) gives
    the same parameter gradients and the same summed statistics as the drop-in rl_step loop."""
    import copy

    from pipelinerl_amd.finetune.rl import RLConfig, rl_step
    from pipelinerl_amd.finetune_loop import NativeLearnerStep
    from pipelinerl_amd.hotpath import HotPathStep, dense_micro_batches
    from pipelinerl_amd.synthetic import make_ragged

    V = 128
    rag_h, _ = make_ragged(4, attempts=4, seq_length=40, vocab=V, seed=3, prompt_min=3, prompt_max=8, with_ref=True)
    rag = rag_h.to(cuda_device)
    mbs = dense_micro_batches(rag_h, 100)
    rl = RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.05, final_kl_coef=0.05,
                  divide_advantage_by_std=True, clamp_log_ratio_ref_new_value=5, entropy_bonus=0.01, final_entropy_bonus=0.01)
    torch.manual_seed(0)
    model_a = TinyLM(V).to(cuda_device)
    model_b = copy.deepcopy(model_a)

    # A: native step (lr = 0 so that the gradients stay inspectable: SGD with lr 0 keeps params)
    captured = {}
    opt_a = torch.optim.SGD(model_a.parameters(), lr=0.0)
    orig_zero = opt_a.zero_grad
    opt_a.zero_grad = lambda *a, **k: captured.update(g=[p.grad.detach().clone() for p in model_a.parameters()]) or orig_zero(*a, **k)
    native = NativeLearnerStep(model_a, opt_a, rl, eos_token_id=2, samples_per_step=16, max_train_steps=10)
    res = native.step(rag, mbs)
    stats_a = native.stats_dict(res["stats"])

    # B: drop-in loop over the same micro-batches
    cfg_b = rl.model_copy()
    cfg_b.batch_size = 16
    hp = HotPathStep(cfg_b, 2, 0, 10)
    batches = hp.preprocess(rag, mbs)
    agg = {}
    for b in batches:
        loss, st = rl_step(model_b, b, 0, 10, cfg_b)
        loss.backward()
        for k, v in st.items():
            agg.setdefault(k, []).append(v)
    for ga, pb in zip(captured["g"], model_b.parameters()):
        assert torch.allclose(ga, pb.grad, rtol=1e-4, atol=1e-7)
    assert abs(stats_a["loss"] - sum(agg["loss"])) <= 1e-5 * max(1.0, abs(sum(agg["loss"])))
    for k in ("reward", "entropy", "kl", "ratio_new_old", "ratio_new_old_sum", "advantage", "token_weight"):
        assert abs(stats_a[k] - sum(agg[k])) <= 1e-4 * max(1.0, abs(sum(agg[k]))), k
    for k in ("max_kl", "max_advantage", "max_reward"):
        assert abs(stats_a[k] - max(agg[k])) <= 1e-6 * max(1.0, abs(max(agg[k]))), k
    assert stats_a["num_output_tokens_sum"] == sum(agg["num_output_tokens_sum"])
    assert res["micro_batches"] == len(mbs) and native.metrics.completed_steps == 1


def test_rl_step_drives_a_huggingface_causal_lm(libprl, cuda_device):
    """Drop-in check with the model class the reference trains (transformers Qwen2ForCausalLM,
    random init, tiny config): `rl_step(model, batch, ...)` calls `model(**inputs)`, reads
    `outputs.logits`, and the backward reaches every parameter.  Loss / gradients equal a plain
    torch implementation of the same PPO objective on the model's own logits."""
    transformers = pytest.importorskip("transformers")
    from pipelinerl_amd.finetune.data import collate
    from pipelinerl_amd.finetune.rl import RLConfig, rl_step
    from pipelinerl_amd.synthetic import make_entries
    from oracle import preprocess as opre

    V = 512
    cfg_m = transformers.Qwen2Config(vocab_size=V, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                                     num_key_value_heads=2, max_position_embeddings=256, tie_word_embeddings=False)
    torch.manual_seed(0)
    model = transformers.Qwen2ForCausalLM(cfg_m).to(cuda_device).float()
    raw = make_entries(2, attempts=4, seq_length=40, vocab=V, seed=17, prompt_min=4, prompt_max=10)
    data = opre.preprocess_chunk(raw, 2, False)
    tok = types.SimpleNamespace(eos_token_id=2, padding_side="right")
    batch = collate([{k: v for k, v in e.items() if k != "finish_reason"} for e in data], tok)  # [8, Lp] padded batch on the GPU
    rl = RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.0, final_kl_coef=0.0, batch_size=8,
                  divide_advantage_by_std=False)
    loss, stats = rl_step(model, batch, 0, 10, rl)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters()}
    assert all(g is not None and torch.isfinite(g).all() for g in grads.values())
    assert any(g.abs().max() > 0 for g in grads.values())
    # plain torch reference of the same objective
    model.zero_grad()
    logits = model(input_ids=batch.input_ids, attention_mask=batch.attention_mask).logits.float()
    lp = torch.log_softmax(logits[:, :-1], -1).gather(2, batch.input_ids[:, 1:, None])[..., 0]
    m = (batch.labels[:, 1:] != -100).float()
    ratio = torch.exp(lp - batch.old_logprobs[:, 1:])
    adv = batch.advantages[:, 1:]
    pol = torch.minimum(ratio * adv, ratio.clamp(0.8, 1.2) * adv)
    ref_loss = -(pol * m / 8).sum()
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * max(abs(ref_loss.item()), 1e-6)
    for n, p in model.named_parameters():
        assert torch.allclose(grads[n], p.grad, rtol=2e-3, atol=1e-6), n
    assert stats["num_output_tokens_sum"] == int(m.sum().item()) and stats["input_size"] == batch.input_ids.numel()


@pytest.mark.parametrize("layout", ["logits_model", "hf_layout_fused_head"])
def test_native_step_with_reference_model_on_the_learner(libprl, cuda_device, layout):
    """KL-to-reference with the reference policy living on the learner GPU: NativeLearnerStep(ref_model=...)
    must equal the drop-in loop in which every batch was annotated by `annotate_ref_logprobs`, and the
    annotated column must equal the ORACLE's log-probs of the reference model's logits.  `hf_layout_fused_head`: the
    reference policy exposes body and head separately and its head runs on the MFMA kernels (no logits, its
    `forward` is never called)."""
    import copy

    from pipelinerl_amd.finetune.rl import RLConfig, annotate_ref_logprobs, rl_step
    from pipelinerl_amd.finetune_loop import NativeLearnerStep
    from pipelinerl_amd.hotpath import HotPathStep, dense_micro_batches
    from pipelinerl_amd.synthetic import make_ragged

    V = 96
    rag_h, _ = make_ragged(3, attempts=4, seq_length=36, vocab=V, seed=9, prompt_min=3, prompt_max=8)
    rag = rag_h.to(cuda_device)
    mbs = dense_micro_batches(rag_h, 90)
    rl = RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.3, final_kl_coef=0.3, clamp_log_ratio_ref_new_value=5,
                  divide_advantage_by_std=False)
    torch.manual_seed(1)
    model_a = TinyLM(V).to(cuda_device)
    model_b = copy.deepcopy(model_a)
    torch.manual_seed(2)
    fused = layout == "hf_layout_fused_head"
    ref_model = (TinyHFLM(V) if fused else TinyLM(V)).to(cuda_device).eval()  # a DIFFERENT policy: the KL term is not zero

    captured = {}
    opt_a = torch.optim.SGD(model_a.parameters(), lr=0.0)
    orig_zero = opt_a.zero_grad
    opt_a.zero_grad = lambda *a, **k: captured.update(g=[p.grad.detach().clone() for p in model_a.parameters()]) or orig_zero(*a, **k)
    native = NativeLearnerStep(model_a, opt_a, rl, eos_token_id=2, samples_per_step=12, max_train_steps=10, ref_model=ref_model)
    res = native.step(rag, mbs)
    stats_a = native.stats_dict(res["stats"])
    assert stats_a["kl"] > 1e-4  # the reference policy really differs
    if fused:
        assert ref_model.logits_calls == 0 and ref_model.model.calls == len(mbs)

    cfg_b = rl.model_copy()
    cfg_b.batch_size = 12
    hp = HotPathStep(cfg_b, 2, 0, 10)
    batches = hp.preprocess(rag, mbs)
    agg = {}
    for b in batches:
        annotate_ref_logprobs(ref_model, b, cfg_b.temperature)
        want = _oracle_ref_logprobs(ref_model, b, cfg_b.temperature)
        assert torch.allclose(b.ref_logprobs.double(), want, rtol=1e-4, atol=2e-5)
        loss, st = rl_step(model_b, b, 0, 10, cfg_b)
        loss.backward()
        for k, v in st.items():
            agg.setdefault(k, []).append(v)
    for ga, pb in zip(captured["g"], model_b.parameters()):
        assert torch.allclose(ga, pb.grad, rtol=1e-4, atol=1e-7)
    for k in ("loss", "kl", "ref_logprobs", "ratio_ref_new"):
        assert abs(stats_a[k] - sum(agg[k])) <= 1e-4 * max(1.0, abs(sum(agg[k]))), k


@pytest.mark.parametrize("layout", ["logits_model", "hf_layout_fused_head"])
def test_preprocessor_fills_ref_logprobs_from_a_model_on_its_gpu(libprl, cuda_device, tmp_path, layout):
    """PreprocessorLoop(ref_model=...): the published batches carry log p_ref of every labelled token
    (what the reference fetches over HTTP, preprocess.py:86-104), equal to the oracle's log-probs."""
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.preprocess import PreprocessorConfig, PreprocessorLoop
    from pipelinerl_amd.synthetic import make_entries

    streams.reset_streams_backend()
    streams.set_streams_backend("files")
    try:
        attempts, V = 2, 64
        raw = make_entries(4, attempts=attempts, seq_length=40, vocab=V, seed=33, prompt_min=3, prompt_max=8)
        rl = RLConfig(policy_loss="ppo", kl_coef=0.1, final_kl_coef=0.1, divide_advantage_by_std=False)
        cfg = PreprocessorConfig(exp_path=tmp_path, num_trainers=1, train_batch_size=1, gradient_accumulation_passes=8,
                                 seq_length=80, attempts=attempts, rl=rl, eos_token_id=2, chunk_n_groups=2)
        torch.manual_seed(5)
        ref_model = (TinyHFLM(V) if layout == "hf_layout_fused_head" else TinyLM(V)).to(cuda_device).eval()
        with streams.write_to_streams(streams.SingleStreamSpec(exp_path=tmp_path, topic="actor")) as w:
            for g in range(4):
                w.write(raw[g * attempts:(g + 1) * attempts])
        n = PreprocessorLoop(cfg, cuda_device, ref_model=ref_model).run(max_published_samples=8, idle_timeout=2.0)
        assert n == 8
        seen = 0
        with streams.read_stream(streams.SingleStreamSpec(exp_path=tmp_path, topic="training_data", partition=0)) as r:
            for rec in r.read():
                b = PipelineBatchEncoding(**rec).to_device(cuda_device)
                if b.sentinel:
                    continue
                want = _oracle_ref_logprobs(ref_model, b)
                assert torch.allclose(b.ref_logprobs.double(), want, rtol=1e-4, atol=2e-5)
                assert not torch.equal(b.ref_logprobs, b.old_logprobs)
                seen += int(b.seq_boundaries.shape[0]) - 1
                if seen >= 8:
                    break
        assert seen == 8
    finally:
        streams.reset_streams_backend()


def _write_actor_groups(streams, tmp_path, raw, attempts, n_groups, binary):
    from pipelinerl_amd.ragged import RaggedRollouts

    spec = streams.SingleStreamSpec(exp_path=tmp_path, topic="actor")
    with streams.write_to_streams(spec) as w:
        for g in range(n_groups):
            group = raw[g * attempts:(g + 1) * attempts]
            w.write(RaggedRollouts.from_entries(group) if binary else group)


def _collect(streams, tmp_path, topic, partition, n, timeout=20.0):
    out = []

    def run():
        with streams.read_stream(streams.SingleStreamSpec(exp_path=tmp_path, topic=topic, partition=partition)) as r:
            for rec in r.read():
                out.append(rec)
                if len(out) >= n:
                    return

    t = threading.Thread(target=run, daemon=True)
    t.start()
    t.join(timeout)
    return out


def test_preprocessor_unpacked_mode_publishes_padded_batches(libprl, cuda_device, tmp_path):
    """seq_packing = False: fixed train_batch_size rows per micro-batch through K7 (reference `collate`,
    preprocess.py:639-648), equal to the oracle's collate of the same samples."""
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.preprocess import PreprocessorConfig, PreprocessorLoop
    from pipelinerl_amd.synthetic import make_entries

    streams.reset_streams_backend()
    streams.set_streams_backend("shm", owner=True)
    try:
        attempts = 4
        raw = make_entries(4, attempts=attempts, seq_length=40, vocab=64, seed=5, prompt_min=3, prompt_max=8)
        rl = RLConfig(divide_advantage_by_std=False)
        cfg = PreprocessorConfig(exp_path=tmp_path, num_trainers=1, train_batch_size=4, gradient_accumulation_passes=2, seq_length=64,
                                 attempts=attempts, rl=rl, eos_token_id=2, seq_packing=False, padding_side="left")
        _write_actor_groups(streams, tmp_path, raw, attempts, 4, binary=True)
        n = PreprocessorLoop(cfg, cuda_device).run(max_published_samples=16, idle_timeout=2.0)
        assert n == 16
        recs = _collect(streams, tmp_path, "training_data", 0, 4)
        data = opre.preprocess_chunk(raw, 2, False)
        for k, rec in enumerate(recs):
            got = PipelineBatchEncoding(**rec)
            want = opre.collate(data[k * 4:(k + 1) * 4], "left")
            assert not got.is_packed and got.input_ids.shape == want["input_ids"].shape
            for key in ("input_ids", "labels", "attention_mask", "rewards", "advantages", "old_logprobs", "group_tokens", "num_labels", "overflow"):
                assert torch.equal(getattr(got, key).cpu(), torch.from_numpy(want[key])), key
    finally:
        streams.reset_streams_backend()


def test_preprocessor_sequence_parallel_slices_and_counts(libprl, cuda_device, tmp_path):
    """seq_parallel = 2: every packed micro-batch is padded with a filler sequence to an even length
    (data.py:222-230), cut in two slices for trainers (lead, lead + 1), `padding` is carried, and the
    trainer-side sequence count ignores the filler (finetune_loop.py:303-312)."""
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.finetune_loop import get_batch_sequence_count
    from pipelinerl_amd.preprocess import PreprocessorConfig, PreprocessorLoop
    from pipelinerl_amd.synthetic import make_entries

    streams.reset_streams_backend()
    streams.set_streams_backend("shm", owner=True)
    try:
        attempts = 4
        raw = make_entries(4, attempts=attempts, seq_length=31, vocab=64, seed=9, prompt_min=3, prompt_max=8)
        cfg = PreprocessorConfig(exp_path=tmp_path, num_trainers=2, train_batch_size=1, gradient_accumulation_passes=8, seq_length=64,
                                 attempts=attempts, rl=RLConfig(divide_advantage_by_std=False), eos_token_id=2, seq_parallel=2)
        _write_actor_groups(streams, tmp_path, raw, attempts, 4, binary=False)
        loop = PreprocessorLoop(cfg, cuda_device)
        assert loop.run(max_published_samples=8, idle_timeout=2.0) >= 8
        a = _collect(streams, tmp_path, "training_data", 0, 3)
        b = _collect(streams, tmp_path, "training_data", 1, 3)
        assert len(a) == len(b) == 3
        total_seqs = 0
        for ra, rb in zip(a, b):
            sa, sb = PipelineBatchEncoding(**ra), PipelineBatchEncoding(**rb)
            assert sa.input_ids.shape == sb.input_ids.shape and sa.padding == sb.padding
            full = torch.cat([sa.input_ids, sb.input_ids], dim=1)
            assert full.shape[1] % 2 == 0
            if sa.padding:
                assert (torch.cat([sa.labels, sb.labels], dim=1)[0, -sa.padding:] == -100).all()
            starts = int((torch.cat([sa.position_ids, sb.position_ids], dim=1) == 0).sum())  # sequences incl. the filler
            assert get_batch_sequence_count(sa) == get_batch_sequence_count(sb) == starts - (1 if sa.padding else 0)
            total_seqs += get_batch_sequence_count(sa)
        assert 3 <= total_seqs <= loop.sched.published_samples
    finally:
        streams.reset_streams_backend()


def test_preprocessor_drops_old_samples_and_reports_stats(libprl, cuda_device, tmp_path):
    """A ring of 6 samples fed with 24: with `pop_old_data` the oldest are dropped before they are
    scheduled (reference :572-585), the newest survive, and `preprocessor_stats` records are written."""
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.preprocess import PreprocessorConfig, PreprocessorLoop
    from pipelinerl_amd.state import TrainerState
    from pipelinerl_amd.synthetic import make_entries

    streams.reset_streams_backend()
    streams.set_streams_backend("shm", owner=True)
    try:
        attempts = 4
        raw = make_entries(6, attempts=attempts, seq_length=40, vocab=64, seed=11, prompt_min=3, prompt_max=8)
        cfg = PreprocessorConfig(exp_path=tmp_path, num_trainers=1, train_batch_size=1, gradient_accumulation_passes=4, seq_length=200,
                                 attempts=attempts, rl=RLConfig(divide_advantage_by_std=False), eos_token_id=2, chunk_n_groups=6,
                                 ring_buffer_size=6, log_every_n_samples=1)
        _write_actor_groups(streams, tmp_path, raw, attempts, 6, binary=True)
        loop = PreprocessorLoop(cfg, cuda_device)
        n = loop.run(max_published_samples=4, idle_timeout=2.0)
        assert loop.ring.popped == 24 - 6 and n == 4  # one chunk of 24 samples through a ring of 6
        recs = _collect(streams, tmp_path, "training_data", 0, 1)
        first = PipelineBatchEncoding(**recs[0])
        # the first published sample is sample 18 of the chunk: the 18 older ones were dropped
        want_ids = raw[18]["input_ids"]
        assert first.input_ids[0, :len(want_ids)].tolist() == want_ids
        stats = _collect(streams, tmp_path, "preprocessor_stats", 0, 1)
        assert stats and stats[0]["preprocessor/published_samples"] >= 4 and "preprocessor/queue/raw" in stats[0]
    finally:
        streams.reset_streams_backend()


def test_oov_patch_kernel_matches_the_reference(libprl, cuda_device):
    import json

    from helpers import GOLDEN
    from pipelinerl_amd.preprocess import OovPatcher
    from pipelinerl_amd.ragged import RaggedRollouts

    g = json.loads((GOLDEN / "preprocess_loop.json").read_text())["oov"]
    entries = [{"input_ids": e["input_ids"], "labels": e["labels"], "logprobs": e["logprobs"], "reward": 0.0, "group_id": "g", "finished": True,
                "metadata": {"model_version": 0, "rollout_index": i, "step_index": 0}} for i, e in enumerate(g["data"])]
    rag = RaggedRollouts.from_entries(entries).to(cuda_device)
    patcher = OovPatcher(g["vocab_ids"], g["the_token_id"], cuda_device)
    patcher.apply(rag)
    toks = rag.tokens.cpu().tolist()
    off = rag.host_seq_off
    assert [toks[off[i]:off[i + 1]] for i in range(len(entries))] == g["patched_input_ids"]
    labs = rag.labels.cpu().tolist()
    assert [labs[off[i]:off[i + 1]] for i in range(len(entries))] == g["labels_after"]
    n_bad = sum(a != b for e, p in zip(g["data"], g["patched_input_ids"]) for a, b in zip(e["input_ids"], p))
    assert int(patcher.count.item()) == n_bad


def test_native_step_finiteness_assert_over_every_row_is_a_switch(libprl, cuda_device):
    """The reference asserts `isfinite(new_logprobs)` over EVERY position (rl/__init__.py:213).  The native step skips the logits
    rows that predict an unlabelled token by default (they reach neither the loss nor a statistic), so a NaN there passes;
    `skip_unlabelled=False` reads every row and the step's statistics raise the reference's assert."""
    from pipelinerl_amd.finetune.rl import RLConfig
    from pipelinerl_amd.finetune_loop import NativeLearnerStep
    from pipelinerl_amd.hotpath import dense_micro_batches
    from pipelinerl_amd.synthetic import make_ragged

    V = 64
    rag_h, _ = make_ragged(2, attempts=4, seq_length=40, vocab=V, seed=9, prompt_min=4, prompt_max=8)
    rag = rag_h.to(cuda_device)
    mbs = dense_micro_batches(rag_h, 100)
    rl = RLConfig(policy_loss="ppo", epsilon_low=0.2, epsilon_high=0.2, kl_coef=0.0, final_kl_coef=0.0, divide_advantage_by_std=False,
                  clamp_log_ratio_ref_new_value=5)

    class PoisonedLM(TinyLM):
        """NaN logits in row 1 of every micro-batch: it predicts token 2, a PROMPT token (prompts are >= 4 tokens) - unlabelled."""

        def forward(self, input_ids=None, **kw):
            logits = self.head(self.emb(input_ids)).float()
            mask = torch.zeros_like(logits)
            mask[:, 1] = float("nan")
            return types.SimpleNamespace(logits=logits + mask)

    for skip, raises in ((True, False), (False, True)):
        torch.manual_seed(0)
        model = PoisonedLM(V).to(cuda_device)
        native = NativeLearnerStep(model, torch.optim.SGD(model.parameters(), lr=0.0), rl, eos_token_id=2, samples_per_step=8, max_train_steps=10,
                                   skip_unlabelled=skip)
        res = native.step(rag, mbs)
        if raises:
            with pytest.raises(AssertionError, match="new_logprobs is not finite"):
                native.stats_dict(res["stats"])
        else:
            stats = native.stats_dict(res["stats"])
            assert stats["num_output_tokens_sum"] > 0 and torch.isfinite(res["loss"]).item()
