"""Host-side logic of the path (CPU only): group filtering / group-size check against outputs of
the reference functions (executed from source, tests/golden/make_schedule_golden.py),
PipelineBatchEncoding slicing, stats aggregation rules, rollout plugin types."""

import json

import numpy as np
import pytest
import torch

from helpers import GOLDEN


@pytest.mark.parametrize("case", json.loads((GOLDEN / "filter_groups.json").read_text()), ids=lambda c: f"n{len(c['data'])}")
def test_group_filter_and_size_check_match_reference(case):
    from pipelinerl_amd.preprocess import check_group_sizes, filter_zero_advantage_groups

    kept, dropped = filter_zero_advantage_groups(list(case["data"]))
    assert [e["uid"] for e in kept] == case["kept_uids"] and dropped == case["dropped"]
    for gs, ok in case["group_size_ok"].items():
        assert check_group_sizes(case["data"], int(gs)) == ok


def _packed_batch(T=16, n_seq=3):
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    i64 = lambda: torch.arange(T, dtype=torch.long).unsqueeze(0)  # noqa: E731
    f32 = lambda: torch.arange(T, dtype=torch.float32).unsqueeze(0) / 7  # noqa: E731
    return PipelineBatchEncoding(
        input_ids=i64(), attention_mask=torch.ones(1, T, dtype=torch.long), labels=i64(), position_ids=i64(), segment_ids=i64() // 6,
        rewards=f32(), advantages=f32(), ref_logprobs=f32(), old_logprobs=f32(), group_tokens=f32() + 1, num_labels=f32() + 1,
        overflow=f32() * 0, model_version=4, is_packed=True, seq_boundaries=[0, 6, 12, 16], padding=0)


def test_batch_encoding_slices_and_coercion():
    """make_slices (types.py:145-180): equal token ranges, metadata shared; list inputs coerced to
    the reference's dtypes; unknown keys ignored; missing required fields rejected."""
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding

    b = _packed_batch()
    parts = b.make_slices(4)
    assert len(parts) == 4 and all(p.input_ids.shape == (1, 4) for p in parts)
    assert torch.equal(torch.cat([p.old_logprobs for p in parts], dim=1), b.old_logprobs)
    assert all(p.model_version == 4 and p.is_packed and torch.equal(p.seq_boundaries, b.seq_boundaries) for p in parts)
    with pytest.raises(ValueError):
        b.make_slices(5)  # 16 % 5 != 0
    with pytest.raises(ValueError):
        b.make_slices(32)
    d = {k: (v.tolist() if isinstance(v, torch.Tensor) else v) for k, v in b.model_dump().items()}
    d["some_future_field"] = 1
    c = PipelineBatchEncoding(**d)
    assert c.input_ids.dtype == torch.long and c.rewards.dtype == torch.float32 and c.seq_boundaries.dtype == torch.int32
    assert torch.equal(c.labels, b.labels) and c.pixel_values is None and c.sentinel is False
    e = PipelineBatchEncoding.from_dict({"extra": 3}, **{k: v for k, v in d.items() if k != "some_future_field"})
    assert e.model_extra == {"extra": 3}
    with pytest.raises(ValueError):
        PipelineBatchEncoding(input_ids=[[1]], attention_mask=[[1]], labels=[[1]], model_version=0)
    unpacked = PipelineBatchEncoding(**{**d, "position_ids": None, "is_packed": False})
    with pytest.raises(ValueError):
        unpacked.make_slices(2)


def test_sequence_count_and_token_count():
    from pipelinerl_amd.finetune_loop import get_batch_sequence_count, get_batch_token_count
    from pipelinerl_amd.finetune.utils import create_sentinel_batch

    b = _packed_batch()
    assert get_batch_sequence_count(b) == 3 and get_batch_token_count(b) == 16
    b.padding = 4  # the last "sequence" is the sequence-parallel filler (finetune_loop.py:302-312)
    assert get_batch_sequence_count(b) == 2
    s = create_sentinel_batch(None)
    assert get_batch_sequence_count(s) == 1 and s.sentinel and int((s.labels != -100).sum()) == 0


def test_aggregate_rl_stats_rules():
    """Aggregation op by key substring, in the reference's precedence (rl/utils.py:9-23)."""
    from pipelinerl_amd.finetune.rl.utils import aggregate_rl_stats, effective_sample_size

    stats = {"min_loss": [3.0, -1.0], "max_kl": [0.5, 2.0], "loss": [1.0, 2.0], "ratio_new_old_sum": [10.0, 30.0],
             "ratio_new_old_squared_sum": [12.0, 28.0], "num_output_tokens_sum": [20, 20], "reward": [4.0, 8.0], "max_loss": [1.0, 5.0]}
    out = aggregate_rl_stats(stats, num_samples=4)
    assert out["rl/min_loss"] == -1.0 and out["rl/max_kl"] == 2.0 and out["rl/max_loss"] == 5.0
    assert out["rl/loss"] == 3.0 and out["rl/ratio_new_old_sum"] == 40.0 and out["rl/reward"] == 3.0
    assert effective_sample_size(out) == pytest.approx(40.0 ** 2 / 40.0 / 40.0)


def test_linear_decay_and_loss_config():
    from pipelinerl_amd.finetune.rl import RLConfig, linear_decay_coef, make_loss_config

    assert linear_decay_coef(3, 10, 0.3, 0.1) == pytest.approx(0.3 + (0.1 - 0.3) * 0.3)
    c, kl, ent = make_loss_config(RLConfig(policy_loss="reinforce", epsilon_low=0.1, epsilon_high=0.3, kl_coef=0.3, final_kl_coef=0.1, batch_size=8), 3, 10)
    assert c.policy_loss == 1 and c.clip_lo == np.float32(0.9) and c.clip_hi == np.float32(1.3) and c.token_weight == np.float32(0.125)
    assert kl == pytest.approx(0.24) and ent == 0 and c.use_entropy_loss == 0
    with pytest.raises(ValueError):
        make_loss_config(RLConfig(policy_loss="dpo"), 0, 1)
    assert RLConfig(**{"policy_loss": "ppo", "aggregate_loss": "sum"}).policy_loss == "ppo"  # unknown yaml keys are ignored


def test_rollout_plugin_types_and_stamping():
    from pipelinerl_amd.ragged import RaggedRollouts
    from pipelinerl_amd.rollouts import BaseMetrics, RolloutResult, TrainingText, resolve_plugin, stamp_group, summarize_training_texts

    t = TrainingText(text="prompt answer", n_predicted=6, input_ids=[5, 6, 7, 8], labels=[-100, -100, 7, 8], logprobs=[-0.1, -0.2], reward=1.0, finished=True)
    t.check_consistency()
    assert t.prompt_text == "prompt " and t.output_text == "answer"
    bad = t.model_copy(update={"logprobs": [-0.1]})
    with pytest.raises(ValueError):
        bad.check_consistency()
    results = [RolloutResult(training_texts=[t.model_copy(deep=True)], metrics=BaseMetrics(reward=1, success=True, no_error=True, no_answer=False), latency=0.1) for _ in range(3)]
    record = stamp_group(results, "problem-7", model_version=48)
    assert [e["metadata"]["rollout_index"] for e in record] == [0, 1, 2] and all(e["group_id"] == "problem-7" for e in record)
    rag = RaggedRollouts.from_entries(record)  # the stream record is directly ingestible
    assert rag.n_seqs == 3 and rag.host_model_version.tolist() == [48, 48, 48] and rag.finished.tolist() == [1, 1, 1]
    assert summarize_training_texts([t]).overflow is False
    assert resolve_plugin("pipelinerl_amd.rollouts.stamp_group") is stamp_group
    with pytest.raises(ValueError):
        resolve_plugin("nodots")


class _ToyTok:
    """Whitespace tokenizer with offsets (the generator of tests/golden/text_path.json uses the same)."""

    eos_token_id, padding_side = 2, "right"

    def __call__(self, text, return_offsets_mapping=True, max_length=None, truncation=True):
        ids, offs, pos = [], [], 0
        for w in text.split(" "):
            if w:
                ids.append(3 + (sum(map(ord, w)) % 50))
                offs.append((pos, pos + len(w)))
            pos += len(w) + 1
        if max_length is not None:
            ids, offs = ids[:max_length], offs[:max_length]
        return {"input_ids": ids, "attention_mask": [1] * len(ids), "offset_mapping": offs}


def test_text_path_helpers_match_reference():
    """mask_labels / validate_spans / preprocess_fn on text entries (data.py:47-160) against outputs
    of the reference functions."""
    from pipelinerl_amd.finetune.data import mask_labels, preprocess_fn, validate_spans

    g = json.loads((GOLDEN / "text_path.json").read_text())
    for c in g["cases"]:
        spans = [tuple(s) for s in c["spans"]]
        labels, mids = mask_labels(c["input_ids"], [tuple(o) for o in c["offset_mapping"]], spans)
        assert labels == c["labels"] and mids == c["midpoints"]
        entry = dict(c["entry"])
        if "predicted_spans" in entry:
            entry["predicted_spans"] = [tuple(s) for s in entry["predicted_spans"]]
        out = preprocess_fn(entry, _ToyTok(), seq_length=c["seq_length"], is_rl=False)
        for k, v in c["preprocess"].items():
            assert out[k] == v, k
    for b in g["invalid"]:
        spans = [tuple(s) for s in b["spans"]]
        if b["error"] is None:
            validate_spans("hello world", spans)
        else:
            with pytest.raises(ValueError):
                validate_spans("hello world", spans)


def test_make_rl_data_callback():
    from pipelinerl_amd.finetune.rl import RLConfig, make_rl_data_callback, populate_rl_data

    cb = make_rl_data_callback(None, None, RLConfig(divide_advantage_by_std=False), None)
    assert cb.func is populate_rl_data and cb.keywords["config"].divide_advantage_by_std is False
    assert make_rl_data_callback(None, None, None, None) is None


def test_sample_accounting_matches_reference():
    """get_batch_token_count / get_batch_sequence_count / calculate_train_steps against the
    reference's own functions (tests/golden/make_counts_golden.py executes them on the batches its
    collate functions produced - the same batches stored in preprocess_*.npz)."""
    import json
    import types

    import torch

    from helpers import GOLDEN, PREPROCESS_CASES, load_preprocess_case
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.finetune.utils import create_sentinel_batch
    from pipelinerl_amd.finetune_loop import calculate_train_steps, get_batch_sequence_count, get_batch_token_count

    g = json.loads((GOLDEN / "batch_counts.json").read_text())

    def build(d):
        kw = {}
        for k, v in d.items():
            if k.startswith("__"):
                continue
            kw[k] = torch.from_numpy(v) if v.ndim > 0 else v.item()
        return PipelineBatchEncoding(**kw)

    seen = 0
    for name in PREPROCESS_CASES:
        case = load_preprocess_case(name)
        for kind in ("packed", "padded"):
            for plan, arrays in case[kind].items():
                want = g["batches"][f"{name}/{kind}/{plan}"]
                b = build(arrays)
                assert get_batch_token_count(b) == want["tokens"] and get_batch_sequence_count(b) == want["sequences"], (name, kind, plan)
                if "padding" in want:
                    assert int(b.padding) == want["padding"]
                seen += 1
    assert seen == len(g["batches"]) - 1
    s = create_sentinel_batch(device="cpu", tokenizer=types.SimpleNamespace(eos_token_id=7), model_version=5)
    assert get_batch_token_count(s) == g["batches"]["sentinel"]["tokens"] and get_batch_sequence_count(s) == g["batches"]["sentinel"]["sequences"]
    for rec in g["train_steps"]:
        args = types.SimpleNamespace(interrupt_train_steps=rec["cfg_interrupt"], max_train_steps=rec["max"])
        assert calculate_train_steps(args, rec["arg"]) == rec["result"], rec


def test_aggregate_rl_stats_matches_reference():
    """tests/golden/aggregate.json: the reference's aggregate_rl_stats / linear_decay_coef run on the
    statistics of the rl_step fixtures."""
    import json

    from helpers import GOLDEN
    from pipelinerl_amd.finetune.rl import linear_decay_coef
    from pipelinerl_amd.finetune.rl.utils import aggregate_rl_stats

    g = json.loads((GOLDEN / "aggregate.json").read_text())
    for case in g["cases"]:
        got = aggregate_rl_stats(case["stats"], case["num_samples"])
        assert list(got) == list(case["out"])
        for k, want in case["out"].items():
            assert got[k] == pytest.approx(want, rel=2e-6, abs=1e-9), k  # fp32 summation order only
    for rec in g["linear_decay"]:
        assert linear_decay_coef(*rec["args"]) == rec["out"]


def test_rollout_models_match_reference():
    """tests/golden/rollouts_models.json: field tables and dumps of the reference's own plugin models."""
    import json
    import sys

    from helpers import GOLDEN
    from pipelinerl_amd import rollouts

    sys.path.insert(0, str(GOLDEN))
    from make_rollouts_golden import describe

    want = json.loads((GOLDEN / "rollouts_models.json").read_text())
    got = json.loads(json.dumps(describe(rollouts)))
    assert got == want


def test_rlconfig_accepts_the_reference_keys_with_the_reference_defaults():
    import json
    import sys

    from helpers import GOLDEN
    from pipelinerl_amd.finetune.rl import RLConfig

    sys.path.insert(0, str(GOLDEN))
    from make_rollouts_golden import field_table

    want = json.loads((GOLDEN / "rlconfig_fields.json").read_text())
    got = json.loads(json.dumps(field_table(RLConfig)))
    assert {k: got[k] for k in want} == want
    assert set(got) - set(want) == {"fused_logits_grad", "inplace_logits_grad", "expected_loss_scale", "skip_unlabelled_rows",
                                    "fused_head_keep_logits", "fused_head_chunk_rows"}  # MI355X extensions


def test_host_stats_with_and_without_a_value_head():
    """`host_stats`: the step's statistics vector -> the reference's dict.  With the value head's five entries appended
    the reported (and asserted) loss is policy + value_loss_coef * value_loss and the five keys close the dict, in the
    reference's order (rl/__init__.py:381-386, 399-401, 441-448) - checked against the c18 golden's own numbers."""
    from helpers import load_rl_case
    from pipelinerl_amd._lib import PRL_NUM_STATS, STAT_INDEX
    from pipelinerl_amd.finetune.rl import VALUE_STAT_KEYS, host_stats

    case = load_rl_case("c18_ppo_value_head")
    want = case["stats"]
    coef = case["config"]["value_loss_coef"]
    dev = [0.0] * PRL_NUM_STATS
    for k, i in STAT_INDEX.items():
        if k in want:
            dev[i] = want[k]
    dev[STAT_INDEX["loss"]] = want["loss"] - coef * want["value_loss"]  # what K2+K3 report: the policy part
    dev[STAT_INDEX["num_sequences"]] = round(want["kl_coef"] / 0.01)
    vs = [want[k] for k in VALUE_STAT_KEYS]
    got = host_stats(torch.tensor(dev + vs, dtype=torch.float64), int(want["input_size"]), 0.01, 0.0, coef)
    assert list(got) == list(want) and len(got) == 37
    for k, w in want.items():
        assert abs(got[k] - w) <= 1e-6 * max(1.0, abs(w)), k
    plain = host_stats(torch.tensor(dev, dtype=torch.float64), int(want["input_size"]), 0.01, 0.0)
    assert list(plain) == list(want)[:32] and abs(plain["loss"] - dev[STAT_INDEX["loss"]]) < 1e-6
    dev[STAT_INDEX["num_output_tokens_sum"]] = 0
    assert host_stats(torch.tensor(dev + vs, dtype=torch.float64), 7, 0.01, 0.0, coef) == {"input_size": 7.0}
    dev[STAT_INDEX["loss"]] = float("inf")
    with pytest.raises(AssertionError, match="Non-finite loss"):
        host_stats(torch.tensor(dev + vs, dtype=torch.float64), 7, 0.01, 0.0, coef)


def test_empty_example_lists_raise_what_the_reference_raises():
    """collate([]) / collate_packed([]) index examples[0] in the reference (data.py:170, 246): IndexError, before any device work."""
    from pipelinerl_amd.finetune.data import collate, collate_packed

    tok = type("Tok", (), {"eos_token_id": 2, "padding_side": "right"})()
    with pytest.raises(IndexError):
        collate_packed([], tok, seq_parallel=1)
    with pytest.raises(IndexError):
        collate([], tok)
    from pipelinerl_amd.finetune.rl import RLConfig, populate_rl_data

    with pytest.raises(KeyError, match="group_id"):  # pandas' column selection on an empty frame (rl/__init__.py:456-459)
        populate_rl_data([], 2, RLConfig())
