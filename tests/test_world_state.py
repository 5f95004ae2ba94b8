"""GPU partition arithmetic against the reference's WorldMap (golden), trainer-state follower and
trainer messages over the files stream (CPU only)."""

import json

import pytest

from helpers import GOLDEN


@pytest.mark.parametrize("rec", json.loads((GOLDEN / "world.json").read_text()), ids=lambda r: f"{r['total']}g-{r['actor_fraction']}:{r['preprocessor_fraction']}:{r['finetune_fraction']}-tp{r['tp']}pp{r['pp']}-r{r['replicas']}")
def test_split_gpus_matches_reference(rec):
    from pipelinerl_amd.world import split_gpus

    kw = dict(total_gpus=rec["total"], actor_fraction=rec["actor_fraction"], preprocessor_fraction=rec["preprocessor_fraction"],
              finetune_fraction=rec["finetune_fraction"], tensor_parallel_size=rec["tp"], pipeline_parallel_size=rec["pp"], replicas=rec["replicas"])
    if rec["error"] is not None:
        with pytest.raises(ValueError):
            split_gpus(**kw)
        return
    p = split_gpus(**kw)
    assert p.total_finetune_gpus == rec["total_finetune_gpus"]
    assert len(p.actor_gpus) == rec["gpus_per_actor"] * rec["replicas"]
    assert len(p.preprocessor_gpus) == rec["gpus_per_preprocessor"] * rec["replicas"]
    assert p.llms_per_actor == rec["llms_per_actor"]
    assert p.total_actor_llms == rec["total_actor_llms"]
    assert p.weight_update_group_size == rec["weight_update_group_size"]
    assert sorted(p.actor_gpus + p.preprocessor_gpus + p.finetune_gpus) == list(range(rec["total"]))


def test_baseline_configs_partition():
    """BASELINE.json configs: 2+2 on 4 GPUs, 4+4 on 8, TP=2 x 2 actors + 4 learners."""
    from pipelinerl_amd.world import round_up_accumulation_passes, split_gpus

    assert split_gpus(4).total_finetune_gpus == 2 and split_gpus(4).weight_update_group_size == 3
    assert split_gpus(8).total_finetune_gpus == 4 and split_gpus(8).weight_update_group_size == 5
    p = split_gpus(8, tensor_parallel_size=2)
    assert (p.total_actor_llms, p.gpus_per_llm, p.total_finetune_gpus) == (2, 2, 4)
    assert round_up_accumulation_passes(4096, 4) == 4096 and round_up_accumulation_passes(10, 4) == 12


def test_trainer_state_follows_messages(tmp_path):
    from pipelinerl_amd import streams
    from pipelinerl_amd.finetune_loop import TRAINER_TOPIC, SamplesProcessed, TrainingDone, WeightUpdateSuccess, parse_trainer_message
    from pipelinerl_amd.state import TrainerState

    streams.reset_streams_backend()
    streams.set_streams_backend("files")
    try:
        st = TrainerState(tmp_path)
        spec = streams.SingleStreamSpec(exp_path=tmp_path, topic=TRAINER_TOPIC)
        with streams.write_to_streams(spec) as w:
            w.write(SamplesProcessed(samples_processed=0))
            st.start_listening()
            assert st.wait_for_processed_samples() == 0
            w.write(WeightUpdateSuccess(version=16))
            assert st.wait_for_model_version() == 16
            w.write(SamplesProcessed(samples_processed=32))
            w.write(TrainingDone())
            assert st.wait_for_training_done(timeout=5)
            assert st.samples_processed == 32 and st.training_done
        msg = parse_trainer_message({"kind": "weight_update_request", "version": 3, "parameters_info": [{"name": "w", "shape": [2, 3], "dtype": "torch.bfloat16"}]})
        assert msg.parameters_info[0].shape == [2, 3] and msg.transport == "per_tensor"  # no field = reference trainer
        with pytest.raises(ValueError):
            parse_trainer_message({"kind": "nope"})
        dbg = TrainerState(tmp_path)
        dbg.debug_mode_init()
        assert dbg.propagated_weight_version == 0 and dbg.samples_processed == 0 and dbg.wait_for_training_done(0)
    finally:
        streams.reset_streams_backend()


def test_trainer_messages_and_state_match_reference(tmp_path):
    """tests/golden/trainer_messages.json: the reference's own message models, files backend and
    TrainerState executed on a message sequence.  (a) this package's models dump the same fields,
    (b) its TrainerState, following the file the REFERENCE wrote, passes through the same states."""
    import json
    import time

    from helpers import GOLDEN
    from pipelinerl_amd import finetune_loop as fl
    from pipelinerl_amd import streams
    from pipelinerl_amd.state import TrainerState

    g = json.loads((GOLDEN / "trainer_messages.json").read_text())
    assert fl.TRAINER_TOPIC == g["topic"]
    # (a) every reference dump parses into the same-kind model, and our dump carries the reference's
    # fields with equal values (extra keys are the MI355X transport extensions with defaults)
    extensions = {"transport", "bucket_bytes", "ipc_handles", "ipc_nbytes", "ipc_max_allocation", "tp_size"}
    for want in g["dumps"]:
        msg = fl.parse_trainer_message(want)
        got = msg.model_dump()
        assert got["kind"] == want["kind"]
        assert {k: got[k] for k in want} == want
        assert set(got) - set(want) <= extensions
    # (b) follow the reference-written stream line by line
    rel, text = next(iter(g["files"].items()))
    lines = text.splitlines(keepends=True)
    path = tmp_path / rel
    path.parent.mkdir(parents=True)
    streams.reset_streams_backend()
    streams.set_streams_backend("files")
    try:
        st = TrainerState(tmp_path)
        snap = lambda: {"propagated_weight_version": st.propagated_weight_version, "samples_processed": st.samples_processed,  # noqa: E731
                        "training_done": st.training_done}
        trace = g["state_trace"]
        assert snap() == {k: trace[0][k] for k in snap()}
        with open(path, "a") as f:
            f.write(lines[0])
            f.flush()
            st.start_listening()
            for i, want in enumerate(trace[1:]):
                if i > 0:
                    f.write(lines[i])
                    f.flush()
                want_state = {k: want[k] for k in ("propagated_weight_version", "samples_processed", "training_done")}
                deadline = time.monotonic() + 5.0
                while snap() != want_state and time.monotonic() < deadline:
                    time.sleep(0.02)
                assert snap() == want_state, (i, snap(), want_state)
        assert st.wait_for_training_done(timeout=1.0) == g["wait_for_training_done"]
    finally:
        streams.reset_streams_backend()


def test_validate_packing_config_like_the_reference():
    import types

    from pipelinerl_amd.finetune_loop import validate_packing_config

    validate_packing_config(types.SimpleNamespace(seq_packing=False, use_flash_attention=False))
    validate_packing_config(types.SimpleNamespace(seq_packing=True, use_flash_attention=True))
    with pytest.raises(ValueError, match="Sequence packing requires flash attention"):
        validate_packing_config(types.SimpleNamespace(seq_packing=True, use_flash_attention=False))
