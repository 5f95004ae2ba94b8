"""The batched-transfer layout of the preprocessor loop (pipelinerl_amd/staging.py, finetune/data.py) on the host: a CPU
"device" runs the same packing, alignment and view arithmetic as the GPU path (the copies themselves are covered by
tests/test_gpu_pipeline.py and tests/test_gpu_actor_flow.py)."""

import numpy as np
import torch

from pipelinerl_amd.finetune.data import PackedStep, _alloc_outputs, _column_stride
from pipelinerl_amd.staging import PinnedStager
from pipelinerl_amd.synthetic import make_ragged


def test_upload_packs_many_arrays_into_one_buffer_of_aligned_views():
    st = PinnedStager("cpu", slots=2, min_bytes=1 << 10)
    arrays = [np.arange(7, dtype=np.int32), None, np.linspace(0, 1, 1001, dtype=np.float32), np.zeros(0, dtype=np.int64),
              torch.arange(5, dtype=torch.float64), np.arange(300, dtype=np.uint8).astype(np.uint8), np.arange(12, dtype=np.int64).reshape(3, 4)]
    out = st.upload(arrays)
    assert out[1] is None and st.uploads == 1
    for a, t in zip(arrays, out):
        if a is None:
            continue
        want = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        assert t.dtype == want.dtype and tuple(t.shape) == tuple(want.shape) and torch.equal(t, want)
        assert t.storage_offset() * t.element_size() % 256 == 0
    live = [t for t in out if t is not None and t.numel()]
    assert len({t.untyped_storage().data_ptr() for t in live}) == 1  # views of ONE allocation
    # the ring: a slot is reused, earlier results stay intact (they are copies on the "device")
    again = st.upload([np.full(7, 9, dtype=np.int32)])
    assert torch.equal(out[0], torch.arange(7, dtype=torch.int32)) and int(again[0][0]) == 9
    assert st.upload([None, np.zeros(0, dtype=np.float32)])[1].numel() == 0


def test_ragged_rollouts_travel_in_one_upload_with_the_k5_plan():
    from pipelinerl_amd.finetune.rl import plan_groups

    rag, _ = make_ragged(3, attempts=4, seq_length=64, vocab=100, seed=2, prompt_min=3, prompt_max=9, with_ref=True)
    plan = plan_groups(rag.host_group_index, rag.host_step_index, rag.host_rollout_index)
    st = PinnedStager("cpu", min_bytes=1 << 10)
    dev, plan_dev = rag.to("cpu", stager=st, extra=plan)
    assert st.uploads == 1 and len(plan_dev) == 5
    for name in ("tokens", "labels", "logprobs", "ref_logprobs", "seq_off", "lp_off", "reward", "group_index", "step_index", "rollout_index",
                 "model_version", "finished", "finish_code"):
        assert torch.equal(getattr(dev, name), getattr(rag, name)), name
    for a, t in zip(plan, plan_dev):
        assert np.array_equal(t.numpy(), a)
    assert dev.host_seq_off is rag.host_seq_off and dev.group_ids == rag.group_ids  # host metadata rides along untouched


def test_output_block_columns_are_aligned_and_come_back_in_one_download():
    for total in (1, 63, 64, 1001, 8192):
        out = _alloc_outputs(total, torch.device("cpu"), packed=True)
        block = out.pop("__block__")
        assert block.numel() == _column_stride(total) * 68
        for k, v in out.items():
            assert v.numel() == total and v.is_contiguous() and (v.data_ptr() - block.data_ptr()) % 256 == 0, (total, k)  # device blocks start 256-B aligned
            v.fill_(len(k))
        step = PackedStep({**out, "__block__": block}, np.array([0, total], dtype=np.int64), np.array([0, 1], dtype=np.int64),
                          np.zeros(1, dtype=np.int64), None)
        host = step.to_host(PinnedStager("cpu", min_bytes=1 << 10))
        b = host[0]
        for k in out:
            assert torch.equal(getattr(b, k)[0], out[k]), k
        assert b.input_ids.shape == (1, total) and b.is_packed
