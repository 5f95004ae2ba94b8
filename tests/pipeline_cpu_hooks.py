"""Stand-ins that let `pipelinerl_amd.pipeline_run` run its N-learner x M-engine TOPOLOGY on host tensors (`PipelineSpec(platform="cpu",
hooks="pipeline_cpu_hooks")`, tests/test_pipeline_topology_cpu.py): the stages, streams, schedulers, sample accounting, process groups and
the weight-update protocol are the product's; the three pieces that need a HIP device are replaced HERE, in tests/:

  build_policy        a four-tensor language model with the Hugging Face names the parameter probe looks for
  rl_step_fn          a torch loss with `rl_step`'s signature (masked log-prob x advantage over the global batch size: partial losses add)
  preprocessor_stage  the product's `MicroBatchScheduler` fed by the ORACLE's preprocess_fn + populate_rl_data + collate_packed
                      (test infrastructure; the product's preprocessor computes K5 / K6 on the GPU and has no CPU path)
"""

from __future__ import annotations

import time
import types
from pathlib import Path

import numpy as np
import torch


class _Body(torch.nn.Module):
    def __init__(self, vocab: int, dim: int):
        super().__init__()
        self.embed_tokens = torch.nn.Embedding(vocab, dim)
        self.norm = torch.nn.LayerNorm(dim)

    def forward(self, input_ids=None, **kw):
        return (self.norm(self.embed_tokens(input_ids)),)


class TinyPolicy(torch.nn.Module):
    def __init__(self, vocab: int, dim: int = 16):
        super().__init__()
        self.model = _Body(vocab, dim)
        self.lm_head = torch.nn.Linear(dim, vocab, bias=False)

    def forward(self, input_ids=None, **kw):
        return types.SimpleNamespace(logits=self.lm_head(self.model(input_ids=input_ids)[0]))


def build_policy(spec, device, seed: int):
    torch.manual_seed(seed)
    return TinyPolicy(spec.shape["vocab"]).to(device)


def rl_step_fn(model, batch, current_step, max_step, config, seq_parallel_group=None):
    logits = model(input_ids=batch.input_ids, attention_mask=batch.attention_mask, labels=batch.labels).logits
    lp = torch.log_softmax(logits[:, :-1].float(), -1).gather(2, batch.input_ids[:, 1:, None])[..., 0]
    mask = (batch.labels[:, 1:] != -100).float()
    loss = -(lp * batch.advantages[:, 1:] * mask).sum() / config.batch_size
    n = int(mask.sum().item())
    if n == 0:
        return loss, {"input_size": float(batch.input_ids.numel())}
    return loss, {"loss": loss.item(), "num_output_tokens_sum": n, "ratio_new_old_sum": float(n), "ratio_new_old_squared_sum": float(n)}


def preprocessor_stage(spec) -> None:
    from oracle import preprocess as opre
    from pipelinerl_amd import pipeline_run, streams
    from pipelinerl_amd.finetune.types import PipelineBatchEncoding
    from pipelinerl_amd.finetune.utils import create_sentinel_batch
    from pipelinerl_amd.preprocess import MicroBatchScheduler
    from pipelinerl_amd.ragged import RaggedRollouts
    from pipelinerl_amd.state import TrainerState

    pipeline_run._set_backend(spec)
    exp = Path(spec.exp_path)
    state = TrainerState(exp)
    state.start_listening()
    state.wait_for_processed_samples()
    sched = MicroBatchScheduler(num_trainers=spec.n_learners, train_batch_size=1, gradient_accumulation_passes=spec.global_batch, seq_length=spec.budget,
                                length_of=lambda e: len(e["input_ids"]))
    target = spec.steps * spec.global_batch
    tok = types.SimpleNamespace(eos_token_id=2)
    log: list = []  # (trainer, uids or "sentinel") in emission order: the test reads the schedule back
    t0 = time.perf_counter()
    steps_done = 0
    out_spec = streams.StreamRangeSpec(exp_path=exp, topic="training_data", partition_range=(0, spec.n_learners))
    with streams.read_stream(streams.SingleStreamSpec(exp_path=exp, topic="actor")) as reader, streams.write_to_streams(out_spec) as writer:
        chunk: list = []
        for rec in reader.read():
            group = rec.to_entries() if isinstance(rec, RaggedRollouts) else rec
            chunk.append(group)
            if len(chunk) < spec.chunk_n_groups:
                continue
            entries = [e for g in chunk for e in g]
            chunk = []
            sched.push(opre.preprocess_chunk(entries, 2, False))
            while sched.queue and steps_done < spec.steps:
                mbs, done = sched.drain()
                for mb in mbs:
                    if mb.sentinel:
                        b = create_sentinel_batch(None, tokenizer=tok, model_version=0)
                        log.append([mb.trainer_id, "sentinel"])
                    else:
                        d = opre.collate_packed(mb.samples, 2, 1)
                        b = PipelineBatchEncoding(**{k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in d.items()})
                        log.append([mb.trainer_id, [len(s["input_ids"]) for s in mb.samples]])
                    writer.write(b, partition=mb.trainer_id)
                steps_done += int(done)
                if not mbs and not done:
                    break
            if steps_done >= spec.steps:
                break
    assert sched.published_samples >= target
    pipeline_run._report(spec, "preprocessor", {"published_samples": sched.published_samples, "wall_s": time.perf_counter() - t0, "busy_s": 0.0, "busy_frac": 0.0,
                                                "backpressure_waits": 0, "queue_depth": {}, "schedule": log, "samples_per_trainer": sched.samples_per_trainer})
    state.wait_for_training_done(timeout=spec.stage_timeout_s)  # the logs this process wrote stay mapped until the learners are done
