"""Golden trainer control-plane traffic, produced by EXECUTING the reference's own code:

  * the trainer message models (`pipelinerl/finetune_loop.py:138-171`: `TRAINER_TOPIC`, `ParameterInfo`,
    `WeightUpdateRequest`, `WeightUpdateSuccess`, `SamplesProcessed`, `TrainingDone`, `TrainerMessage`)
    — the module cannot be imported (deepspeed, ring_flash_attn ... absent), so that block is cut from
    the source at generation time and `exec`ed; nothing of it is stored here;
  * `pipelinerl/state.py` (`TrainerState`) imported for real on top of it;
  * `pipelinerl/streams.py` (`files` backend) imported for real with the stand-ins of
    make_streams_golden.py.

The script writes a message sequence the way the trainer does, lets the reference `TrainerState`
follow the stream, and records the stream file plus the state after every message.

    python tests/golden/make_trainer_golden.py
"""

from __future__ import annotations

import json
import sys
import tempfile
import time
import types
from pathlib import Path
from typing import Literal

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
SRC = Path("/root/reference/pipelinerl/finetune_loop.py")


def reference_message_block() -> str:
    lines = SRC.read_text().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith("TRAINER_TOPIC ="))
    end = next(i for i, l in enumerate(lines) if l.startswith("class WeightUpdateManager"))
    return "\n".join(lines[start:end])


def load_reference():
    from make_streams_golden import install_stubs

    install_stubs()
    sys.path.insert(0, "/root/reference")
    from pydantic import BaseModel

    fake = types.ModuleType("pipelinerl.finetune_loop")
    ns = fake.__dict__
    ns.update(BaseModel=BaseModel, Literal=Literal, time=time)
    sys.modules["pipelinerl.finetune_loop"] = fake  # pydantic resolves the annotations through the module
    exec(compile(reference_message_block(), "reference_trainer_messages", "exec"), ns)
    for k in ("TRAINER_TOPIC", "ParameterInfo", "WeightUpdateRequest", "WeightUpdateSuccess", "SamplesProcessed", "TrainingDone", "TrainerMessage"):
        assert k in ns, k
    from pipelinerl import state, streams

    return fake, state, streams


def message_sequence(m):
    T = 1234.5  # fixed timestamps: deterministic fixture
    info = [m.ParameterInfo(name="model.embed_tokens.weight", shape=[16, 8], dtype="torch.bfloat16"),
            m.ParameterInfo(name="lm_head.weight", shape=[16, 8], dtype="torch.float32")]
    return [
        m.SamplesProcessed(samples_processed=0, timestamp=T),
        m.WeightUpdateRequest(version=1, parameters_info=info, timestamp=T),
        m.WeightUpdateSuccess(version=1, timestamp=T),
        m.SamplesProcessed(samples_processed=64, timestamp=T),
        m.SamplesProcessed(samples_processed=128, timestamp=T),
        m.WeightUpdateSuccess(version=2, timestamp=T),
        m.TrainingDone(timestamp=T),
    ]


def main() -> None:
    m, state, streams = load_reference()
    streams.set_streams_backend("files")
    with tempfile.TemporaryDirectory() as td:
        exp = Path(td)
        spec = streams.SingleStreamSpec(exp_path=exp, topic=m.TRAINER_TOPIC)
        ts = state.TrainerState(exp)
        trace = [{"after": "init", "propagated_weight_version": ts.propagated_weight_version, "samples_processed": ts.samples_processed,
                  "training_done": ts.training_done}]
        with streams.write_to_streams(spec) as w:
            w.write(message_sequence(m)[0])  # the reader waits for the file to exist
            ts.start_listening()
            time.sleep(0.5)
            trace.append({"after": 0, "propagated_weight_version": ts.propagated_weight_version, "samples_processed": ts.samples_processed,
                          "training_done": ts.training_done})
            for i, msg in enumerate(message_sequence(m)[1:], start=1):
                w.write(msg)
                time.sleep(0.35)
                trace.append({"after": i, "propagated_weight_version": ts.propagated_weight_version,
                              "samples_processed": ts.samples_processed, "training_done": ts.training_done})
        done = ts.wait_for_training_done(timeout=1.0)
        files = {str(p.relative_to(exp)): p.read_text() for p in sorted(exp.rglob("*")) if p.is_file()}
    out = {"topic": m.TRAINER_TOPIC, "files": files, "state_trace": trace, "wait_for_training_done": bool(done),
           "dumps": [x.model_dump() for x in message_sequence(m)]}
    (HERE / "trainer_messages.json").write_text(json.dumps(out, indent=1))
    for t in trace:
        print(t)
    print(files)


if __name__ == "__main__":
    main()
