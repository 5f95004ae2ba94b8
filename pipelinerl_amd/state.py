"""`TrainerState`: a follower of the trainer's message topic (reference pipelinerl/state.py:20-65).

Actor, rollout workers, preprocessor and launcher each keep one; a daemon thread tails the
`weight_update_request` topic and tracks the last propagated weight version, the number of samples
the trainer has consumed (back-pressure input of the preprocessor, preprocess.py:587-592) and the
end of training.
"""

from __future__ import annotations

import logging
import threading
import time
from pathlib import Path

from .finetune_loop import (
    TRAINER_TOPIC,
    SamplesProcessed,
    TrainingDone,
    WeightUpdateSuccess,
    parse_trainer_message,
)
from .streams import SingleStreamSpec, read_stream

logger = logging.getLogger(__name__)


class TrainerState:
    def __init__(self, exp_path: Path):
        self.exp_path = exp_path
        self.propagated_weight_version: int | None = None
        self.samples_processed: int | None = None
        self.training_done: bool = False
        self._training_done_event = threading.Event()
        self._thread: threading.Thread | None = None

    def debug_mode_init(self) -> None:
        """No trainer around (debug.mode actor / preprocessor): pretend version 0, nothing consumed."""
        self.propagated_weight_version = 0
        self.samples_processed = 0
        self.training_done = True
        self._training_done_event.set()

    def _apply(self, message) -> None:
        if isinstance(message, WeightUpdateSuccess):
            self.propagated_weight_version = message.version
        elif isinstance(message, SamplesProcessed):
            self.samples_processed = message.samples_processed
        elif isinstance(message, TrainingDone):
            self.training_done = True
            self._training_done_event.set()

    def start_listening(self) -> None:
        stream = SingleStreamSpec(exp_path=self.exp_path, topic=TRAINER_TOPIC)

        def listen():
            with read_stream(stream) as reader:
                for record in reader.read():
                    self._apply(parse_trainer_message(record))

        self._thread = threading.Thread(target=listen, daemon=True)
        self._thread.start()

    def wait_for_training_done(self, timeout: float | None = None) -> bool:
        return self._training_done_event.wait(timeout=timeout)

    def _wait_for(self, attr: str, what: str, poll: float = 0.05, log_every: float = 1.0):
        last = 0.0
        while getattr(self, attr) is None:
            if time.time() - last >= log_every:
                logger.info(f"Waiting for the trainer to declare {what}")
                last = time.time()
            time.sleep(poll)
        return getattr(self, attr)

    def wait_for_processed_samples(self):
        return self._wait_for("samples_processed", "the number of processed samples")

    def wait_for_model_version(self):
        return self._wait_for("propagated_weight_version", "the initial weight version")
