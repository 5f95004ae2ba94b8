"""`TrainerState`: a follower of the trainer's message topic (role of reference
pipelinerl/state.py:20-65).

Actor, rollout workers, preprocessor and launcher each keep one.  A daemon thread tails the
`weight_update_request` topic and folds every message into three facts: the last weight version
that reached the inference servers, how many samples the trainer has consumed (the
preprocessor's back-pressure input, preprocess.py:587-592) and whether training has ended.
"""

from __future__ import annotations

import logging
import threading
import time
from pathlib import Path
from typing import Callable

from . import finetune_loop as fl
from .streams import SingleStreamSpec, read_stream

logger = logging.getLogger(__name__)


class TrainerState:
    #: how often the blocking waits re-check / log
    POLL_S = 0.05
    LOG_EVERY_S = 1.0

    def __init__(self, exp_path: Path):
        self.exp_path = exp_path
        self.propagated_weight_version: int | None = None
        self.samples_processed: int | None = None
        self.training_done: bool = False
        self._done = threading.Event()
        self._thread: threading.Thread | None = None
        self._handlers: dict[type, Callable] = {
            fl.WeightUpdateSuccess: self._on_weights,
            fl.SamplesProcessed: self._on_samples,
            fl.TrainingDone: self._on_done,
        }

    # -- message handlers ------------------------------------------------------------------------
    def _on_weights(self, m: "fl.WeightUpdateSuccess") -> None:
        self.propagated_weight_version = m.version

    def _on_samples(self, m: "fl.SamplesProcessed") -> None:
        self.samples_processed = m.samples_processed

    def _on_done(self, _m: "fl.TrainingDone") -> None:
        self.training_done = True
        self._done.set()

    def apply(self, message) -> None:
        handler = self._handlers.get(type(message))
        if handler is not None:  # WeightUpdateRequest and unknown kinds are not state
            handler(message)

    # -- lifecycle -------------------------------------------------------------------------------
    def debug_mode_init(self) -> None:
        """Stage-isolation modes without a trainer (debug.mode actor / preprocessor): version 0,
        nothing consumed, training over."""
        self.apply(fl.WeightUpdateSuccess(version=0))
        self.apply(fl.SamplesProcessed(samples_processed=0))
        self.apply(fl.TrainingDone())

    def start_listening(self) -> None:
        spec = SingleStreamSpec(exp_path=self.exp_path, topic=fl.TRAINER_TOPIC)

        def follow() -> None:
            with read_stream(spec) as reader:
                for record in reader.read():
                    self.apply(fl.parse_trainer_message(record))

        self._thread = threading.Thread(target=follow, name="trainer-state", daemon=True)
        self._thread.start()

    # -- blocking accessors ------------------------------------------------------------------------
    def wait_for_training_done(self, timeout: float | None = None) -> bool:
        return self._done.wait(timeout=timeout)

    def _block_until_known(self, attr: str, what: str):
        logged = 0.0
        while (value := getattr(self, attr)) is None:
            now = time.time()
            if now - logged >= self.LOG_EVERY_S:
                logger.info("Waiting for the trainer to declare %s", what)
                logged = now
            time.sleep(self.POLL_S)
        return value

    def wait_for_processed_samples(self) -> int:
        return self._block_until_known("samples_processed", "the number of processed samples")

    def wait_for_model_version(self) -> int:
        return self._block_until_known("propagated_weight_version", "the initial weight version")
