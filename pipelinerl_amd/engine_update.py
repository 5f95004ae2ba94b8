"""Engine side of the in-flight weight update: the manager the reference hangs on its vLLM API server
(pipelinerl/vllm1.py:137-186, 244-249) restated engine-agnostic, plus the HTTP endpoint the trainer posts to.

    POST /receive_weight_update  (body: `WeightUpdateRequest`)                      finetune_loop.py:180-192
      -> InflightUpdateManager.receive_weight_update(request)
           async with update_lock:                    one update at a time          vllm1.py:160
             await engine.pause_generation(mode="keep", clear_cache=False)          vllm1.py:164: requests in flight are KEPT,
                                                                                    they continue on the new weights
             await engine.collective_rpc("receive_weight_update", (request_json,))  every worker: `WorkerExtension`
             finally: await engine.resume_generation()                              vllm1.py:176-183
      <- {"status": "ok"} once the weights are in place (the POST blocks, vllm1.py:244-249)

vLLM-ROCm is not part of this image; the manager needs only the three coroutines above, so it drives any engine that has
them (`ScriptedEngine` here: workers are `WorkerExtension`s in this process, "generation" is a loop of device work that the
pause really stops).  The three durations the reference logs per update (pause, update, resume) are kept in `timings`.
"""

from __future__ import annotations

import asyncio
import json
import logging
import threading
import time
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from typing import Any, Callable, Sequence

logger = logging.getLogger(__name__)


class InflightUpdateManager:
    """`WeightUpdateManager` of reference vllm1.py:137-186 (the ENGINE-side one; the trainer-side class of the same name is
    `finetune_loop.WeightUpdateManager`)."""

    def __init__(self, engine: Any):
        self.engine = engine
        self.update_lock = asyncio.Lock()
        self.timings: list[dict[str, float]] = []

    async def input_process_groups(self, actor_llm_idx: int, actor_ngpus: int, init_method: str, world_size: int) -> None:
        """Every worker joins the trainer's update group (vllm1.py:144-153)."""
        await self.engine.collective_rpc("init_actor_update_group", args=(actor_llm_idx, actor_ngpus, init_method, world_size))

    async def receive_weight_update(self, request: Any) -> dict[str, float]:
        payload = request if isinstance(request, str) else (request.model_dump_json() if hasattr(request, "model_dump_json") else json.dumps(request))
        version = (json.loads(payload) if isinstance(payload, str) else payload).get("version", "unknown")
        async with self.update_lock:
            t0 = time.perf_counter()
            logger.info(f"Pausing generation for weight update version={version}")
            await self.engine.pause_generation(mode="keep", clear_cache=False)
            t1 = time.perf_counter()
            t2 = t1
            try:
                await self.engine.collective_rpc("receive_weight_update", args=(payload,))
                t2 = time.perf_counter()
                logger.info(f"Weight update processed version={version} in {t2 - t1:.3f}s")
            finally:
                t_res = time.perf_counter()
                await self.engine.resume_generation()
                t3 = time.perf_counter()
                rec = {"version": version, "pause_s": t1 - t0, "update_s": t2 - t1, "resume_s": t3 - t_res, "total_s": t3 - t0}
                self.timings.append(rec)
        return rec

    async def close_communicator(self) -> None:
        await self.engine.collective_rpc("close_communicator")


class ScriptedEngine:
    """An engine with the three coroutines the manager needs.  `workers`: `WorkerExtension` objects (one per engine GPU).
    `generate_step`: optional callable doing one quantum of "generation" work (e.g. a forward pass on the worker's weights); a
    background thread calls it in a loop while generation is not paused - `pause_generation` returns only when the loop
    is parked BETWEEN two quanta (mode "keep": nothing in flight is dropped, the next quantum runs on the new weights)."""

    def __init__(self, workers: Sequence[Any], generate_step: Callable[[], Any] | None = None):
        self.workers = list(workers)
        self.generate_step = generate_step
        # ONE lock guards both facts - "generation may run" and "a quantum is running": the loop decides to start a quantum and
        # marks it under the lock, so a pause that has cleared `_want_run` under the same lock sees either a quantum to wait for or
        # none that can still start.  (Two independent events left a window: a pause right after a resume saw the previous
        # park and returned while the loop thread was already past its check.)
        self._cond = threading.Condition()
        self._want_run = True
        self._in_quantum = False
        self._stop = False
        self.quanta = 0
        self.quanta_by_version: dict[Any, int] = {}
        self.current_version: Any = 0
        self._thread: threading.Thread | None = None
        if generate_step is not None:
            self._thread = threading.Thread(target=self._loop, name="scripted-engine", daemon=True)
            self._thread.start()

    def _loop(self) -> None:
        while True:
            with self._cond:
                while not self._want_run and not self._stop:
                    self._cond.wait(0.05)
                if self._stop:
                    return
                self._in_quantum = True
            try:
                self.generate_step()
            finally:
                with self._cond:
                    self._in_quantum = False
                    self.quanta += 1
                    self.quanta_by_version[self.current_version] = self.quanta_by_version.get(self.current_version, 0) + 1
                    self._cond.notify_all()

    def generating(self) -> bool:
        """True while a generation quantum is executing (what a weight copy must never overlap)."""
        with self._cond:
            return self._in_quantum

    async def pause_generation(self, mode: str = "keep", clear_cache: bool = False) -> None:
        if mode != "keep":
            raise ValueError("the in-flight update pauses with mode='keep' (vllm1.py:164)")
        with self._cond:
            self._want_run = False
        while self.generating():
            await asyncio.sleep(0.0005)

    async def resume_generation(self) -> None:
        with self._cond:
            self._want_run = True
            self._cond.notify_all()

    async def collective_rpc(self, method: str, args: tuple = ()) -> list:
        loop = asyncio.get_running_loop()
        # every worker at once, like an engine's collective_rpc to its tensor-parallel worker processes: the TP ranks of a sharded update
        # each wait in a collective with the trainer, one after the other they would wait for each other
        out = list(await asyncio.gather(*[loop.run_in_executor(None, lambda w=w: getattr(w, method)(*args)) for w in self.workers]))
        if method == "receive_weight_update":
            try:
                self.current_version = json.loads(args[0]).get("version", self.current_version)
            except Exception:  # noqa: BLE001 - bookkeeping only
                pass
        return out

    def shutdown(self) -> None:
        with self._cond:
            self._stop = True
            self._cond.notify_all()
        if self._thread is not None:
            self._thread.join(timeout=5)


class UpdateServer:
    """The one route of the reference's API server that belongs to the hot path: `POST /receive_weight_update`, answered
    when the update is in place.  The manager's coroutines run on ONE event loop in a thread of their own; the HTTP
    threads hand requests over with `run_coroutine_threadsafe` (concurrent posts queue on the manager's lock)."""

    def __init__(self, manager: InflightUpdateManager, host: str = "127.0.0.1", port: int = 0):
        self.manager = manager
        self.loop = asyncio.new_event_loop()
        self._loop_thread = threading.Thread(target=self._run_loop, name="update-manager-loop", daemon=True)
        self._loop_thread.start()
        server = self

        class Handler(BaseHTTPRequestHandler):
            def log_message(self, *a):  # noqa: D401 - quiet
                pass

            def do_POST(self):  # noqa: N802 - http.server API
                if self.path.rstrip("/") != "/receive_weight_update":
                    self.send_error(404)
                    return
                body = self.rfile.read(int(self.headers.get("Content-Length", 0))).decode()
                try:
                    fut = asyncio.run_coroutine_threadsafe(server.manager.receive_weight_update(body), server.loop)
                    fut.result()
                    out, code = json.dumps({"status": "ok"}).encode(), 200
                except Exception as e:  # noqa: BLE001 - the trainer logs a failed POST and goes on (finetune_loop.py:188-192)
                    logger.exception("weight update failed")
                    out, code = json.dumps({"status": "error", "error": f"{type(e).__name__}: {e}"}).encode(), 500
                self.send_response(code)
                self.send_header("Content-Type", "application/json")
                self.send_header("Content-Length", str(len(out)))
                self.end_headers()
                self.wfile.write(out)

        self.httpd = ThreadingHTTPServer((host, port), Handler)
        self.host, self.port = self.httpd.server_address[:2]
        self._http_thread = threading.Thread(target=self.httpd.serve_forever, name="update-server", daemon=True)
        self._http_thread.start()

    def _run_loop(self) -> None:
        asyncio.set_event_loop(self.loop)
        # the lock must belong to the loop the coroutines run on
        self.manager.update_lock = asyncio.Lock()
        self.loop.run_forever()

    @property
    def url(self) -> str:
        return f"http://{self.host}:{self.port}"

    def close(self) -> None:
        self.httpd.shutdown()
        self.httpd.server_close()
        self.loop.call_soon_threadsafe(self.loop.stop)
        self._loop_thread.join(timeout=5)
