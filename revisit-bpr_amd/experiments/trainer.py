"""Two-engine (train / eval) trainer behind the API of the reference's experiments/trainer.py
(:12-143): ``ModelEvents``, ``Trainer(model, optimizer, accelerator, custom_engines)``,
``.engines``, ``.add_event(engine, event, handler, ...)``, ``.run(loaders, max_iters, epochs)``.

The step is the reference's: FORWARD_STARTED → model(batch) → FORWARD_COMPLETED →
accelerator.backward(loss) → OPTIMIZER_STARTED → optimizer.step() → OPTIMIZER_COMPLETED →
zero_grad, with eval before every training epoch and once more at the end.  With the fused BPR
model those calls drive the HIP engine (revisit_bpr/models/bpr/model.py); nothing here is
BPR-specific.  Uses pytorch-ignite when importable, else the bundled ``engine_lite``.
"""
from __future__ import annotations

from contextlib import nullcontext
from copy import deepcopy
from typing import Any, Callable, Optional

import torch

try:  # pragma: no cover - ignite is absent from the target image
    from ignite.engine import Engine, EventEnum, Events, State
except ImportError:
    from experiments.engine_lite import Engine, EventEnum, Events, State


class ModelEvents(EventEnum):
    FORWARD_STARTED = "forward_started"
    FORWARD_COMPLETED = "forward_completed"
    OPTIMIZER_STARTED = "optimizer_started"
    OPTIMIZER_COMPLETED = "optimizer_completed"


_COUNTERS = {
    ModelEvents.FORWARD_STARTED: "forward_iteration",
    ModelEvents.FORWARD_COMPLETED: "forward_iteration",
    ModelEvents.OPTIMIZER_STARTED: "optimizer_iteration",
    ModelEvents.OPTIMIZER_COMPLETED: "optimizer_iteration",
}
_CHECKPOINTED = ("name", "forward_iteration", "optimizer_iteration", "epoch_iteration",
                 "was_interrupted")


class NullAccelerator:
    """Stand-in for accelerate.Accelerator on a single device (device / backward / accumulate)."""

    def __init__(self, device: Optional[torch.device] = None) -> None:
        self.device = device or torch.device("cuda" if torch.cuda.is_available() else "cpu")

    def backward(self, loss: torch.Tensor) -> None:
        loss.backward()

    def accumulate(self, *_models):
        return nullcontext()

    def prepare(self, *objs):
        return objs if len(objs) != 1 else objs[0]


class Trainer:
    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, accelerator: Any,
                 custom_engines: Optional[dict[str, str]] = None) -> None:
        self.model = model
        self.optimizer = optimizer
        self._accelerator = accelerator
        self.engines = {"train": Engine(self._train_step), "eval": Engine(self._eval_step)}
        for name, source in (custom_engines or {}).items():
            self.engines[name] = deepcopy(self.engines[source])
        self._loaders: dict = {}
        self._max_iters: dict = {}
        for engine in self.engines.values():
            engine.register_events(*ModelEvents, event_to_attr=_COUNTERS)
        self.add_event("train", Events.EPOCH_STARTED | Events.COMPLETED, self._run_eval)
        for name, engine in self.engines.items():
            self.add_event(name, Events.EPOCH_STARTED, self._reset_epoch)
            self.add_event(name, Events.ITERATION_COMPLETED, self._count_iteration)
            self.add_event(name, Events.ITERATION_COMPLETED, self._mean_loss)
            engine.state.name = name
            engine.state.was_interrupted = False
            engine.state.epoch_iteration = 0
            engine.state_dict_user_keys.extend(_CHECKPOINTED)

    def add_event(self, engine: str, event_name: Any, handler: Callable, *args: Any,
                  **kwargs: Any) -> None:
        self.engines[engine].add_event_handler(event_name, handler, *args, **kwargs)

    def run(self, loaders: dict, max_iters: Optional[dict[str, int]] = None,
            epochs: Optional[int] = None) -> State:
        self._loaders = loaders
        self._max_iters = max_iters or {}
        self.engines["train"].run(loaders["train"], epoch_length=self._max_iters.get("train"),
                                  max_epochs=epochs)
        return self.engines["eval" if "eval" in loaders else "train"].state

    # ---- steps --------------------------------------------------------------------------------
    def _forward(self, engine: Engine, batch: dict) -> dict:
        """model(batch) between the two FORWARD events; the running loss sum feeds `_mean_loss`."""
        st = engine.state
        st.forward_iteration += 1
        engine.fire_event(ModelEvents.FORWARD_STARTED)
        st.output = result = self.model(batch)
        engine.fire_event(ModelEvents.FORWARD_COMPLETED)
        return result

    def _train_step(self, engine: Engine, batch: dict) -> dict:
        self.model.train()
        with self._accelerator.accumulate(self.model):
            result = self._forward(engine, batch)
            loss = result.get("loss")
            if loss is not None:
                self._accelerator.backward(loss)
                engine.state.optimizer_iteration += 1
                engine.fire_event(ModelEvents.OPTIMIZER_STARTED)
                self.optimizer.step()
                engine.fire_event(ModelEvents.OPTIMIZER_COMPLETED)
                self.optimizer.zero_grad()
                engine.state.metrics["_loss"] += loss.detach()
        return result

    @torch.no_grad()
    def _eval_step(self, engine: Engine, batch: dict) -> dict:
        self.model.eval()
        result = self._forward(engine, batch)
        if "loss" in result:
            engine.state.metrics["_loss"] += result["loss"].detach()
        return result

    # ---- built-in handlers ----------------------------------------------------------------------
    def _run_eval(self) -> None:
        train, evl = self.engines["train"].state, self.engines["eval"].state
        if train.was_interrupted and not evl.was_interrupted:
            return  # resume: the interrupted epoch's eval already ran
        loader = self._loaders.get("eval")
        if loader is not None:
            self.engines["eval"].run(loader, epoch_length=self._max_iters.get("eval"))

    def _reset_epoch(self, engine: Engine) -> None:
        if engine.state.was_interrupted:
            return
        engine.state.metrics["_loss"] = torch.tensor(0.0, device=self._accelerator.device)
        engine.state.epoch_iteration = 0

    @staticmethod
    def _count_iteration(engine: Engine) -> None:
        engine.state.epoch_iteration += 1

    @staticmethod
    def _mean_loss(engine: Engine) -> None:
        st = engine.state
        st.metrics["loss"] = st.metrics["_loss"] / st.epoch_iteration
