"""A small event-driven loop engine with the slice of pytorch-ignite's ``Engine`` API that the
reference trainer and its handlers use (experiments/trainer.py, experiments/options.py of the
reference).  ignite is not installed in the target image; when it is importable the trainer uses
the real thing instead.

Supported: ``Engine(process_function)``, ``state`` (iteration / epoch / epoch_length / max_epochs /
output / batch / metrics / dataloader + registered custom counters), ``register_events``,
``add_event_handler`` (handlers with or without the leading ``engine`` argument), ``fire_event``,
``run(data, max_epochs, epoch_length)`` (re-runnable), ``interrupt`` / ``terminate`` /
``terminate_epoch``, ``Events`` with ``|`` unions and ``(every=N)`` / ``(once=N)`` /
``(event_filter=f)`` filters, ``EventEnum`` for custom events, ``state_dict`` /
``load_state_dict`` with ``state_dict_user_keys``.
"""
from __future__ import annotations

import inspect
from enum import Enum
from typing import Any, Callable, Iterable, Optional


class _Filtered:
    """An event plus a predicate on its counter value."""

    def __init__(self, event, predicate: Callable[["Engine", int], bool]) -> None:
        self.event, self.predicate = event, predicate

    def __or__(self, other):
        return EventsList([self]) | other


class EventsList:
    def __init__(self, events: Iterable) -> None:
        self.events = list(events)

    def __or__(self, other):
        more = other.events if isinstance(other, EventsList) else [other]
        return EventsList(self.events + more)

    def __iter__(self):
        return iter(self.events)


class EventEnum(Enum):
    """Base of event enums; members combine with ``|`` and can be called to add a filter."""

    def __or__(self, other):
        return EventsList([self]) | other

    def __call__(self, every: Optional[int] = None, once: Optional[int] = None,
                 event_filter: Optional[Callable] = None):
        given = [x is not None for x in (every, once, event_filter)]
        if sum(given) != 1:
            raise ValueError("exactly one of every / once / event_filter is required")
        if every is not None:
            if every < 1:
                raise ValueError("every must be >= 1")
            return _Filtered(self, lambda _e, count: count % every == 0)
        if once is not None:
            return _Filtered(self, lambda _e, count: count == once)
        return _Filtered(self, event_filter)


class Events(EventEnum):
    STARTED = "started"
    COMPLETED = "completed"
    EPOCH_STARTED = "epoch_started"
    EPOCH_COMPLETED = "epoch_completed"
    ITERATION_STARTED = "iteration_started"
    ITERATION_COMPLETED = "iteration_completed"
    GET_BATCH_STARTED = "get_batch_started"
    GET_BATCH_COMPLETED = "get_batch_completed"
    DATALOADER_STOP_ITERATION = "dataloader_stop_iteration"
    EXCEPTION_RAISED = "exception_raised"
    TERMINATE = "terminate"
    TERMINATE_SINGLE_EPOCH = "terminate_single_epoch"
    INTERRUPT = "interrupt"


_COUNTER_OF = {
    Events.STARTED: "epoch", Events.COMPLETED: "epoch", Events.EPOCH_STARTED: "epoch",
    Events.EPOCH_COMPLETED: "epoch", Events.TERMINATE: "epoch", Events.INTERRUPT: "iteration",
    Events.ITERATION_STARTED: "iteration", Events.ITERATION_COMPLETED: "iteration",
    Events.GET_BATCH_STARTED: "iteration", Events.GET_BATCH_COMPLETED: "iteration",
    Events.DATALOADER_STOP_ITERATION: "iteration", Events.EXCEPTION_RAISED: "iteration",
    Events.TERMINATE_SINGLE_EPOCH: "iteration",
}


class State:
    def __init__(self) -> None:
        self.iteration = 0
        self.epoch = 0
        self.epoch_length: Optional[int] = None
        self.max_epochs: Optional[int] = None
        self.output: Any = None
        self.batch: Any = None
        self.metrics: dict = {}
        self.dataloader: Any = None
        self.seed: Optional[int] = None
        self.times: dict = {}

    def __repr__(self) -> str:
        keys = ("iteration", "epoch", "epoch_length", "max_epochs")
        return "State(" + ", ".join(f"{k}={getattr(self, k)}" for k in keys) + ")"


class Engine:
    def __init__(self, process_function: Callable[["Engine", Any], Any]) -> None:
        self._process = process_function
        self.state = State()
        self._handlers: dict = {}
        self._counter_of = dict(_COUNTER_OF)
        self._allowed = set(Events)
        self.should_terminate = False
        self.should_terminate_single_epoch = False
        self.should_interrupt = False
        self.state_dict_user_keys: list[str] = []

    # ---- events -------------------------------------------------------------------------------
    def register_events(self, *events, event_to_attr: Optional[dict] = None) -> None:
        for ev in events:
            self._allowed.add(ev)
            attr = (event_to_attr or {}).get(ev)
            if attr is not None:
                self._counter_of[ev] = attr
                if not hasattr(self.state, attr):
                    setattr(self.state, attr, 0)

    def add_event_handler(self, event_name, handler: Callable, *args, **kwargs) -> None:
        if isinstance(event_name, EventsList):
            for ev in event_name:
                self.add_event_handler(ev, handler, *args, **kwargs)
            return
        predicate = None
        if isinstance(event_name, _Filtered):
            event_name, predicate = event_name.event, event_name.predicate
        if event_name not in self._allowed:
            raise ValueError(f"Event {event_name} is not registered on this engine")
        self._handlers.setdefault(event_name, []).append(
            (handler, args, kwargs, predicate, self._wants_engine(handler, args, kwargs)))

    def on(self, event_name, *args, **kwargs):
        def decorator(fn):
            self.add_event_handler(event_name, fn, *args, **kwargs)
            return fn

        return decorator

    def has_event_handler(self, handler: Callable, event_name=None) -> bool:
        events = [event_name] if event_name is not None else list(self._handlers)
        return any(h[0] == handler for ev in events for h in self._handlers.get(ev, []))

    def _wants_engine(self, handler, args, kwargs) -> bool:
        try:
            inspect.signature(handler).bind(self, *args, **kwargs)
            return True
        except TypeError:
            inspect.signature(handler).bind(*args, **kwargs)  # raises if neither form fits
            return False
        except ValueError:  # builtins without a signature
            return True

    def fire_event(self, event_name) -> None:
        self._fire(event_name)

    def _fire(self, event_name, *event_args) -> None:
        count = getattr(self.state, self._counter_of.get(event_name, "iteration"), 0)
        for handler, args, kwargs, predicate, wants_engine in list(self._handlers.get(event_name, [])):
            if predicate is not None and not predicate(self, count):
                continue
            first = (self,) if wants_engine else ()
            handler(*first, *event_args, *args, **kwargs)

    # ---- control ------------------------------------------------------------------------------
    def terminate(self) -> None:
        self.should_terminate = True

    def terminate_epoch(self) -> None:
        self.should_terminate_single_epoch = True

    def interrupt(self) -> None:
        self.should_interrupt = True

    # ---- checkpointing ------------------------------------------------------------------------
    def state_dict(self) -> dict:
        keys = ["epoch_length", "max_epochs", "iteration"] + list(self.state_dict_user_keys)
        return {k: getattr(self.state, k, None) for k in keys}

    def load_state_dict(self, state_dict: dict) -> None:
        for k, v in state_dict.items():
            setattr(self.state, k, v)
        if self.state.epoch_length:
            self.state.epoch = self.state.iteration // self.state.epoch_length

    # ---- the loop -----------------------------------------------------------------------------
    def run(self, data: Optional[Iterable] = None, max_epochs: Optional[int] = None,
            epoch_length: Optional[int] = None) -> State:
        st = self.state
        finished = st.max_epochs is None or st.epoch >= st.max_epochs
        if finished and not self.should_interrupt:  # fresh run (engines are re-runnable)
            st.iteration = st.epoch = 0
            st.max_epochs = max_epochs if max_epochs is not None else 1
            st.epoch_length = epoch_length
        elif max_epochs is not None:
            st.max_epochs = max_epochs
        if data is not None:
            st.dataloader = data
        if st.epoch_length is None:
            st.epoch_length = epoch_length if epoch_length is not None else len(st.dataloader)
        self.should_terminate = self.should_terminate_single_epoch = self.should_interrupt = False
        try:
            if st.iteration == 0:
                self._fire(Events.STARTED)
            while st.epoch < st.max_epochs and not self.should_terminate:
                st.epoch += 1
                self._fire(Events.EPOCH_STARTED)
                self._run_epoch()
                if self.should_interrupt:
                    st.epoch -= 1
                    self._fire(Events.INTERRUPT)
                    return st
                if self.should_terminate:
                    break
                self._fire(Events.EPOCH_COMPLETED)
            if self.should_terminate:
                self._fire(Events.TERMINATE)
            self._fire(Events.COMPLETED)
        except BaseException as exc:  # noqa: BLE001 - handlers decide
            if self._handlers.get(Events.EXCEPTION_RAISED):
                self._fire(Events.EXCEPTION_RAISED, exc)
            else:
                raise
        return st

    def _run_epoch(self) -> None:
        st = self.state
        it = iter(st.dataloader)
        done_in_epoch = st.iteration - (st.epoch - 1) * st.epoch_length
        while done_in_epoch < st.epoch_length:
            self._fire(Events.GET_BATCH_STARTED)
            try:
                st.batch = next(it)
            except StopIteration:
                self._fire(Events.DATALOADER_STOP_ITERATION)
                it = iter(st.dataloader)
                st.batch = next(it)
            self._fire(Events.GET_BATCH_COMPLETED)
            st.iteration += 1
            done_in_epoch += 1
            self._fire(Events.ITERATION_STARTED)
            st.output = self._process(self, st.batch)
            self._fire(Events.ITERATION_COMPLETED)
            if self.should_terminate or self.should_interrupt:
                return
            if self.should_terminate_single_epoch:
                self._fire(Events.TERMINATE_SINGLE_EPOCH)
                self.should_terminate_single_epoch = False
                st.iteration = st.epoch * st.epoch_length
                return
