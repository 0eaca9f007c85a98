"""Config loading without hydra: jinja2 render → YAML → a minimal ``instantiate``.

The reference's configs (configs/**/*.yaml.j2) use three hydra keys only — ``_target_`` (1126
occurrences), ``_partial_`` (26), ``_convert_`` (4) — so this covers them:
  _target_:  dotted path of a callable, called with the node's other keys (instantiated recursively)
  _partial_: return functools.partial(target, **kwargs) instead of calling it
  _convert_: accepted and ignored (nodes are already plain dict / list)
"""
from __future__ import annotations

import functools
import importlib
import re
from pathlib import Path
from typing import Any, Mapping

import yaml


def parse_extra_vars(text: str | None) -> dict[str, str]:
    """"a=1;b=x/y" → {"a": "1", "b": "x/y"} (the reference's --extra-vars syntax)."""
    out: dict[str, str] = {}
    for part in re.split(r"[;\n]", text or ""):
        if part.strip():
            key, _, value = part.partition("=")
            out[key.strip()] = value.strip()
    return out


def render(path: str | Path, variables: Mapping[str, Any]) -> dict:
    """Render a .yaml.j2 template with StrictUndefined (a missing variable is an error) and parse it."""
    import jinja2

    env = jinja2.Environment(undefined=jinja2.StrictUndefined)
    text = env.from_string(Path(path).read_text()).render(**variables)
    return yaml.safe_load(text)


def locate(dotted: str):
    module, _, attr = dotted.rpartition(".")
    obj: Any = None
    parts = dotted.split(".")
    for cut in range(len(parts) - 1, 0, -1):  # longest importable module prefix
        try:
            obj = importlib.import_module(".".join(parts[:cut]))
        except ImportError:
            continue
        for name in parts[cut:]:
            obj = getattr(obj, name)
        return obj
    raise ImportError(f"cannot locate {dotted!r}")


def instantiate(node: Any, **overrides: Any) -> Any:
    """Recursively build the object described by `node`; `overrides` are extra keyword arguments
    for the top-level target."""
    if isinstance(node, (list, tuple)):
        return [instantiate(x) for x in node]
    if not isinstance(node, Mapping):
        return node
    if "_target_" not in node:
        return {k: instantiate(v) for k, v in node.items()}
    target = locate(node["_target_"])
    kwargs = {k: instantiate(v) for k, v in node.items()
              if k not in ("_target_", "_partial_", "_convert_", "_recursive_")}
    kwargs.update(overrides)
    if node.get("_partial_", False):
        return functools.partial(target, **kwargs)
    return target(**kwargs)
