"""Process-wide engines: one per GPU, created on first use (never on a CPU-only host)."""
from __future__ import annotations

from .engine import Engine

_engines: dict[int, Engine] = {}


def get_engine(device_id: int = 0) -> Engine:
    eng = _engines.get(device_id)
    if eng is None:
        eng = Engine(device_id)  # raises SdrError when no MI355X is visible: there is no CPU path
        _engines[device_id] = eng
    return eng


def close_all():
    for eng in _engines.values():
        eng.close()
    _engines.clear()
