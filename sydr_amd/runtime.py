"""Process-wide engines: one per GPU, created on first use (never on a CPU-only host).

Which GPU the function-level drop-ins (`sydr_amd.dsp.*`, a `CircularBuffer` made without an engine, a `ChannelManager`
made without a device) use when nobody says: `set_default_device(n)`, else the environment's SYDR_AMD_DEVICE, else 0.
A job of one process per GPU sets it to its LOCAL_RANK once; a manager over several devices names them itself."""
from __future__ import annotations

import os

from .engine import Engine

_engines: dict[int, Engine] = {}
_default_device: int | None = None


def default_device() -> int:
    if _default_device is not None:
        return _default_device
    return int(os.environ.get("SYDR_AMD_DEVICE", "0"))


def set_default_device(device_id: int) -> None:
    global _default_device
    _default_device = int(device_id)


def get_engine(device_id: int | None = None) -> Engine:
    device_id = default_device() if device_id is None else int(device_id)
    eng = _engines.get(device_id)
    if eng is None:
        eng = Engine(device_id)  # raises SdrError when no MI355X is visible: there is no CPU path
        _engines[device_id] = eng
    return eng


def close_all():
    for eng in _engines.values():
        eng.close()
    _engines.clear()
