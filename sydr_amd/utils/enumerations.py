"""Enumerations of the plugin surface (same names and values as sydr/utils/enumerations.py:59-147),
so packets produced here are interchangeable with the reference's.

`ChannelMessage`, `ChannelState`, `GNSSSystems` and `GNSSSignalType` are plain `Enum`s in the reference: its
receiver compares `packet['type'] == ChannelMessage.DECODING_UPDATE` (receiver_gps_l1ca.py:113-131) against ITS
classes, and its database sink adapts them through `__conform__`.  A look-alike class would compare unequal, so
when the reference package is importable on this host (the intended deployment keeps `sydr/receiver` untouched, so
it is) the reference's own classes are re-exported at the bottom of this module; the definitions below serve a
host without it."""
import importlib
import sqlite3
from enum import Enum, IntEnum, unique


class _DbText:
    """sqlite3 adaptation of the reference's enums (enumerations.py:28-35): stored by name."""

    def __conform__(self, protocol):
        if protocol is sqlite3.PrepareProtocol:
            return str(self.name)


@unique
class GNSSSystems(_DbText, Enum):
    UNKNOWN = 0
    GPS = 1
    GLONASS = 2
    GALILEO = 3
    BEIDOU = 4
    QZSS = 5
    IRNSS = 6
    SBAS = 7

    def __str__(self):
        return str(self.name)


@unique
class GNSSSignalType(_DbText, Enum):
    GPS_L1_CA = 0

    def __str__(self):
        return str(self.name).replace("_", " ")


@unique
class ChannelState(_DbText, Enum):
    OFF = 0
    IDLE = 1
    ACQUIRING = 2
    TRACKING = 3

    def __str__(self):
        return str(self.name)


@unique
class ChannelMessage(_DbText, Enum):
    END_OF_PIPE = 0
    CHANNEL_UPDATE = 1
    ACQUISITION_UPDATE = 2
    TRACKING_UPDATE = 3
    DECODING_UPDATE = 4

    def __str__(self):
        return str(self.name)


@unique
class TrackingFlags(IntEnum):
    UNKNOWN = 0
    CODE_LOCK = 1
    BIT_SYNC = 2
    SUBFRAME_SYNC = 4
    TOW_DECODED = 8
    EPH_DECODED = 16
    TOW_KNOWN = 32
    EPH_KNOWN = 64
    FINE_LOCK = 128

    def __str__(self):
        return str(self.name)


@unique
class LoopLockState(IntEnum):
    UNKNOWN = 0
    PULL_IN = 1
    WIDE_TRACK = 2
    NARROW_TRACK = 3

    def __str__(self):
        return str(self.name)


def _adopt_reference_enums():
    """Re-export the reference's classes when `sydr` is on this host's path (never on the GPU-only box)."""
    try:
        ref = importlib.import_module("sydr.utils.enumerations")
    except Exception:          # not installed (or a broken install): this module's own definitions stand
        return False
    for name in ("GNSSSystems", "GNSSSignalType", "ChannelState", "ChannelMessage", "TrackingFlags", "LoopLockState"):
        theirs, ours = getattr(ref, name, None), globals()[name]
        if theirs is None or {m.name: m.value for m in theirs} != {m.name: m.value for m in ours}:
            return False       # a reference of another vintage: keep the two worlds apart rather than half-mixed
    for name in ("GNSSSystems", "GNSSSignalType", "ChannelState", "ChannelMessage", "TrackingFlags", "LoopLockState"):
        globals()[name] = getattr(ref, name)
    return True


USING_REFERENCE_ENUMS = _adopt_reference_enums()
