"""Enumerations of the plugin surface (same names and values as sydr/utils/enumerations.py:59-147),
so packets produced here are interchangeable with the reference's."""
from enum import Enum, IntEnum, unique


@unique
class GNSSSystems(Enum):
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
class GNSSSignalType(Enum):
    GPS_L1_CA = 0

    def __str__(self):
        return str(self.name).replace("_", " ")


@unique
class ChannelState(Enum):
    OFF = 0
    IDLE = 1
    ACQUIRING = 2
    TRACKING = 3

    def __str__(self):
        return str(self.name)


@unique
class ChannelMessage(Enum):
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
