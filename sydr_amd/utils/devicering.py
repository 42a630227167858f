"""The RF ring buffer, with the reference's bookkeeping (sydr/utils/circularbuffer.py:21-148) and its
storage in HBM (the engine's IQ ring) instead of host shared memory.

Kept: maxSize / idxWrite / size / full, shift(), shiftIdxWrite(), getSlice(), getNbUnreadSamples()
with identical index arithmetic.  Changed: samples live on the GPU in the file's native format
(interleaved int8 / int16; complex128 only when asked for), so shift() is one H2D copy and the
correlator kernels read the ring in place (modulo addressing)."""
from __future__ import annotations

import numpy as np

from ..engine import FMT_CF64, FMT_CI16, FMT_CI8, Engine

_FMT_OF_DTYPE = {np.dtype(np.int8): FMT_CI8, np.dtype(np.int16): FMT_CI16, np.dtype(np.complex128): FMT_CF64,
                 np.dtype(np.float64): FMT_CF64}


class CircularBuffer:
    def __init__(self, size: int, dtype=complex, sharedMemory=None, engine: Engine | None = None, fmt=None):
        if size % 8:
            raise ValueError("device ring size must be a multiple of 8 samples")
        if engine is None:
            from ..runtime import get_engine
            engine = get_engine()
        self.engine = engine
        self.maxSize = int(size)
        self.dtype = dtype
        self.sharedMemory = sharedMemory  # accepted for signature compatibility; unused (no shm, no fork)
        self.fmt = fmt if fmt is not None else _FMT_OF_DTYPE.get(np.dtype(dtype), FMT_CF64)
        engine.iq_alloc(self.maxSize, self.fmt)
        # element type of the interleaved I,Q the ring stores (what a slab needs no conversion from)
        self.rawDtype = np.dtype({FMT_CI8: np.int8, FMT_CI16: np.int16}.get(self.fmt, np.float64 if self.fmt == FMT_CF64 else np.float32))
        self.full = False
        self.idxWrite = 0
        self.idxRead = 0
        self.size = 0
        self.channelBank = None   # the tracking state of this ring's channels, resident on the same GPU

    def bankFor(self, cid: int):
        """The device-resident channel bank that goes with this ring, grown to hold channel `cid`."""
        from ..channel.bank import ChannelBank
        if self.channelBank is None:
            self.channelBank = ChannelBank(self.engine, max(32, int(cid) + 1), self)
        elif self.channelBank.max_channels <= int(cid):
            self.channelBank = self.channelBank.grown(max(2 * self.channelBank.max_channels, int(cid) + 1))
        return self.channelBank

    # ---------------------------------------------------------------- writes (circularbuffer.py:54-108)
    def stage(self, data):
        """What shift() would upload and where: (samples in the ring's format, ring offset, sample count).  The
        caller uploads them (ChannelManager fuses the copy into the tick's device call) and then advances the
        write index with shiftIdxWrite(count)."""
        data = np.asarray(data)
        if np.iscomplexobj(data):
            shift = data.size
            if self.fmt != FMT_CF64:
                raw = np.empty(2 * shift, dtype=np.int8 if self.fmt == FMT_CI8 else np.int16)
                re, im = np.rint(data.real.reshape(-1)), np.rint(data.imag.reshape(-1))
                if not (np.array_equal(re, data.real.reshape(-1)) and np.array_equal(im, data.imag.reshape(-1))):
                    raise ValueError("non-integer samples cannot enter an integer ring; allocate it with dtype=complex")
                raw[0::2], raw[1::2] = re, im
                data = raw
        else:
            if data.size % 2:
                raise ValueError("interleaved I,Q data needs an even number of elements")
            shift = data.size // 2
        if self.maxSize % shift != 0:
            raise ValueError("Data shift need to be a multiple from the max buffer size.")
        return data, self.idxWrite, shift

    def shift(self, data):
        """Append one block.  `data` is complex (one value per sample) or raw interleaved I,Q integers."""
        data, offset, count = self.stage(data)
        self.engine.iq_upload(data, offset)
        self.shiftIdxWrite(count)

    def shiftIdxWrite(self, shift: int):
        self.idxWrite += shift
        self.size = self.idxWrite
        if self.full:
            self.idxWrite %= self.maxSize
        else:
            if self.idxWrite >= self.maxSize:
                self.full = True
                self.idxWrite %= self.maxSize
            if self.size > self.maxSize:
                self.size = self.maxSize

    # ---------------------------------------------------------------- reads (circularbuffer.py:114-148)
    def getSlice(self, idxStart: int = None, samplesRequired: int = 0):
        """Host copy of a slice as complex128 (1, n) -- diagnostics / compatibility only; the kernels
        never need it."""
        if idxStart is None:
            idxStart = self.idxRead
        raw = self.engine.iq_download(int(samplesRequired), int(idxStart) % self.maxSize)
        raw = raw.astype(np.float64)
        return (raw[0::2] + 1j * raw[1::2]).reshape(1, -1)

    def getNbUnreadSamples(self, currentSample: int):
        if currentSample <= self.idxWrite:
            return self.idxWrite - currentSample
        return self.maxSize - currentSample + self.idxWrite
