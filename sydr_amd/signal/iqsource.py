"""IQ recording -> slabs for the device ring.

Reads the [RFSIGNAL] section of the reference's receiver.ini (keys `filepath`, `sampling_frequency`,
`is_complex`, `intermediate_frequency`, `data_size`; sydr/signal/rfsignal.py:13-54) and serves the file the way
the GPU wants it: the recording is memory-mapped and a slab is a zero-copy view of its native interleaved
integers (int8 / int16 I,Q -- 2 or 4 bytes per sample), which is byte for byte what the device ring stores.
The reference instead reads 120 ms chunks and inflates every sample to complex128 (16 bytes) before anything
else touches it (rfsignal.py:58-132).  Only what the hot path's callers use is kept of that class's surface:
the front-end attributes, `getMilliseconds`, and `readFile` / `readFileBySamples` / `closeFile` /
`getCurrentSampleIndex` (rfsignal.py:92-204) as views of the mapped file with the reference's cursor semantics.
"""
from __future__ import annotations

import os

import numpy as np

class RFSignal:
    def __init__(self, configuration):
        self.filepath = str(configuration["filepath"])
        self.samplingFrequency = float(configuration["sampling_frequency"])
        self.interFrequency = float(configuration["intermediate_frequency"])
        # rfsignal.py:35: bool(<ini string>) -- ANY non-empty string is True there ("false" included); mirrored as is
        self.isComplex = bool(configuration["is_complex"])
        bits = int(configuration["data_size"])
        if bits not in (8, 16):
            raise ValueError(f"Data type of {bits} bit(s) is not valid.")
        self.fileDataType = np.int8 if bits == 8 else np.int16
        if not self.isComplex:
            raise ValueError("real-valued recordings are not supported: the correlators take I,Q samples")
        self.dtype = np.complex128                      # what a sample IS (the reference's rfSignal.dtype); storage stays integer
        self.samplesPerMs = int(self.samplingFrequency * 1e-3)
        self._map = None
        self._next = 0                                  # samples handed out so far
        self._open = False                              # the reference's `file_id is not None` (readFile keep_open)

    # ------------------------------------------------------------------ the recording
    def _recording(self) -> np.ndarray:
        if self._map is None:
            if not os.path.isfile(self.filepath):
                raise FileNotFoundError(f"IQ recording {self.filepath!r} does not exist")
            self._mapping = np.memmap(self.filepath, dtype=self.fileDataType, mode="r")
            # (a plain ndarray over the mapping: slicing an np.memmap costs ~10 us per slab in subclass bookkeeping)
            self._map = np.asarray(self._mapping) if self._mapping.size else self._mapping
        return self._map

    @property
    def totalSamples(self) -> int:
        return self._recording().size // 2

    @property
    def position(self) -> int:
        """Index of the next sample `getMilliseconds` will deliver."""
        return self._next

    def seek(self, sample: int):
        if not 0 <= sample <= self.totalSamples:
            raise ValueError(f"sample {sample} outside the recording's {self.totalSamples}")
        self._next = int(sample)

    def slab(self, n_samples: int) -> np.ndarray:
        """The next n_samples as interleaved integers [I0, Q0, I1, Q1, ...] -- a view of the mapped file."""
        rec = self._recording()
        stop = self._next + int(n_samples)
        if stop > rec.size // 2:
            raise EOFError(f"recording ends at sample {rec.size // 2}, {stop} requested")
        out = rec[2 * self._next:2 * stop]
        self._next = stop
        return out

    # ------------------------------------------------------------------ the reference's call (receiver.py:124)
    def getMilliseconds(self, nbMilliseconds: int = 1, raw: bool = True):
        """Next `nbMilliseconds` of signal.  raw=True (default): interleaved integers for the device ring;
        raw=False: complex128 like the reference's RFSignal.getMilliseconds (rfsignal.py:58-88)."""
        block = self.slab(self.samplesPerMs * int(nbMilliseconds))
        if raw:
            return block
        return block[0::2].astype(np.float64) + 1j * block[1::2].astype(np.float64)

    # ------------------------------------------------------------------ the reference's file readers (rfsignal.py:92-204)
    def _read(self, n_samples: int, skip: int, keep_open: bool, raw: bool):
        """`skip` counts from the cursor while the file is "open" (np.fromfile(fid, offset=...) on a kept descriptor),
        from the start of the recording otherwise; a short read at the end of the file returns what is there."""
        rec = self._recording()
        first = (self._next if self._open else 0) + int(skip)
        stop = min(first + int(n_samples), rec.size // 2)
        block = rec[2 * first:2 * max(stop, first)]
        if keep_open:
            self._open, self._next = True, max(stop, first)
        elif self._open:              # the kept descriptor is closed by a read without keep_open (rfsignal.py:118-121)
            self._open = False
        if raw:
            return block
        return block[0::2].astype(np.float64) + 1j * block[1::2].astype(np.float64)

    def readFile(self, timeLength, skip=0, keep_open=False, raw=False):
        """`timeLength` milliseconds of signal (complex128 like the reference; raw=True: the interleaved integers)."""
        return self._read(int((timeLength * 1e-3) * self.samplingFrequency), skip, keep_open, raw)

    def readFileBySamples(self, nb_values, skip=0, keep_open=False, raw=False):
        return self._read(int(nb_values), skip, keep_open, raw)

    def closeFile(self):
        if not self._open:
            raise Warning("File was already close.")
        self._open = False

    def getCurrentSampleIndex(self):
        if not self._open:
            raise Warning("Signal file not open, cannot return current cursor position.")
        return int(self._next)
