"""RF front-end description + IQ file reader with the reference's configuration contract
(sydr/signal/rfsignal.py:13-132; config/receiver.ini [RFSIGNAL]).  Unlike the reference it hands
out the file's native interleaved integers (2 B per ci8 sample) rather than complex128 (16 B):
that is the layout the device ring stores."""
from __future__ import annotations

import numpy as np


class RFSignal:
    CHUNCK_SIZE_MS = 120

    def __init__(self, configuration: dict):
        self.filepath = str(configuration['filepath'])
        self.samplingFrequency = float(configuration['sampling_frequency'])
        self.isComplex = bool(configuration['is_complex'])  # reference behaviour: any non-empty string is True
        self.interFrequency = float(configuration['intermediate_frequency'])
        dataSize = int(configuration['data_size'])
        if dataSize == 8:
            self.fileDataType = np.int8
        elif dataSize == 16:
            self.fileDataType = np.int16
        else:
            raise ValueError(f"Data type of {dataSize} bit(s) is not valid.")
        self.dtype = np.complex128 if self.isComplex else self.fileDataType
        self.file_id = None
        self.samplesPerMs = int(self.samplingFrequency * 1e-3)
        self.chunck = None
        self.chunckMsCounter = self.CHUNCK_SIZE_MS

    def getMilliseconds(self, nbMilliseconds: int, raw: bool = True):
        """Next block of samples: interleaved integers (raw=True, default) or complex128 like the reference."""
        if self.CHUNCK_SIZE_MS % nbMilliseconds:
            raise ValueError(f"The number of millisecond requested should be a multiple of the chunck size for "
                             f"optimal read ({nbMilliseconds} not multiple of {self.CHUNCK_SIZE_MS}).")
        if self.chunckMsCounter == self.CHUNCK_SIZE_MS:
            self.chunck = self.readFile(timeLength=self.CHUNCK_SIZE_MS, keep_open=True)
            self.chunckMsCounter = 0
        start = self.chunckMsCounter * self.samplesPerMs
        stop = start + self.samplesPerMs * nbMilliseconds
        self.chunckMsCounter += nbMilliseconds
        block = self.chunck[2 * start:2 * stop]
        if raw:
            return block
        return block[0::2] + 1j * block[1::2]

    def readFile(self, timeLength, skip=0, keep_open=False):
        """Interleaved I,Q integers for `timeLength` ms (complex files only, as the reference's data)."""
        count = int(2 * (timeLength * 1e-3) * self.samplingFrequency)
        offset = int(np.dtype(self.fileDataType).itemsize * skip * 2)
        fid = open(self.filepath, 'rb') if self.file_id is None else self.file_id
        data = np.fromfile(fid, self.fileDataType, offset=offset, count=count)
        if keep_open:
            self.file_id = fid
        else:
            fid.close()
        return data

    def readFileBySamples(self, nb_values, skip=0, keep_open=False):
        """Interleaved I,Q integers for `nb_values` samples, skipping `skip` samples first (rfsignal.py:138-181)."""
        count = int(2 * nb_values)
        offset = int(np.dtype(self.fileDataType).itemsize * skip * 2)
        fid = open(self.filepath, 'rb') if self.file_id is None else self.file_id
        data = np.fromfile(fid, self.fileDataType, offset=offset, count=count)
        if keep_open:
            self.file_id = fid
        else:
            fid.close()
        return data

    def closeFile(self):
        if self.file_id is None:
            raise Warning("File was already close.")
        self.file_id.close()
        self.file_id = None

    def getCurrentSampleIndex(self):
        """Index of the next sample the open file will deliver (rfsignal.py:195-203: byte position / 2 for I,Q files of
        one byte per component, as the reference computes it)."""
        if self.file_id is None:
            raise Warning("Signal file not open, cannot return current cursor position.")
        return int(self.file_id.tell() / 2)
