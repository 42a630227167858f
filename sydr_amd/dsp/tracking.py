"""Function-level drop-in for `EPL` of sydr/dsp/tracking.py:92-116 on the GPU.

The scalar discriminators / loop filters of that module (tracking.py:39-61,120-186,246-279) have no host copy
here: closed-loop tracking runs them on the device (sydr_amd/csrc/track.hip), and a reference plugin used
through the GpuCorrelatorSeams mixin keeps calling the reference's own."""
from __future__ import annotations

import numpy as np

from ..engine import FMT_CF64, make_items
from ..runtime import get_engine

_code_cache: dict[bytes, int] = {}


def EPL(rfData, code, samplingFrequency, carrierFrequency, remainingCarrier, remainingCode, codeStep,
        correlatorsSpacing):
    """[I_tap0, Q_tap0, I_tap1, ...] of tracking.py:92-116 for complex128 rfData; `code` is the padded
    replica [c[-1], c[0..L-1], c[0]] the reference plugins keep (channel_l1ca_kaplan.py:104-107)."""
    rf = np.squeeze(np.asarray(rfData, dtype=np.complex128))
    n = rf.size
    chips = np.asarray(code)[1:-1]
    eng = get_engine(0)
    if getattr(eng, "n_slots", 0) < 4:
        eng.code_slots(4, 4092)
    eng.set_code(2, chips.astype(np.int8))
    cap = (n + 15) // 8 * 8
    if eng.iq_fmt != FMT_CF64 or eng.iq_capacity < cap:
        eng.iq_alloc(cap, FMT_CF64)
    eng.iq_upload(rf, 0)
    items = make_items(2, n, 0, float(carrierFrequency), float(remainingCarrier), float(remainingCode),
                       float(codeStep))
    out = eng.epl_batch(items, tuple(float(s) for s in correlatorsSpacing), samplingFrequency)[0]
    return [float(v) for v in out]
