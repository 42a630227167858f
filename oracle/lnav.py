"""GPS LNAV subframe ENCODER, a word-level checker and a restatement of the reference plugins' subframe
synchroniser -- TEST INFRASTRUCTURE ONLY (part of the oracle: nothing under sydr_amd/ may import it).

The reference decodes LNAV (sydr/dsp/decoding.py) but has no encoder and ships no recorded subframes; to
obtain the reference's own DECODING_UPDATE packets as fixtures (tests/golden/make_golden.py `decoding`) the
synthetic stream has to carry valid subframes.  This module builds them from the published frame layout
(IS-GPS-200, 20.3.2-20.3.5: TLM word with the 10001011 preamble, HOW with the 17-bit TOW count and the
subframe ID, the (32,26) Hamming parity of Table 20-XIV, the two non-information bits of words 2 and 10
solved so that D29 = D30 = 0).  The reference's decoder accepting these frames is what validates it.

`WordChecker` re-derives preamble / parity / TOW from the same table, and `SubframeSync` restates the state
machine the reference's plugins run on the bit stream (channel_l1ca_kaplan.py:756-868, channel_l1ca_borre.py:
493-579).  Together they are the stub "other decoder" of the decoder seam's protocol
(sydr_amd/channel/navdecoder.py) and exist so that the seam can be exercised where the reference cannot be
imported (the GPU box).  Parity: PINNED by tests/golden/g10_decoding.npz -- the reference's own DECODING_UPDATE
packets, flags and `tow` on a 20.6 s stream (tests/test_oracle_golden.py).
"""
from __future__ import annotations

import numpy as np

PREAMBLE = (1, 0, 0, 0, 1, 0, 1, 1)

# Table 20-XIV: data bits d1..d24 (1-based) entering D25..D30, and which of D29*/D30* each one starts from
_PARITY_ROWS = (
    (29, (1, 2, 3, 5, 6, 10, 11, 12, 13, 14, 17, 18, 20, 23)),
    (30, (2, 3, 4, 6, 7, 11, 12, 13, 14, 15, 18, 19, 21, 24)),
    (29, (1, 3, 4, 5, 7, 8, 12, 13, 14, 15, 16, 19, 20, 22)),
    (30, (2, 4, 5, 6, 8, 9, 13, 14, 15, 16, 17, 20, 21, 23)),
    (30, (1, 3, 5, 6, 7, 9, 10, 14, 15, 16, 17, 18, 21, 22, 24)),
    (29, (3, 5, 6, 8, 9, 10, 11, 13, 15, 19, 22, 23, 24)),
)


def word_parity(d, d29s, d30s):
    """D25..D30 for source bits d[0..23] (before the D30* inversion) and the previous word's last two bits."""
    out = []
    for star, taps in _PARITY_ROWS:
        p = d29s if star == 29 else d30s
        for t in taps:
            p ^= int(d[t - 1])
        out.append(p)
    return out


def encode_word(d, d29s, d30s, solve_tail=False):
    """30 transmitted bits of one word.  With `solve_tail` the source bits d23, d24 are chosen so that the
    transmitted D29 = D30 = 0 (words 2 and 10)."""
    d = [int(x) for x in d]
    if solve_tail:
        for d24 in (0, 1):
            for d23 in (0, 1):
                d[22], d[23] = d23, d24
                p = word_parity(d, d29s, d30s)
                if p[4] == 0 and p[5] == 0:
                    break
            else:
                continue
            break
    p = word_parity(d, d29s, d30s)
    return [b ^ d30s for b in d] + p


def _bits(value, width):
    return [(value >> (width - 1 - k)) & 1 for k in range(width)]


def encode_subframe(tow_count, subframe_id, payload, d29s=0, d30s=0):
    """300 transmitted bits.  `tow_count` = the HOW's 17-bit count (x 6 s = start of the NEXT subframe),
    `payload` = 8 x 24 source bits for words 3..10 (the last two of word 10 are overwritten)."""
    payload = np.asarray(payload).reshape(8, 24)
    tlm = list(PREAMBLE) + _bits(0x1555, 14) + [0, 0]
    how = _bits(tow_count, 17) + [0, 0] + _bits(subframe_id, 3) + [0, 0]
    out = []
    for k, src in enumerate([tlm, how] + [list(r) for r in payload]):
        w = encode_word(src, d29s, d30s, solve_tail=(k in (1, 9)))
        d29s, d30s = w[28], w[29]
        out += w
    return out


def lnav_stream(first_tow_count, first_subframe_id, n_subframes, seed):
    """Consecutive subframes (IDs cycling 1..5, TOW count advancing by one each) as 0/1 bits."""
    rng = np.random.default_rng(seed)
    bits, sid, tow = [], first_subframe_id, first_tow_count
    for _ in range(n_subframes):
        bits += encode_subframe(tow, sid, rng.integers(0, 2, size=(8, 24)))   # every subframe ends with D29 = D30 = 0
        sid, tow = sid % 5 + 1, tow + 1
    return np.array(bits, dtype=np.int8)


class WordChecker:
    """The two word-level operations a subframe synchroniser needs (the role of LNAV_CheckPreambule and
    LNAV_DecodeTOW of sydr/dsp/decoding.py:245-277,280-313), from Table 20-XIV with plain integer bits."""

    @staticmethod
    def _word_ok(b32):
        d29s, d30s = int(b32[0]), int(b32[1])
        src = [int(x) ^ d30s for x in b32[2:26]]
        return word_parity(src, d29s, d30s) == [int(x) for x in b32[26:32]]

    @classmethod
    def check_preamble(cls, bits62) -> bool:
        head = tuple(int(x) for x in bits62[2:10])
        if head != PREAMBLE and head != tuple(1 - x for x in PREAMBLE):
            return False
        return cls._word_ok(bits62[0:32]) and cls._word_ok(bits62[30:62])

    @staticmethod
    def decode_tow(bits300, d30star):
        out, star = [], int(d30star)
        for w in range(10):
            word = [int(x) for x in bits300[30 * w:30 * w + 30]]
            if star:
                word[:24] = [1 - x for x in word[:24]]
            star = word[29]
            out += word
        text = "".join(map(str, out))
        return int(text[30:47], 2) * 6, int(text[49:52], 2), text


FLAG_SUBFRAME_SYNC, FLAG_TOW_DECODED, FLAG_EPH_DECODED, FLAG_TOW_KNOWN, FLAG_EPH_KNOWN = 4, 8, 16, 32, 64


class SubframeSync:
    """NavDecoder protocol (`push(bit, track_flags) -> (flags, event | None)`, `reset()`), following
    kaplan:756-868 (`plugin="kaplan"`: packet tow = int(aligned tow)) or borre:493-579 (packet tow = the HOW's)."""

    MIN_BITS = 62            # 2 previous bits + two words
    SIZE = 362               # one subframe + MIN_BITS

    def __init__(self, cid=0, plugin="kaplan"):
        self.cid, self.plugin = cid, plugin
        self.reset()

    def reset(self):
        self.buf = np.zeros(self.SIZE, dtype=np.int64)
        self.count, self.preamble_found, self.seen, self.tow = 0, False, [False] * 5, 0

    def push(self, bit, flags):
        from types import SimpleNamespace
        b = self.buf
        b[self.count] = bit
        self.count += 1
        if self.count < self.MIN_BITS:
            return flags, None
        if not flags & FLAG_SUBFRAME_SYNC:
            idx = self.count - self.MIN_BITS
            if not WordChecker.check_preamble(b[idx:idx + self.MIN_BITS]):
                if self.count == self.SIZE:           # slide by one bit until a first preamble shows up
                    b[:-1] = b[1:].copy()
                    self.count -= 1
                return flags, None
            if self.preamble_found and idx == 300:
                flags |= FLAG_SUBFRAME_SYNC
            else:                                     # keep the preamble's 62 bits, drop what came before
                b[:self.MIN_BITS] = b[idx:idx + self.MIN_BITS].copy()
                b[self.MIN_BITS:] = 0
                self.count, self.preamble_found = self.MIN_BITS, True
        if self.count < self.SIZE:
            return flags, None
        idx = self.count - self.MIN_BITS
        if not WordChecker.check_preamble(b[idx:idx + self.MIN_BITS]):
            self.count = 0
            return flags ^ FLAG_SUBFRAME_SYNC, None
        tow, sid, text = WordChecker.decode_tow(b[2:302], b[1])
        b[:self.MIN_BITS] = b[idx:idx + self.MIN_BITS].copy()
        self.count = self.MIN_BITS
        self.tow = tow + self.count * 20 * 1e-3
        if sid > 5:                                   # kaplan:835-845: IDs 6, 7 raise IndexError there (ID 0 indexes [-1])
            return flags ^ FLAG_TOW_DECODED ^ FLAG_TOW_KNOWN, None
        self.seen[sid - 1] = True
        flags |= FLAG_TOW_DECODED | FLAG_TOW_KNOWN
        if not flags & FLAG_EPH_DECODED and all(self.seen[0:3]):
            flags |= FLAG_EPH_DECODED | FLAG_EPH_KNOWN
        return flags, SimpleNamespace(subframe_id=sid, tow=int(self.tow) if self.plugin == "kaplan" else tow, bits=text,
                                      channel_tow=self.tow)
