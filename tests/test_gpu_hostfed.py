"""A recording fed from the host a chunk at a time (sdr_iq_upload_queue from sdr_host_alloc memory and from a pageable
memmap) while another stream correlates the chunk before: what the kernels read is what the link carried, and the outputs
equal the one-launch pass over the same samples bit for bit (bench.py's `host_fed` leg at a size the test suite affords)."""
import numpy as np
import pytest

from oracle import sydr_oracle as orc
from sydr_amd import SdrError
from sydr_amd.engine import FMT_CI8, FMT_CI16, Engine, make_items

pytestmark = pytest.mark.gpu


def _items(fs, n_ch, total, rng):
    step = orc.CODE_RATE * (1.0 + rng.uniform(-3e-6, 3e-6, n_ch)) / fs
    start = rng.integers(0, 2000, n_ch).astype(np.int64)
    rem = rng.uniform(0, 0.03, n_ch)
    rows = []
    while True:
        n = np.ceil((1023.0 - rem) / step).astype(np.int64)
        if (start + n).max() > total:
            break
        rows.append((n.copy(), start.copy(), rem.copy()))
        rem = rem + n * step - 1023.0
        start = start + n
    e = len(rows)
    items = make_items(np.tile(np.arange(n_ch), e), np.stack([r[0] for r in rows]).reshape(-1), np.stack([r[1] for r in rows]).reshape(-1),
                       np.tile(rng.uniform(-4000, 4000, n_ch), e), np.tile(rng.uniform(0, 6.28, n_ch), e),
                       np.stack([r[2] for r in rows]).reshape(-1), np.tile(step, e))
    return items, e


@pytest.mark.parametrize("pageable", [False, True])
def test_chunks_queued_from_the_host_equal_the_one_launch_pass(engine, tmp_path, pageable):
    fs, n_ch, chunk = 25e6, 8, 200_000
    total = 12 * chunk
    rng = np.random.default_rng(5150 + pageable)
    raw = rng.integers(-100, 100, 2 * total).astype(np.int8)
    engine.iq_alloc(total, FMT_CI8)
    engine.code_slots(n_ch)
    for c in range(n_ch):
        engine.load_gps_code(c, 3 + c)
    items, n_epochs = _items(fs, n_ch, total, rng)
    spacing = (-0.5, 0.0, 0.5)
    engine.iq_upload(raw, 0)
    plan = engine.epl_plan(items, spacing, fs)
    plan.run()
    want = plan.fetch().copy()
    variant = plan.variant
    plan.close()
    assert variant & 0xF00                                   # (a straight-line kernel: reads the flipped ring image)
    # the same samples over the link, a chunk at a time, into a ring of zeros
    if pageable:
        path = tmp_path / "recording.ci8"
        raw.tofile(path)
        source = np.asarray(np.memmap(path, dtype=np.int8, mode="r"))
    else:
        source = engine.host_alloc(2 * total, np.int8)
        source[:] = raw
    try:
        engine.iq_alloc(total, FMT_CI8)
        assert not engine.iq_download(4096, 0).any()
        plan = engine.epl_plan(items, spacing, fs)
        batch = engine.stream_create()
        ends = (items["start_sample"] + items["n_samples"]).reshape(n_epochs, n_ch).max(axis=1)
        done = 0
        for k in range(total // chunk):
            engine.iq_upload_queue(source[2 * k * chunk:2 * (k + 1) * chunk], k * chunk)
            upto = int(np.searchsorted(ends, (k + 1) * chunk, side="right")) * n_ch
            if upto > done:
                plan.run(done, upto - done, stream=batch)
                done = upto
        engine.stream_sync(batch)
        engine.sync()
        assert done == len(items)
        assert plan.fetch().tobytes() == want.tobytes()
        assert np.array_equal(engine.iq_download(total, 0), raw)
        plan.close()
    finally:
        if not pageable:
            engine.host_free(source)
    if not pageable:
        with pytest.raises(ValueError):
            engine.host_free(source)                          # (given back already)


def test_upload_queue_wraps_and_refuses_what_it_cannot_take(engine):
    cap = 4096
    engine.iq_alloc(cap, FMT_CI16)
    rng = np.random.default_rng(2)
    slab = rng.integers(-3000, 3000, 2 * 1024).astype(np.int16)
    engine.iq_upload_queue(slab, cap - 256)                  # wraps at the end of the ring
    engine.sync()
    got = engine.iq_download(cap, 0)
    assert np.array_equal(got[2 * (cap - 256):], slab[:512]) and np.array_equal(got[:2 * 768], slab[512:])
    with pytest.raises(ValueError):
        engine.iq_upload_queue(slab.astype(np.int8), 0)      # not the ring's element type
    with pytest.raises(ValueError):
        engine.iq_upload_queue(slab[::2], 0)                 # not contiguous
    with pytest.raises(SdrError):
        engine.iq_upload_queue(np.zeros(2 * (cap + 8), dtype=np.int16), 0)   # longer than the ring


def test_a_ci8_ring_round_trips_whatever_way_the_samples_came_in(engine):
    """Round 6: a ci8 ring holds its bytes with the sign bit flipped (the straight-line correlators' form) -- flipped where
    samples enter, flipped back where they leave.  Every way in (synchronous upload, queued upload, the page-locked slab of a
    tick, in place and staged) at odd offsets, odd lengths and across the ring's end must read back as the host wrote it, the
    bytes around an upload untouched, a fresh ring all zeros; and every reader must see the same samples: the per-sample
    correlator, the boundary variant and the straight-line kernel agree with the oracle on a ring filled in pieces."""
    rng = np.random.default_rng(606)
    cap = 1 << 16
    engine.iq_alloc(cap, FMT_CI8)
    assert not engine.iq_download(cap, 0).any()                       # (zero samples, not 0x80 bytes)
    mirror = np.zeros(2 * cap, dtype=np.int8)
    pinned = engine.host_alloc(2 * 4096, np.int8)
    try:
        for k in range(40):
            n = int(rng.integers(1, 3000))
            off = int(rng.integers(0, cap)) if k % 3 else cap - int(rng.integers(1, n + 1))   # (every third one wraps)
            data = rng.integers(-128, 128, 2 * n).astype(np.int8)
            how = k % 4
            if how == 0:
                engine.iq_upload(data, off)
            elif how == 1:
                engine.iq_upload_queue(data, off)
                engine.sync()
            elif how == 2:                                            # a tick's slab: staged (any address) ...
                engine.iq_upload_begin(data, off)
                engine.sync()
            else:                                                     # ... or read in place out of page-locked memory (16-byte granules)
                n = 8 * max(1, min(n, 4096) // 8)
                off = off // 8 * 8
                data = data[:2 * n]
                pinned[:2 * n] = data
                engine.iq_upload_begin(pinned[:2 * n], off)
                engine.sync()
            idx = (2 * off + np.arange(2 * n)) % (2 * cap)
            mirror[idx] = data
            if k % 5 == 0:
                assert np.array_equal(engine.iq_download(cap, 0), mirror), (k, how)
        assert np.array_equal(engine.iq_download(cap, 0), mirror)
        # a download across the ring's end, at an odd offset
        got = engine.iq_download(777, cap - 300)
        assert np.array_equal(got, mirror[(2 * (cap - 300) + np.arange(2 * 777)) % (2 * cap)])
    finally:
        engine.host_free(pinned)
    # the same ring through three correlator families against the oracle
    engine.code_slots(2)
    engine.load_gps_code(0, 5)
    engine.load_gps_code(1, 17)
    rf = orc.iq_to_complex(mirror)
    spacing = (-0.5, 0.0, 0.5)
    for fs in (4.3e6, 12.3e6, 25e6):                                 # per sample / 8-sample boundary groups / chip-aligned straight line
        step = orc.CODE_RATE / fs
        n = orc.required_samples(0.2, step)
        items = make_items([0, 1, 0], n, [11, 20001, cap - n - 64], [1500.0, -3100.0, 250.0], [0.3, 1.1, 2.0], 0.2, step)
        got = engine.epl_batch(items, spacing, fs)
        for k in range(3):
            it = items[k]
            a = int(it["start_sample"])
            ref = np.array(orc.epl(rf[a:a + n], orc.pad_code(orc.gold_code(5 if it["code_slot"] == 0 else 17)), fs, float(it["carrier_hz"]),
                                   float(it["rem_carrier"]), 0.2, step, spacing))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), np.sqrt(float(n)) * 50.0), 2)
            assert np.max(np.abs(got[k] - ref) / scale) < 1e-9, (fs, k)


def test_plans_made_and_dropped_in_a_row_reuse_their_buffers_and_keep_their_results(engine):
    """Round 6: sdr_epl_plan_destroy leaves a plan's device buffers in a pool of the engine and the next plan of a similar size
    takes them (a stream correlated segment by segment makes and drops a plan per segment).  Plans of many sizes in a row,
    several alive at once, a plan dropped while another one's launch is still queued on another stream: every plan's outputs
    equal those of a plan of the same items made on a fresh engine state -- nothing of a previous owner shows through."""
    rng = np.random.default_rng(607)
    fs, cap = 25e6, 1 << 20
    raw = rng.integers(-90, 90, 2 * cap).astype(np.int8)
    engine.iq_alloc(cap, FMT_CI8)
    engine.iq_upload(raw, 0)
    engine.code_slots(4)
    for s in range(4):
        engine.load_gps_code(s, 7 * s + 1)
    spacing = (-0.5, 0.0, 0.5)
    step = orc.CODE_RATE / fs

    def some_items(count):
        n = orc.required_samples(0.1, step)
        start = rng.integers(0, cap - n - 64, count)
        return make_items(rng.integers(0, 4, count), n, start, rng.uniform(-4000, 4000, count), rng.uniform(0, 6.28, count), 0.1, step)

    sizes = [5000, 4100, 9000, 5000, 300, 5000, 12000, 6000, 4500, 5000]
    lists = [some_items(c) for c in sizes]
    want = [engine.epl_batch(it, spacing, fs) for it in lists]          # (the one-shot path: the engine's workspaces, no pool)
    other = engine.stream_create()
    alive = []
    for k, it in enumerate(lists):
        plan = engine.epl_plan(it, spacing, fs)
        plan.run(stream=other if k % 2 else 0)
        alive.append((k, plan))
        if len(alive) > 2:                                               # drop the oldest while the newest is still queued
            j, old = alive.pop(0)
            assert old.fetch().tobytes() == want[j].tobytes(), j
            old.close()
    for j, old in alive:
        assert old.fetch().tobytes() == want[j].tobytes(), j
        old.close()
