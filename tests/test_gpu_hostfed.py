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
