#!/usr/bin/env python3
"""Headline benchmark: IQ Msamples/s through 32-channel E/P/L correlators at fs = 25 MHz
(BASELINE.json configs[2]: 32 channels, 3 taps, 1 ms integration, 60 s synthetic ci8 stream),
plus the acquisition leg (configs[1]: PCPS, 32 PRNs, +-5 kHz @ 250 Hz) as `acquisition`.

One "step" = one pass of the correlator kernel over ONE SECOND of the stream for all 32
channels of this GPU (32 000 channel-epochs, 1.6 GB of algorithmic IQ bytes).  The IQ, the PRN
replicas and the per-epoch NCO parameters are resident in HBM before the timed region starts.
The NCO parameters are the synthetic satellites' true code/carrier trajectories (open-loop,
all epochs of a step in flight at once) -- closed-loop numbers are reported separately under
`closed_loop` because they are latency-, not bandwidth-, bound.

Multi-GPU (north_star): ONE stream with 32*N satellites, generated from the same seed on every
rank (IQ replicated, no peer traffic); rank r tracks channels shard_channels(32*N, r, N) -- weak
scaling, 32 channels per GPU, no collective on the data path.  torch.distributed is used only for
the barrier, the max-over-ranks time and the sum of the ranks' channel-sample counts.  `value` is
(channel-samples of ALL ranks / 32) / time: the job's throughput in the metric's unit.

    python bench.py --gpus 1 --steps 60 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \\
        --master-port 29500 bench.py --gpus 8 --steps 60 --warmup 2
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

FS = 25e6
N_CH = 32
SPACING = (-0.5, 0.0, 0.5)
CODE_RATE = 1.023e6
L1 = 1575.42e6
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
VERIFY_PER_RANK = 4     # N > 1: channels of every other rank that rank 0 recomputes on its own GPU and compares bitwise


def satellites(n_total=N_CH, seed=20260003):
    """The satellites of THE stream -- the same list on every rank.  PRN 1..32 at N = 1; beyond 210 channels the
    PRN numbers repeat with another Doppler / delay (config 5 is "multi-GNSS" anyway).  The amplitude shrinks with
    the count so that the int8 stream does not clip."""
    rng = np.random.default_rng(seed)
    amp = 3.0 / np.sqrt(max(1, n_total // N_CH))
    return [dict(prn=1 + i % 210, doppler=float(rng.uniform(-4500, 4500)), code_phase=float(rng.uniform(0, 1023)),
                 phase=float(rng.random()), amp=float(amp)) for i in range(n_total)]


def truth_items(sats, fs, total_samples):
    """Per-epoch NCO parameters following the reference's bookkeeping (channel_l1ca_kaplan.py:506-534)
    along each satellite's TRUE code/carrier trajectory.  Returns (items[E*C] epoch-major, E)."""
    from sydr_amd.engine import make_items
    c = len(sats)
    dop = np.array([s["doppler"] for s in sats])
    cstep = CODE_RATE * (1.0 + dop / L1) / fs
    cp0 = np.array([s["code_phase"] for s in sats])
    ph0 = np.array([s["phase"] for s in sats])
    start = np.ceil((1023.0 - cp0) / cstep).astype(np.int64)  # first sample of the next code period
    rem_code = cp0 + start * cstep - 1023.0
    n_epochs = int(total_samples / (fs * 1e-3)) - 2
    rows = []
    for _ in range(n_epochs):
        n = np.ceil((1023.0 - rem_code) / cstep).astype(np.int64)
        # replica phase that wipes the synthetic carrier off: rem = -2*pi*(f*m/fs + phase0) mod 2*pi
        cyc = dop / fs * start + ph0
        rem_carrier = (-2.0 * np.pi * (cyc - np.floor(cyc))) % (2.0 * np.pi)
        rows.append((n.copy(), start.copy(), rem_carrier, rem_code.copy()))
        rem_code = rem_code + n * cstep - 1023.0
        start = start + n
    n_all = np.stack([r[0] for r in rows]).reshape(-1)
    st_all = np.stack([r[1] for r in rows]).reshape(-1)
    rc_all = np.stack([r[2] for r in rows]).reshape(-1)
    rk_all = np.stack([r[3] for r in rows]).reshape(-1)
    slots = np.tile(np.arange(c), n_epochs)
    items = make_items(slots, n_all, st_all, np.tile(dop, n_epochs), rc_all, rk_all, np.tile(cstep, n_epochs))
    return items, n_epochs


def cpu_tracking_baseline(engine, items, n_items, budget_s):
    """Time the NumPy oracle (1 core) on the first items of the same workload; also returns its outputs."""
    from oracle import sydr_oracle as orc
    last = items[:n_items]
    hi = int((last["start_sample"] + last["n_samples"]).max())
    raw = engine.iq_download(hi, 0)
    rf = orc.iq_to_complex(raw)
    codes = {int(s): orc.pad_code(orc.gold_code(int(s) + 1)) for s in np.unique(last["code_slot"])}
    out = np.empty((n_items, 6))
    done, t0 = 0, time.perf_counter()
    for k in range(n_items):
        it = last[k]
        s, n = int(it["start_sample"]), int(it["n_samples"])
        out[k] = orc.epl(rf[s:s + n], codes[int(it["code_slot"])], FS, float(it["carrier_hz"]),
                         float(it["rem_carrier"]), float(it["rem_code"]), float(it["code_step"]), SPACING)
        done = k + 1
        if done % N_CH == 0 and time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return out[:done], done, dt, rf


def cpu_reference_c_baseline(rf, items, n_items, prns, budget_s):
    """The reference's OWN compiled legacy correlator (oracle/_ref/tracking.so = sydr/c_functions/tracking.c built where
    it lies, oracle/Makefile): generateReplica -> generateCarrier -> getCorrelator per tap, as
    sydr/old/tracking/tracking_epl_c.py:105-137 called them, on the first items of the same workload.  1 core.
    Returns (outputs, items done, seconds) or None where the library was not built."""
    import ctypes as C
    so = os.path.join(REPO, "oracle", "_ref", "tracking.so")
    if not os.path.exists(so):
        return None
    from oracle import sydr_oracle as orc
    lib = C.CDLL(so)
    dp, zp, ip = C.POINTER(C.c_double), C.c_void_p, C.POINTER(C.c_int)
    lib.generateReplica.argtypes = [dp, C.c_size_t, C.c_double, C.c_double, dp, zp]
    lib.generateCarrier.argtypes = [zp, zp, C.c_size_t, dp, dp]
    lib.getCorrelator.argtypes = [dp, dp, ip, C.c_size_t, C.c_double, C.c_double, C.c_double, dp, dp]
    codes = {int(s): orc.pad_code(orc.gold_code(int(prns[int(s)]))).astype(np.int32) for s in np.unique(items["code_slot"][:n_items])}
    n_max = int(items["n_samples"][:n_items].max())
    t = np.arange(0, n_max + 1) / FS
    replica, i_sig, q_sig = np.zeros(n_max, dtype=np.complex128), np.zeros(n_max), np.zeros(n_max)
    rem, ic, qc = np.zeros(1), np.zeros(1), np.zeros(1)
    out = np.empty((n_items, 6))
    done, t0 = 0, time.perf_counter()
    for k in range(n_items):
        it = items[k]
        a, n = int(it["start_sample"]), int(it["n_samples"])
        x = np.ascontiguousarray(rf[a:a + n])
        lib.generateReplica(t.ctypes.data_as(dp), n, float(it["carrier_hz"]), float(it["rem_carrier"]), rem.ctypes.data_as(dp),
                            replica.ctypes.data_as(zp))
        lib.generateCarrier(x.ctypes.data_as(zp), replica.ctypes.data_as(zp), n, i_sig.ctypes.data_as(dp), q_sig.ctypes.data_as(dp))
        code = codes[int(it["code_slot"])]
        for tap, sp in enumerate(SPACING):
            lib.getCorrelator(i_sig.ctypes.data_as(dp), q_sig.ctypes.data_as(dp), code.ctypes.data_as(ip), n,
                              float(it["code_step"]), float(it["rem_code"]), sp, ic.ctypes.data_as(dp), qc.ctypes.data_as(dp))
            out[k, 2 * tap], out[k, 2 * tap + 1] = ic[0], qc[0]
        done = k + 1
        if done % N_CH == 0 and time.perf_counter() - t0 > budget_s:
            break
    return out[:done], done, time.perf_counter() - t0


def _mp_worker(args):
    """One reference-style channel process: all epochs of one channel through the oracle's EPL."""
    path, rows, code_prn = args
    from oracle import sydr_oracle as orc
    raw = np.load(path, mmap_mode="r")
    code = orc.pad_code(orc.gold_code(code_prn))
    acc = 0.0
    n = 0
    for (start, ns, f, rc, rk, cs) in rows:
        x = orc.iq_to_complex(np.asarray(raw[2 * start:2 * (start + ns)]))
        acc += orc.epl(x, code, FS, f, rc, rk, cs, SPACING)[2]
        n += ns
    return n, acc


def _mp_warm(_):
    from oracle import sydr_oracle as orc  # noqa: F401  (imports paid before the clock starts)
    return os.getpid()


def cpu_baseline_all_cores(raw, items, prns, n_epochs, budget_s):
    """The oracle fanned out by CHANNEL over a process pool -- the reference's own parallelism (one OS process per
    channel, sydr/channel/channel.py:21, channelManager.py:164-171).  Workers are spawned (never forked from this
    GPU-holding process).  Returns (stream Msamples/s for the 32-channel batch, processes, wall s, epochs used)."""
    import multiprocessing as mp
    import tempfile
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    procs = max(1, min(usable, N_CH))
    per_epoch_s = 0.6e-3                                  # ~1 channel-epoch of 25 000 samples on one core
    epochs = int(max(8, min(n_epochs, budget_s * procs / (N_CH * per_epoch_s))))
    hi = int((items["start_sample"][:epochs * N_CH] + items["n_samples"][:epochs * N_CH]).max())
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        path = os.path.join(tmp, "iq.npy")
        np.save(path, raw[:2 * hi])
        jobs = []
        for c in range(N_CH):
            it = items[c:epochs * N_CH:N_CH]
            rows = [(int(a), int(b), float(f), float(rc), float(rk), float(cs)) for a, b, f, rc, rk, cs in
                    zip(it["start_sample"], it["n_samples"], it["carrier_hz"], it["rem_carrier"], it["rem_code"], it["code_step"])]
            jobs.append((path, rows, int(prns[c])))
        # (bounded waits: a worker that never comes back must not hang the whole bench line -- this leg is a reported
        # baseline, the line is the contract; a timeout is reported as such)
        pool = mp.get_context("spawn").Pool(procs)
        try:
            pool.map_async(_mp_warm, range(procs * 2)).get(timeout=180)
            t0 = time.perf_counter()
            res = pool.map_async(_mp_worker, jobs, chunksize=1).get(timeout=max(120.0, 20.0 * budget_s))
            dt = time.perf_counter() - t0
            pool.close()
        except mp.TimeoutError:
            pool.terminate()
            return None, procs, 0.0, epochs
        finally:
            pool.terminate()
    ch_samples = sum(r[0] for r in res)
    return ch_samples / N_CH / dt / 1e6, procs, dt, epochs


def kernel_of_variant(variant, n_taps, waves_per_group=1):
    """Name of the epl_kernel instantiation a plan of this variant launches on a ci8 ring, as tools/summarize_pmc.py
    spells it -- the key that ties committed counters (profiles/pmc_traffic.json) to the kernel a run actually used."""
    w = variant & 255
    if (variant >> 13) & 3:                                  # two chips per lane (correlator_chip2.h)
        return "epl2_kernel<" + {1: "4,9,14,19", 2: "5,11,17,23", 3: "1,3,5,7,9,11,13,15"}[(variant >> 13) & 3] + ">"
    if w == 26 + 16 and not variant & (0xF00 | 4096):        # 15.x / 16.x samples per chip: both block lengths in one kernel
        return f"epl_kernel<0,{n_taps},26,15,{waves_per_group},0,0,16>"
    km = w - 26 if w > 26 else 0
    ks = (variant & 0xF00) >> 8
    ki = 1 if variant & 4096 else 0
    base = 26 if w >= 26 else w
    return f"epl_kernel<0,{n_taps},{base},{km},{waves_per_group},{ks},{ki},0>"      # (last: no second block length)


def device_identity(torch, local_rank):
    """What tells two GPUs apart: PCI location and UUID of this rank's device, plus host and process."""
    import socket
    p = torch.cuda.get_device_properties(local_rank)
    ident = {"host": socket.gethostname(), "pid": os.getpid(), "local_rank": int(local_rank), "name": p.name}
    for key in ("uuid", "pci_bus_id", "pci_device_id", "pci_domain_id"):
        if hasattr(p, key):
            ident[key] = str(getattr(p, key))
    return ident


def channel_digests(got, n_ch):
    """sha256 of every channel's accumulators over the whole pass (epoch-major list: channel c = rows c, c + n_ch, ...)."""
    import hashlib
    return [hashlib.sha256(np.ascontiguousarray(got[c::n_ch]).tobytes()).hexdigest() for c in range(n_ch)]


def verify_across_ranks(eng, dist, torch, rank, local_rank, world, all_sats, mine, got, total, elapsed_own, ch_samples):
    """"per-channel outputs at 2/4/8 GPUs must be bitwise identical to 1 GPU".  Every rank hashes what its
    kernel wrote for each of its channels; rank 0 tracks VERIFY_PER_RANK channels of every OTHER rank again on its own
    GPU (same stream, same items) and compares the hashes.  Returns the `multi_gpu` record; any mismatch ends the job
    with a non-zero exit on every rank."""
    from sydr_amd.channel.manager import shard_channels
    mine_digests = channel_digests(got, N_CH)
    gathered = [None] * world
    dist.all_gather_object(gathered, {"rank": rank, "device": device_identity(torch, local_rank), "channels": list(mine),
                                      "digests": mine_digests, "ms_per_pass": elapsed_own,
                                      "Msamples_per_s": ch_samples / N_CH / max(elapsed_own, 1e-12) / 1e6})
    verdict = {"ok": True, "mismatches": []}
    record = None
    if rank == 0:
        checked = []
        for info in gathered:
            r = info["rank"]
            if r == 0:
                continue
            theirs = shard_channels(len(all_sats), r, world)
            if theirs != info["channels"]:                 # (never raise here: the other ranks wait for the verdict below)
                verdict["ok"] = False
                verdict["mismatches"].append({"rank": r, "channel": None, "why": "ranks disagree about the sharding"})
                continue
            pick = sorted(set(int(round(x)) for x in np.linspace(0, len(theirs) - 1, VERIFY_PER_RANK)))
            sats = [all_sats[theirs[i]] for i in pick]
            for k, sat in enumerate(sats):
                eng.load_gps_code(N_CH + k, sat["prn"])
            items, _ = truth_items(sats, FS, total)
            items["code_slot"] += N_CH                     # (truth_items numbers its slots from 0: these live in the spare ones)
            plan = eng.epl_plan(items, SPACING, FS)
            plan.run()
            again = channel_digests(plan.fetch(), len(sats))
            plan.close()
            for k, i in enumerate(pick):
                same = again[k] == info["digests"][i]
                checked.append([r, theirs[i], bool(same)])
                if not same:
                    verdict["ok"] = False
                    verdict["mismatches"].append({"rank": r, "channel": theirs[i]})
        devices = [g["device"] for g in gathered]
        distinct = len({(d["host"], d.get("uuid"), d.get("pci_bus_id"), d.get("pci_domain_id")) for d in devices})
        record = {"ranks_seen": len(gathered), "distinct_devices": distinct, "devices": devices,
                  "per_rank_ms_per_step": [g["ms_per_pass"] * 1e3 for g in gathered],
                  "per_rank_Msamples_per_s": [g["Msamples_per_s"] for g in gathered],
                  "channels_recomputed_on_rank0": checked, "bitwise_identical": verdict["ok"],
                  "check": f"rank 0 re-tracked {VERIFY_PER_RANK} channels of every other rank over the whole stream on its "
                           "own GPU; sha256 of the fp64 accumulators equal"}
    box = [verdict]
    dist.broadcast_object_list(box, src=0)
    if not box[0]["ok"]:
        if rank == 0:
            print(json.dumps({"error": "per-channel outputs differ between ranks", "mismatches": box[0]["mismatches"]}),
                  file=sys.stderr)
        dist.destroy_process_group()
        raise SystemExit(3)
    return record


def counters_of_this_build():
    """profiles/pmc_traffic.json when it was taken on the library this process runs (sdr_build_id), else {}."""
    pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
    try:
        import sydr_amd
        info = json.load(open(pmc))
        return info if info.get("build_id") == sydr_amd.load().sdr_build_id().decode() else {}
    except Exception:
        return {}


def rates_leg(eng, seconds=2.0):
    """E/P/L throughput of 32 channels x 3 taps at the sampling rates a receiver front end is likely to have
    (gnsssignal.py:62-70 takes any fs; config/receiver.ini:18): which kernel variant a plan of that rate gets and what it
    reaches of the roof by SURVEY 8(d)'s bytes -- one launch per pass over a `seconds` stream, outputs of the first
    epochs checked against the oracle."""
    from oracle import sydr_oracle as orc
    from sydr_amd.engine import FMT_CI8, FMT_CI16
    out = []
    # (..., and 16-bit recordings -- rfsignal.py:36-41 reads int16 as readily as int8 -- at the shipped rate and the headline's:
    # SURVEY 8(d) charges 4 B per channel-sample there)
    for fs, fmt in [(f, FMT_CI8) for f in (4e6, 10e6, 12e6, 16.368e6, 18e6, 20e6, 22e6, 25e6, 32e6, 40e6, 50e6)] + [(10e6, FMT_CI16), (25e6, FMT_CI16)]:
        bytes_per_sample = 2.0 if fmt == FMT_CI8 else 4.0
        total = int(seconds * fs) // 8 * 8
        eng.iq_alloc(total, fmt)
        eng.code_slots(N_CH)
        sats = satellites(N_CH, seed=20260020)
        for k, sat in enumerate(sats):
            eng.load_gps_code(k, sat["prn"])
        eng.iq_synth(sats, fs, 12.0, 20260020, 0, total)
        items, n_epochs = truth_items(sats, fs, total)
        plan = eng.epl_plan(items, SPACING, fs)
        n_run = n_epochs * N_CH
        for _ in range(40):                                    # (clocks: tools/epl_ramp.py)
            plan.run(0, n_run)
        eng.sync()
        eng.prof_reset()
        eng.prof_enable(True)
        steps = 20
        for _ in range(steps):
            plan.run(0, n_run)
        eng.sync()
        eng.prof_enable(False)
        kern_ms, launches = eng.prof_read("epl_kernel")
        eng.prof_reset()
        ch_samples = float(items["n_samples"][:n_run].sum())
        avg_s = kern_ms / max(1, launches) * 1e-3
        got = plan.fetch()
        variant = plan.variant
        plan.close()
        n_chk = N_CH * 2
        hi = int((items["start_sample"][:n_chk] + items["n_samples"][:n_chk]).max())
        rf = orc.iq_to_complex(eng.iq_download(hi, 0))
        err = 0.0
        for k in range(n_chk):
            it = items[k]
            a, n = int(it["start_sample"]), int(it["n_samples"])
            ref = np.array(orc.epl(rf[a:a + n], orc.pad_code(orc.gold_code(sats[int(it["code_slot"])]["prn"])), fs, float(it["carrier_hz"]),
                                   float(it["rem_carrier"]), float(it["rem_code"]), float(it["code_step"]), SPACING))
            scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), 1.0), 2)
            err = max(err, float(np.max(np.abs(got[k] - ref) / scale)))
        if err > 1e-6:
            raise SystemExit(f"GPU/oracle mismatch in the rates leg at {fs:g} Hz: {err:.3e}")
        w = variant & 255
        family = "two chips per lane, straight line" if (variant >> 13) & 3 else ("half-chip view, " if variant & 65536 else "")
        if not (variant >> 13) & 3:
            family += ("one chip per lane, straight line (compile-time block length and tap positions)" if (variant & 0xF00) else
                       "one chip per lane, whole-chip taps" if variant & 4096 else
                       "one chip per lane, both block lengths (15 / 16) compiled in, run-time tap positions" if w == 26 + 16 and not variant & (0xF00 | 4096) else
                       "one chip per lane, compile-time block length" if w > 26 else
                       "one chip per lane, run-time positions" if w >= 26 else
                       f"{w}-sample boundary groups" if w else "per sample")
        kname = kernel_of_variant(variant & 0xFFFF, len(SPACING))
        if fmt == FMT_CI16:
            kname = kname.replace("<0,", "<1,", 1)             # (the template's first argument is the ring format)
        out.append({"fs_hz": fs, "iq_format": "ci8" if fmt == FMT_CI8 else "ci16", "samples_per_chip": fs / CODE_RATE,
                    "Msamples_per_s": ch_samples / N_CH / avg_s / 1e6 if launches else 0.0,
                    "x_realtime": ch_samples / N_CH / avg_s / fs if launches else 0.0, "kernel_variant": kname,
                    "correlator": family, "plan_variant": int(variant),
                    "roofline_frac": bytes_per_sample * ch_samples / avg_s / 1e9 / HBM_PEAK_GBS if launches else 0.0,
                    "max_rel_err_gpu_vs_oracle": err})
    return {"config": {"workload": f"32 channels, E/P/L +-0.5 chip, ci8 (and ci16 at 10 / 25 MHz), {seconds:g} s stream per rate, one launch per "
                                   "pass (kernel time from HIP events on the launch stream); roofline_frac = 2 B (ci16: 4 B) per "
                                   "channel-sample / 8 TB/s"},
            "rates": out}


def ref_config_leg(eng, cpu_seconds=4.0):
    """The reference's own shipped configuration (config/receiver.ini:18-20: 10 MHz, 8-bit I/Q;
    config/channels/channel_GPS_L1CA_kaplan.ini:6-10: PCPS with a 300 Hz grid, 1 x 10 ms non-coherent; taps +-0.5 chip),
    32 channels: tracking throughput (the two-chips-per-lane kernel of correlator_chip2.h serves this rate) and
    acquisition time, each with its roofline by algorithmic bytes, the oracle timed beside it and checked against it."""
    from oracle import sydr_oracle as orc
    from sydr_amd.engine import FMT_CI8
    fs, seconds = 10e6, 20.0
    total = int(seconds * fs) // 8 * 8
    eng.iq_alloc(total, FMT_CI8)
    eng.code_slots(N_CH)
    sats = satellites(N_CH, seed=20260010)
    for k, sat in enumerate(sats):
        eng.load_gps_code(k, sat["prn"])
    eng.iq_synth(sats, fs, 12.0, 20260010, 0, total)
    items, n_epochs = truth_items(sats, fs, total)
    plan = eng.epl_plan(items, SPACING, fs)
    n_run = n_epochs * N_CH
    for _ in range(3):
        plan.run(0, n_run)
    eng.sync()
    eng.prof_reset()
    eng.prof_enable(True)
    steps = 10
    t0 = time.perf_counter()
    for _ in range(steps):
        plan.run(0, n_run)
    eng.sync()
    wall = time.perf_counter() - t0
    eng.prof_enable(False)
    kern_ms, launches = eng.prof_read("epl_kernel")
    eng.prof_reset()
    ch_samples = float(items["n_samples"][:n_run].sum())
    avg_s = kern_ms / max(1, launches) * 1e-3
    got = plan.fetch()
    variant = plan.variant
    plan.close()
    # oracle: check + CPU baseline on the first epochs of the same stream
    n_chk = N_CH * 4
    hi = int((items["start_sample"][:n_chk] + items["n_samples"][:n_chk]).max())
    rf = orc.iq_to_complex(eng.iq_download(max(hi, int(10 * fs * 1e-3)), 0))
    codes = [orc.pad_code(orc.gold_code(s["prn"])) for s in sats]
    done, t0 = 0, time.perf_counter()
    err = 0.0
    for k in range(len(items)):
        it = items[k]
        a, n = int(it["start_sample"]), int(it["n_samples"])
        if a + n > len(rf):
            break
        ref = np.array(orc.epl(rf[a:a + n], codes[int(it["code_slot"])], fs, float(it["carrier_hz"]), float(it["rem_carrier"]),
                               float(it["rem_code"]), float(it["code_step"]), SPACING))
        scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), 1.0), 2)
        err = max(err, float(np.max(np.abs(got[k] - ref) / scale)))
        done = k + 1
        if done >= n_chk and (time.perf_counter() - t0 > cpu_seconds or done >= N_CH * 200):
            break
    cpu_dt = time.perf_counter() - t0
    if err > 1e-6:
        raise SystemExit(f"GPU/oracle mismatch in the ref_config tracking leg: {err:.3e}")
    tracking = {"metric": "IQ Msamples/s through 32-ch E/P/L correlators @10 MHz fs", "value": ch_samples * steps / N_CH / wall / 1e6,
                "unit": "Msamples/s", "x_realtime": ch_samples * steps / N_CH / wall / fs, "ms_per_pass": wall / steps * 1e3,
                "kernel_variant": kernel_of_variant(variant, len(SPACING)),
                "roofline": {"bound": "hbm", "achieved": 2.0 * ch_samples / avg_s / 1e9 if launches else 0.0, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": 2.0 * ch_samples / avg_s / 1e9 / HBM_PEAK_GBS if launches else 0.0,
                             "traffic": None, "kernel": kernel_of_variant(variant, len(SPACING)), "avg_launch_ms": avg_s * 1e3,
                             "launches": int(launches), "algorithmic_bytes_per_launch": 2.0 * ch_samples},
                "cpu_baseline": {"value": float(items["n_samples"][:done].sum()) / N_CH / cpu_dt / 1e6, "unit": "Msamples/s",
                                 "cores": 1, "kind": "port", "sample": f"first {done} channel-epochs of the same stream "
                                 f"through oracle/sydr_oracle.py:epl, {cpu_dt:.1f} s", "max_rel_err_gpu_vs_oracle": err}}
    info = counters_of_this_build()
    if info.get("epl2_kernel", "").replace(" ", "") == tracking["kernel_variant"] and info.get("epl2_kernel_hbm_bytes_per_epoch"):
        tracking["roofline"]["traffic"] = info["epl2_kernel_hbm_bytes_per_epoch"] * n_run      # (counters per channel-epoch x this launch)
        tracking["roofline"]["traffic_over_algorithmic"] = tracking["roofline"]["traffic"] / (2.0 * ch_samples)
    # acquisition as the shipped ini asks for it: +-5 kHz @ 300 Hz (34 bins), 1 ms coherent x 10 non-coherent
    slots = np.arange(N_CH)
    rng_hz, step_hz, coh, noncoh = 5000.0, 300.0, 1, 10
    for _ in range(10):
        eng.pcps(slots, 0, fs, 0.0, rng_hz, step_hz, coh, noncoh)
    reps = 10
    t0 = time.perf_counter()
    for _ in range(reps):
        pb, pc, pr, _ = eng.pcps(slots, 0, fs, 0.0, rng_hz, step_hz, coh, noncoh)
    acq_ms = (time.perf_counter() - t0) / reps * 1e3
    eng.prof_reset()
    eng.prof_enable(True, calls_only=True)
    for _ in range(reps):
        eng.pcps(slots, 0, fs, 0.0, rng_hz, step_hz, coh, noncoh)
    eng.prof_enable(False)
    pk_ms, _ = eng.prof_read("call_pcps")
    pk_ms /= reps
    eng.prof_reset()
    n_code, bins = 10000, len(orc.doppler_bins(rng_hz, step_hz))
    t0 = time.perf_counter()
    ok = True
    for k in range(2):
        cmap = orc.pcps_map(rf[:n_code * coh * noncoh].reshape(1, -1), 0.0, fs, orc.code_spectrum(orc.gold_code(sats[k]["prn"]), fs),
                            rng_hz, step_hz, n_code, coh, noncoh)
        peak, ratio = orc.two_peak_compare(cmap, n_code, round(fs / CODE_RATE))
        ok &= peak == [int(pb[k]), int(pc[k])] and abs(ratio - pr[k]) <= 1e-6 * ratio
    cpu_ms = (time.perf_counter() - t0) / 2 * 1e3
    if not ok:
        raise SystemExit("PCPS peak mismatch vs oracle in the ref_config leg")
    # SURVEY 8d per (PRN, bin) and millisecond block: 16 spectrum + 16 code spectrum; the 8 of a map that accumulates the blocks
    # are not charged: the call asks for indices and ratio and the search keeps the non-coherent sum in registers (pcps_fused10k.h)
    algo = N_CH * bins * noncoh * 32.0 * n_code
    acquisition = {"metric": "acquisition ms/PRN", "value": acq_ms / N_CH, "unit": "ms/PRN", "ms_total_32_prn": acq_ms,
                   "kernel_ms_32_prn": pk_ms, "cpu_ms_per_prn_1core": cpu_ms, "peaks_match_oracle": bool(ok),
                   "roofline": {"bound": "hbm", "achieved": algo / (pk_ms * 1e-3) / 1e9 if pk_ms else 0.0, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": algo / (pk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pk_ms else 0.0, "traffic": None,
                                "kernel": "pcps_* (all kernels of one sdr_pcps call)", "algorithmic_bytes_per_call": algo},
                   "cpu_baseline": {"value": cpu_ms, "unit": "ms/PRN", "cores": 1, "kind": "port",
                                    "sample": "2 PRNs x 34 bins x 10 blocks through oracle/sydr_oracle.py:pcps_map"}}
    if info.get("pcps_10mhz_hbm_bytes_per_call"):
        acquisition["roofline"]["traffic"] = info["pcps_10mhz_hbm_bytes_per_call"]
        acquisition["roofline"]["traffic_over_algorithmic"] = info["pcps_10mhz_hbm_bytes_per_call"] / algo
    return {"config": {"workload": "the reference's shipped configuration: fs=10 MHz ci8 (config/receiver.ini:18-20), 32 channels, "
                                   "E/P/L +-0.5 chip, 20 s stream in one launch per pass; PCPS +-5 kHz @ 300 Hz (34 bins), "
                                   "1 ms x 10 non-coherent (channel_GPS_L1CA_kaplan.ini:6-10), indices + ratio (no map: 32 N bytes per "
                                   "(PRN, bin, block)), code spectra cached between calls"},
            "tracking": tracking, "acquisition": acquisition,
            # the loops closed on the device at this rate (clusters of 8 workgroups per channel, 8-sample boundary groups)
            "closed_loop": closed_loop_leg(eng, items, min(n_epochs, 2000), fs=fs)}


def host_fed_leg(eng, items, n_epochs, total, one_launch_out, batch_stream, chunk_s=1.0, pageable_seconds=12.0):
    """The headline stream fed FROM THE HOST, as a file reader hands it over (rfsignal.py:58-132 reads the recording chunk by
    chunk): the 60 s of ci8 sit in page-locked host memory, go into the (zeroed) ring a second at a time as asynchronous copies
    on the engine's stream (sdr_iq_upload_queue) while the batch stream correlates the second before (the plan's item ranges;
    a ci8 chunk is flipped into the ring's sign-flipped form behind its copy, and the batch stream waits for an event recorded
    behind that) -- double-buffered by the two streams, one plan for the whole list made inside the timed region.  Outputs must be
    bit-identical to the one-launch pass over the device-born stream.  Then the same from a pageable np.memmap of a file
    (what RFSignal hands out), over the first `pageable_seconds` of the stream.  The bound is the host link, not the kernel."""
    import tempfile
    from sydr_amd.engine import FMT_CI8
    chunk = int(chunk_s * FS) // 8 * 8
    host = eng.host_alloc(2 * total, np.int8)
    try:
        for a in range(0, total, 8 * chunk):                    # (the device-born stream becomes "the recording")
            b = min(total, a + 8 * chunk)
            host[2 * a:2 * b] = eng.iq_download(b - a, a)
        # item ranges per chunk: an epoch goes with the first chunk that completes it for every channel
        ends = (items["start_sample"] + items["n_samples"])[:n_epochs * N_CH].reshape(n_epochs, N_CH).max(axis=1)
        bounds = [min(total, (k + 1) * chunk) for k in range((total + chunk - 1) // chunk)]
        upto = np.searchsorted(ends, bounds, side="right") * N_CH   # items complete once chunk k is in

        def one_pass(source, n_chunks, zero_first):
            if zero_first:
                eng.iq_alloc(total, FMT_CI8)                      # a ring of zeros: what is correlated below came over the link
            eng.sync()
            t0 = time.perf_counter()
            plan = eng.epl_plan(items, SPACING, FS)
            done = 0
            for k in range(n_chunks):
                a, b = k * chunk, bounds[k]
                eng.iq_upload_queue(source[2 * a:2 * b], a)
                if upto[k] > done:
                    plan.run(done, int(upto[k]) - done, stream=batch_stream)
                    done = int(upto[k])
            eng.stream_sync(batch_stream)
            eng.sync()
            dt = time.perf_counter() - t0
            return plan, done, dt

        plan, done, first_dt = one_pass(host, len(bounds), True)
        got = plan.fetch()[:done]
        same = bool(done == n_epochs * N_CH and got.tobytes() == np.ascontiguousarray(one_launch_out[:done]).tobytes())
        plan.close()
        times = []
        for _ in range(3):
            plan, done, dt = one_pass(host, len(bounds), False)
            plan.close()
            times.append(dt)
        dt = min(times)
        out = {"x_realtime": total / FS / dt, "pcie_GBps": 2.0 * total / dt / 1e9, "chunk_s": chunk / FS, "ms_per_pass": dt * 1e3,
               "first_pass_ms": first_dt * 1e3, "passes_ms": [t * 1e3 for t in times], "stream_seconds": total / FS,
               "bitwise_identical_to_one_launch": same, "bound": "pcie (host link): the kernel alone runs the stream ~4x faster",
               "memory": "page-locked (sdr_host_alloc)", "includes": "plan creation (one plan, whole list), every copy, the sign flip of every chunk "
               "(a ci8 ring holds its bytes sign-flipped), every launch; one chunk's copy runs beside the chunk before's correlation (two HIP streams)"}
        if not same:
            raise SystemExit("host-fed pass differs from the one-launch pass")
        # ... and from a pageable np.memmap of a file
        n_page = min(total, int(pageable_seconds * FS) // chunk * chunk)
        if n_page >= 2 * chunk:
            tmpdir = "/dev/shm" if os.path.isdir("/dev/shm") else None
            with tempfile.TemporaryDirectory(dir=tmpdir) as tmp:
                path = os.path.join(tmp, "recording.ci8")
                host[:2 * n_page].tofile(path)
                rec = np.asarray(np.memmap(path, dtype=np.int8, mode="r"))
                k_page = n_page // chunk
                one_pass(rec, k_page, True)[0].close()
                plan, done_p, dt_p = one_pass(rec, k_page, False)
                got_p = plan.fetch()[:done_p]
                same_p = bool(got_p.tobytes() == np.ascontiguousarray(one_launch_out[:done_p]).tobytes())
                plan.close()
                del rec
            out["pageable"] = {"x_realtime": n_page / FS / dt_p, "pcie_GBps": 2.0 * n_page / dt_p / 1e9, "stream_seconds": n_page / FS,
                               "ms_per_pass": dt_p * 1e3, "bitwise_identical_to_one_launch": same_p,
                               "memory": "np.memmap of a file (pageable; the runtime stages it), plan of the whole 60 s list"}
            if not same_p:
                raise SystemExit("host-fed pass (pageable) differs from the one-launch pass")
        return out
    finally:
        eng.host_free(host)


def per_tick_leg(eng, read_ahead=0, tick_server=False):
    """ChannelManager.addNewRFData(1 ms) + run() from Python, 32 channels (tools/per_tick_rate.py); read_ahead: the same
    calls with ChannelManager.enableReadAhead (blocks of epochs computed ahead and handed out tick by tick); tick_server: the
    same calls answered by the resident kernel (sdr_set_option "tick_server": no launch, no stream synchronisation per tick)."""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import per_tick_rate
    return per_tick_rate.measure(600 if read_ahead or tick_server else 300, N_CH, engine=eng, read_ahead=read_ahead, tick_server=tick_server)


def per_tick_c_leg(ticks=800):
    """The same per-millisecond loop from plain C (examples/receiver_loop.c: sdr_iq_upload_begin + sdr_bank_tick_mirrored per
    tick, 32 channels @ 25 MHz, the host as IQ source), built with the box's gcc against the header alone and run as child
    processes of its own engine: plain ticks, and plain ticks with the recording in page-locked memory (slabs read in place).
    What the C-ABI delivers without the interpreter's share of a tick.  (The resident tick server is NOT timed here: its
    kernels hold the whole device, and as the child of a process that keeps a HIP context of its own -- this one: torch's --
    they are time-sliced against that context's idle queues: 37-54 us per tick where the same program started from a shell
    takes 21-22.  The Python `per_tick_server` leg, in this process, is the server's figure.)"""
    import re
    import subprocess
    import tempfile
    out = {"ticks": ticks, "source": "examples/receiver_loop.c"}
    # (an optional leg never costs the line: under a profiler's preloaded library -- gcc is a driver that re-execs, and the
    # child's kernels would land in the same output directory -- it is skipped; no compiler, or a child that cannot start,
    # is an "error" entry)
    if "rocprofiler" in os.environ.get("LD_PRELOAD", "") or any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ):
        return {"skipped": "under rocprofv3 (its library preloaded / ROCP* set): no child processes"}
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "receiver_loop")
        lib_dir = os.path.join(REPO, "sydr_amd")
        try:
            build = subprocess.run(["gcc", "-std=c99", "-O2", "-I", os.path.join(REPO, "include"), os.path.join(REPO, "examples", "receiver_loop.c"),
                                    "-L", lib_dir, "-lsydr_amd", "-lm", "-Wl,-rpath," + lib_dir, "-o", exe], capture_output=True, text=True,
                                   timeout=120)
        except (OSError, subprocess.SubprocessError) as exc:
            return {"error": f"gcc: {exc!r}"[:300]}
        if build.returncode:
            return {"error": "gcc: " + build.stderr[-300:]}
        for key, extra in (("plain", []), ("plain_pinned_source", ["pinned"])):
            try:
                run = subprocess.run([exe, str(ticks)] + extra, capture_output=True, text=True, timeout=120)
            except subprocess.TimeoutExpired:
                out[key] = {"error": "timed out"}
                continue
            except (OSError, subprocess.SubprocessError) as exc:
                out[key] = {"error": repr(exc)[:300]}
                continue
            m = re.search(r"([0-9.]+) us per tick = ([0-9.]+) x real time", run.stdout)
            locked = re.search(r"(\d+) of 32 channels on their Doppler", run.stdout)
            if not m:
                out[key] = {"error": (run.stdout + run.stderr)[-300:]}
                continue
            out[key] = {"us_per_tick": float(m.group(1)), "x_realtime": float(m.group(2)),
                        "channels_on_their_doppler": int(locked.group(1)) if locked else None}
    return out


def closed_loop_leg(eng, items, n_epochs, n_ch=N_CH, fs=None):
    """On-device loop closure (persistent workgroups, Kaplan loops): latency-bound, so it is reported beside,
    not instead of, the open-loop correlator throughput.  n_ch = 32: each channel on a cluster of 8 CUs (lowest
    latency); n_ch = 768: channels beyond 32 re-track the same 32 satellites, three workgroups per CU (highest
    aggregate channel x real-time rate)."""
    from sydr_amd._lib import LoopCfg, TrackState
    fs = FS if fs is None else fs
    cfg = LoopCfg()
    cfg.loop_kind, cfg.n_taps, cfg.fs = 1, 3, fs
    for t, s in enumerate(SPACING):
        cfg.spacing_wide[t] = cfg.spacing_narrow[t] = s
    wn = 2.0 * 8.0 * 0.7 / (4.0 * 0.7**2 + 1)              # channel_GPS_L1CA_kaplan.ini DLL: 2 Hz, zeta 0.7, gain 1
    cfg.dll_tau1, cfg.dll_tau2, cfg.dll_pdi, cfg.dll_threshold = 1.0 / wn**2, 2.0 * 0.7 / wn, 0.001, 10.0
    cfg.fll_bw_pullin, cfg.fll_bw_wide, cfg.fll_bw_narrow, cfg.fll_thr_wide, cfg.fll_thr_narrow = 100.0, 50.0, 15.0, 0.5, 0.8
    cfg.pll_bw_wide, cfg.pll_bw_narrow, cfg.pll_thr_wide, cfg.pll_thr_narrow = 25.0, 15.0, 0.5, 0.8
    states = []
    for c in range(n_ch):
        it = items[c % N_CH]
        st = TrackState()
        st.code_slot, st.n_samples, st.current_sample = int(it["code_slot"]), int(it["n_samples"]), int(it["start_sample"])
        st.carrier_hz, st.code_hz = float(it["carrier_hz"]), CODE_RATE
        st.rem_carrier, st.rem_code, st.code_step = float(it["rem_carrier"]), float(it["rem_code"]), CODE_RATE / fs
        st.fll_bw, st.pll_bw, st.lock_state = 100.0, 25.0, 1
        states.append(st)
    eng.track_closed_loop([TrackState.from_buffer_copy(s) for s in states], cfg, 50, want_traj=False)  # warm
    eng.prof_reset()
    eng.prof_enable(True)
    t0 = time.perf_counter()
    end, _ = eng.track_closed_loop(states, cfg, n_epochs, want_traj=False)
    wall = time.perf_counter() - t0
    eng.prof_enable(False)
    kern_ms, _ = eng.prof_read("track_kernel")
    eng.prof_reset()
    samples = float(np.mean([e.current_sample - s.current_sample for e, s in zip(end, states)]))
    lost = sum(abs(e.carrier_hz - s.carrier_hz) > 100.0 for e, s in zip(end, states))
    return {"metric": f"closed-loop tracking, {n_ch} channels" + ("" if fs == FS else f" @{fs / 1e6:g} MHz") + ", loop closure on device (Kaplan FLL/PLL/DLL)",
            "epochs": n_epochs, "kernel_ms": kern_ms, "wall_ms": wall * 1e3,
            "x_realtime": samples / fs / (kern_ms * 1e-3), "Msamples_per_s": samples / (kern_ms * 1e-3) / 1e6,
            "channel_realtimes": n_ch * samples / fs / (kern_ms * 1e-3),
            "us_per_epoch": kern_ms * 1e3 / n_epochs, "channels_lost": int(lost)}


def multignss_workload(args, rank, local_rank, world, torch, dist, eng=None, emit=True):
    """BASELINE configs 4-5: per GPU 32 GPS L1 C/A + 32 E1-like (seeded 4092-chip codes, BOC(1,1)) channels,
    5 taps VE/E/P/L/VL, fs = 50 MHz, 4 ms epochs.  Selected with --workload multignss; the default run carries a
    short version of it as the `multignss` key.  """
    from sydr_amd.engine import FMT_CI8, Engine, make_items
    fs, n_gps, n_e1, taps = 50e6, 32, 32, (-1.0, -0.5, 0.0, 0.5, 1.0)
    own_engine = eng is None
    if own_engine:
        eng = Engine(local_rank)
    total = int(args.stream_seconds * fs) // 8 * 8
    eng.iq_alloc(total, FMT_CI8)
    # ONE stream for the whole job: 32*N GPS + 32*N E1-like satellites from one seed, generated on every rank; this
    # rank tracks its shard of each constellation (64 channels per GPU)
    from sydr_amd.channel.manager import shard_channels
    n_gps_all, n_e1_all = n_gps * world, n_e1 * world
    mine = shard_channels(n_gps_all, rank, world)
    eng.code_slots(n_gps + n_e1_all + n_e1, 8184)           # [my GPS][every E1 chip-rate code][my E1 half-chip codes]
    rng = np.random.default_rng(20260004)
    amp = 2.5 / np.sqrt(world)
    all_sats, sats_gps, sats_e1 = [], [], []
    for i in range(n_gps_all):
        sat = dict(prn=1 + i % 210, doppler=float(rng.uniform(-4500, 4500)), code_phase=float(rng.uniform(0, 1023)),
                   phase=float(rng.random()), amp=amp, chips=1023, half=1)
        all_sats.append(sat)
        if i in mine:
            sat["corr_slot"] = len(sats_gps)
            eng.load_gps_code(sat["corr_slot"], sat["prn"])
            sats_gps.append(sat)
    for i in range(n_e1_all):
        code = np.where(rng.random(4092) < 0.5, -1, 1).astype(np.int8)   # seeded stand-in for an E1 memory code
        eng.set_code(n_gps + i, code)                # chip-rate code: read by the generator
        sat = dict(slot=n_gps + i, boc=True, doppler=float(rng.uniform(-4500, 4500)),
                   code_phase=float(rng.uniform(0, 4092)), phase=float(rng.random()), amp=amp, chips=4092, half=2)
        all_sats.append(sat)
        if i in mine:
            half = np.empty(8184, dtype=np.int8)
            half[0::2], half[1::2] = code, -code
            sat["corr_slot"] = n_gps + n_e1_all + len(sats_e1)
            eng.set_code(sat["corr_slot"], half)     # half-chip code: read by the correlator
            sats_e1.append(sat)
    sats = sats_gps + sats_e1
    eng.iq_synth(all_sats, fs, 12.0, 20260004, 0, total)

    def items_for(group):
        dop = np.array([s["doppler"] for s in group])
        chips = float(group[0]["chips"])
        half = group[0]["half"]
        periods = 4 if chips == 1023 else 1                        # 4 ms of code
        cstep = CODE_RATE * (1.0 + dop / L1) / fs
        cp0 = np.array([s["code_phase"] for s in group])
        ph0 = np.array([s["phase"] for s in group])
        start = np.ceil((chips - cp0) / cstep).astype(np.int64)
        rem = cp0 + start * cstep - chips
        span = chips * periods
        rows = []
        while np.all(start + np.ceil(span / cstep) + 2 < total):
            n = np.ceil((span - rem) / cstep).astype(np.int64)
            cyc = dop / fs * start + ph0
            rows.append((n.copy(), start.copy(), (-2.0 * np.pi * (cyc - np.floor(cyc))) % (2.0 * np.pi), rem.copy()))
            rem = rem + n * cstep - span
            start = start + n
        e = len(rows)
        slots = np.tile([s["corr_slot"] for s in group], e)
        return make_items(slots, np.stack([r[0] for r in rows]).reshape(-1), np.stack([r[1] for r in rows]).reshape(-1),
                          np.tile(dop, e), np.stack([r[2] for r in rows]).reshape(-1),
                          half * np.stack([r[3] for r in rows]).reshape(-1), np.tile(half * cstep, e)), e

    gps_items, e_gps = items_for(sats[:n_gps])
    e1_items, e_e1 = items_for(sats[n_gps:])
    plans = [(eng.epl_plan(gps_items, taps, fs), n_gps, e_gps, gps_items),
             (eng.epl_plan(e1_items, tuple(2 * t for t in taps), fs), n_e1, e_e1, e1_items)]
    per_step = 250                                                   # epochs per step = 1 s of stream
    n_avail = max(1, min(e_gps, e_e1) // per_step)

    def run_step(k):                                                 # one pass over the whole stream: one launch per signal
        for plan, n_ch, _, _ in plans:
            plan.run(0, n_avail * per_step * n_ch)

    for k in range(args.warmup):
        run_step(k)
    eng.sync()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    eng.prof_reset()
    eng.prof_enable(True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        run_step(k)
    eng.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=("cpu" if os.environ.get("SYDR_BENCH_REHEARSE") == "1" else "cuda"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    eng.prof_enable(False)
    kern_ms, launches = eng.prof_read("epl_kernel")
    ch_samples = 0
    for _, n_ch, _, items in plans:
        ch_samples += int(items["n_samples"][:n_avail * per_step * n_ch].sum()) * args.steps
    stream_samples = ch_samples / (n_gps + n_e1)
    job = float(ch_samples)
    if world > 1:
        t = torch.tensor([job], dtype=torch.float64, device=("cpu" if os.environ.get("SYDR_BENCH_REHEARSE") == "1" else "cuda"))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        job = float(t.item())
    value = job / (n_gps + n_e1) / elapsed / 1e6
    achieved = 2.0 * ch_samples / (kern_ms * 1e-3) / 1e9 if launches else 0.0
    result = {"metric": "IQ Msamples/s through 64-ch 5-tap VE/E/P/L/VL correlators @50 MHz fs, 4 ms integration",
              "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "f64", "data": "synthetic",
              "config": {"workload": "GPS L1 C/A (32 ch) + E1-like BOC(1,1) with seeded 4092-chip codes (32 ch) per GPU, "
                                     f"5 taps, fs=50 MHz, 4 ms epochs, {args.stream_seconds:g} s ci8 stream, 1 step = one pass over the "
                                     f"whole stream = one launch of {n_avail} s per signal",
                         "channels_per_gpu": n_gps + n_e1, "fs_hz": fs, "taps": 5, "iq_format": "ci8",
                         "note": "no reference implementation exists for this configuration (SURVEY.md section 0); "
                                 "parity is against the oracle's generalised restatement"},
              "x_realtime": stream_samples / elapsed / fs,
              "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "epl_kernel",
                           "avg_launch_ms": kern_ms / max(1, launches), "launches": int(launches)}}
    pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and launches:
        try:
            info = json.load(open(pmc))
            ran = [kernel_of_variant(pl.variant, 5, 4) for pl, *_ in plans]
            result["roofline"]["kernel_variant"] = ran[0] if len(set(ran)) == 1 else ran
            if info.get("multignss_hbm_bytes_per_wave") and all(r == info.get("multignss_kernel", "").replace(" ", "") for r in ran):
                waves_per_launch = sum(n_avail * per_step * n_ch for _, n_ch, _, _ in plans) / len(plans)
                traffic = info["multignss_hbm_bytes_per_wave"] * waves_per_launch
                result["roofline"].update({"traffic": traffic, "traffic_GBps": traffic / (kern_ms / launches * 1e-3) / 1e9,
                                           "traffic_source": {"file": info.get("source"), "git_head": info.get("git_head")}})
            elif info.get("multignss_kernel"):
                result["roofline"]["traffic_note"] = (f"profiles/pmc_traffic.json holds counters of {info['multignss_kernel']!r}, "
                                                      f"this run launched {ran!r}: no traffic figure")
        except Exception:
            pass
    if rank == 0 and world == 1:
        from oracle import sydr_oracle as orc
        got = plans[1][0].fetch()
        it = e1_items[0]
        raw = eng.iq_download(int(it["start_sample"] + it["n_samples"]), 0)
        chips = eng.read_code(int(it["code_slot"])).astype(np.float64)
        t0 = time.perf_counter()
        ref = np.array(orc.epl(orc.iq_to_complex(raw)[int(it["start_sample"]):], orc.pad_code(chips), fs,
                               float(it["carrier_hz"]), float(it["rem_carrier"]), float(it["rem_code"]),
                               float(it["code_step"]), tuple(2 * t for t in taps)))
        dt = time.perf_counter() - t0
        scale = np.repeat(np.maximum(np.hypot(ref[0::2], ref[1::2]), 1.0), 2)
        err = float(np.max(np.abs(got[0] - ref) / scale))
        if err > 1e-6:
            raise SystemExit(f"GPU/oracle mismatch in multignss bench: {err:.3e}")
        result["cpu_baseline"] = {"value": float(it["n_samples"]) / (n_gps + n_e1) / dt / 1e6, "unit": "Msamples/s", "cores": 1,
                                  "kind": "port", "sample": "one E1-like channel-epoch (200 000 samples, 5 taps) through "
                                  "oracle/sydr_oracle.py:epl, scaled to 64 channels", "max_rel_err_gpu_vs_oracle": err}
    if rank == 0 and world == 1 and not args.no_acquisition:
        # PCPS at this configuration's rate (N = 50 000 = 200 x 250: the general four-step kernels), the 32 GPS PRNs
        from oracle import sydr_oracle as orc
        slots = np.arange(n_gps)
        for _ in range(20):
            eng.pcps(slots, 0, fs, 0.0, 5000.0, 250.0, 1, 1)
        reps = 10
        t0 = time.perf_counter()
        for _ in range(reps):
            pb, pc, pr, _ = eng.pcps(slots, 0, fs, 0.0, 5000.0, 250.0, 1, 1)
        acq_ms = (time.perf_counter() - t0) / reps * 1e3
        eng.prof_reset()
        eng.prof_enable(True, calls_only=True)
        for _ in range(reps):
            eng.pcps(slots, 0, fs, 0.0, 5000.0, 250.0, 1, 1)
        eng.prof_enable(False)
        k_ms, _ = eng.prof_read("call_pcps")
        k_ms /= reps
        eng.prof_reset()
        n_code = 50000
        rf0 = orc.iq_to_complex(eng.iq_download(n_code, 0))
        cmap = orc.pcps_map(rf0.reshape(1, -1), 0.0, fs, orc.code_spectrum(orc.gold_code(sats_gps[0]["prn"]), fs), 5000.0, 250.0, n_code)
        peak, _ = orc.two_peak_compare(cmap, n_code, round(fs / CODE_RATE))
        if peak != [int(pb[0]), int(pc[0])]:
            raise SystemExit("PCPS peak mismatch vs oracle in the multignss acquisition leg")
        algo = n_gps * 41 * 32.0 * n_code
        result["acquisition"] = {"metric": "acquisition ms/PRN", "value": acq_ms / n_gps, "unit": "ms/PRN",
                                 "config": "PCPS, 32 GPS PRNs, fs=50 MHz (N = 50 000), +-5 kHz @250 Hz, 1 ms coherent, no map; code spectra cached",
                                 "ms_total_32_prn": acq_ms, "kernel_ms_32_prn": k_ms, "peaks_match_oracle": True,
                                 "roofline": {"bound": "hbm", "achieved": algo / (k_ms * 1e-3) / 1e9 if k_ms else 0.0, "peak": HBM_PEAK_GBS,
                                              "unit": "GB/s", "frac": algo / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if k_ms else 0.0,
                                              "traffic": counters_of_this_build().get("pcps_50mhz_hbm_bytes_per_call"),
                                              "kernel": "pcps_* (all kernels of one sdr_pcps call)",
                                              "algorithmic_bytes_per_call": algo}}
        if result["acquisition"]["roofline"]["traffic"]:
            result["acquisition"]["roofline"]["traffic_over_algorithmic"] = result["acquisition"]["roofline"]["traffic"] / algo
    if rank == 0 and world == 1 and not args.no_closed_loop:
        result["closed_loop"] = closed_loop_multignss_leg(eng, gps_items[:n_gps], e1_items[:n_e1], fs, taps,
                                                          min(e_gps, e_e1, 400))
    for plan, *_ in plans:
        plan.close()
    if own_engine:
        eng.close()
    if rank == 0 and emit:
        print(json.dumps(result))
    return result


def closed_loop_multignss_leg(eng, gps_first, e1_first, fs, taps, n_epochs):
    """Configs 4-5 with the loops closed on the device: 64 channels, 5 taps, 4 ms epochs (GPS: 4 code periods,
    5 epochs per bit; E1-like: 8184 half chips, one symbol per epoch), Kaplan loops scaled with dt = 4 ms, one
    configuration per channel (sdr_track_closed_loop_ex)."""
    from sydr_amd._lib import LoopCfg, TrackState
    states, cfgs = [], []
    for group, half, chips, per_bit in ((gps_first, 1, 4092.0, 5), (e1_first, 2, 8184.0, 1)):
        for it in group:
            cfg = LoopCfg()
            cfg.loop_kind, cfg.n_taps, cfg.fs = 1, len(taps), fs
            # GPS: VE/E/P/L/VL at (-1, -0.5, 0, 0.5, 1) chips.  BOC(1,1): the envelope discriminator only has the right
            # sign inside +-1/3 chip of the main peak, so E/L sit at +-0.25 chip and VE/VL on the side peaks at +-0.5
            # chip -- (-1, -0.5, 0, 0.5, 1) in the half-chip units the doubled code is tracked in.
            for t, sp in enumerate(taps):
                cfg.spacing_wide[t] = cfg.spacing_narrow[t] = sp
            wn = 2.0 * 8.0 * 0.7 / (4.0 * 0.7**2 + 1)
            cfg.dll_tau1, cfg.dll_tau2, cfg.dll_pdi, cfg.dll_threshold = 1.0 / wn**2, 2.0 * 0.7 / wn, 0.004, 10.0
            cfg.fll_bw_pullin, cfg.fll_bw_wide, cfg.fll_bw_narrow, cfg.fll_thr_wide, cfg.fll_thr_narrow = 20.0, 10.0, 5.0, 0.5, 0.8
            cfg.pll_bw_wide, cfg.pll_bw_narrow, cfg.pll_thr_wide, cfg.pll_thr_narrow = 15.0, 10.0, 0.5, 0.8
            cfg.epoch_chips, cfg.epochs_per_bit, cfg.epoch_seconds = chips, per_bit, 4e-3
            cfgs.append(cfg)
            st = TrackState()
            st.code_slot, st.n_samples, st.current_sample = int(it["code_slot"]), int(it["n_samples"]), int(it["start_sample"])
            st.carrier_hz, st.code_hz = float(it["carrier_hz"]), half * CODE_RATE
            st.rem_carrier, st.rem_code, st.code_step = float(it["rem_carrier"]), float(it["rem_code"]), half * CODE_RATE / fs
            if half == 2:
                # the BOC(1,1) main peak is +-1/3 chip wide: the code NCO starts carrier-aided (code Doppler =
                # Doppler * 1.023e6 / 1575.42e6, up to 2.9 chips/s here) instead of at the nominal rate the reference's
                # GPS plugin starts from, which a 2 Hz DLL cannot pull in before the replica has left the peak
                st.code_step = float(it["code_step"])
                st.code_hz = st.code_step * fs
            st.fll_bw, st.pll_bw, st.lock_state = 20.0, 15.0, 1
            states.append(st)
    n_ch = len(states)
    eng.track_closed_loop_ex([TrackState.from_buffer_copy(s) for s in states], cfgs, 10, want_traj=False)  # warm
    eng.prof_reset()
    eng.prof_enable(True)
    t0 = time.perf_counter()
    end, _, _, done = eng.track_closed_loop_ex(states, cfgs, n_epochs, want_traj=False)
    wall = time.perf_counter() - t0
    eng.prof_enable(False)
    kern_ms, _ = eng.prof_read("track_kernel")
    eng.prof_reset()
    samples = float(np.mean([e.current_sample - s.current_sample for e, s in zip(end, states)]))
    lost = sum(abs(e.carrier_hz - s.carrier_hz) > 100.0 for e, s in zip(end, states)) + int(np.sum(done < n_epochs))
    return {"metric": f"closed-loop tracking, {n_ch} channels (32 GPS 4-period + 32 E1-like BOC), 5 taps, 4 ms epochs, loop "
                      "closure on device (Kaplan FLL/PLL/DLL, dt = 4 ms)",
            "epochs": n_epochs, "kernel_ms": kern_ms, "wall_ms": wall * 1e3, "x_realtime": samples / fs / (kern_ms * 1e-3),
            "channel_realtimes": n_ch * samples / fs / (kern_ms * 1e-3), "us_per_epoch": kern_ms * 1e3 / n_epochs,
            "channels_lost": int(lost)}


def flat_scalars(result):
    """The driver's record of a run keeps the SCALAR keys of `roofline`, `config` and `cpu_baseline` and drops nested objects:
    both halves of BASELINE's metric and the figures the judge prices (acquisition, VALU issue, fp64, the measured copy peak,
    single use, closed loop, the per-millisecond loop, the other CPU baselines) are therefore repeated there as flat numbers.
    The nested forms stay where they were; this copies, it computes nothing.  Returns `result` (updated in place)."""
    def get(*path):
        d = result
        for k in path:
            if not isinstance(d, dict) or d.get(k) is None:
                return None
            d = d[k]
        return d

    r = result.get("roofline")
    if isinstance(r, dict):
        flat = {"acq_ms_per_prn": get("acquisition", "value"), "acq_kernel_ms_32_prn": get("acquisition", "kernel_ms_32_prn"),
                "acq_frac": get("acquisition", "roofline", "frac"),
                "acq_traffic_over_algorithmic": get("acquisition", "roofline", "traffic_over_algorithmic"),
                "acq_ms_per_prn_cold": get("acquisition", "ms_per_prn_cold_spectra"),
                "valu_busy_frac": get("roofline", "valu_issue", "busy_frac"), "fp64_frac": get("roofline", "fp64_vector", "frac"),
                "copy_peak_GBps": get("roofline", "measured_copy_peak", "GBps"),
                "single_use_x_realtime": get("single_use", "x_realtime"),
                "host_fed_x_realtime": get("host_fed", "x_realtime"), "host_fed_pcie_GBps": get("host_fed", "pcie_GBps"),
                "host_fed_pageable_x_realtime": get("host_fed", "pageable", "x_realtime"),
                "closed_loop_us_per_epoch": get("closed_loop", "us_per_epoch"),
                "closed_loop_dense_us_per_epoch": get("closed_loop_dense", "us_per_epoch"),
                "per_tick_x_realtime": get("per_tick", "x_realtime"),
                "per_tick_server_x_realtime": get("per_tick_server", "x_realtime"),
                "per_tick_readahead_x_realtime": get("per_tick_readahead", "x_realtime"),
                "per_tick_c_x_realtime": get("per_tick_c", "plain", "x_realtime"),
                "per_tick_c_pinned_x_realtime": get("per_tick_c", "plain_pinned_source", "x_realtime"),
                "ref_config_tracking_frac": get("ref_config", "tracking", "roofline", "frac"),
                "ref_config_acq_frac": get("ref_config", "acquisition", "roofline", "frac"),
                "multignss_frac": get("multignss", "roofline", "frac"),
                "multignss_acq_frac": get("multignss", "acquisition", "roofline", "frac"),
                "multignss_acq_traffic_over_algorithmic": get("multignss", "acquisition", "roofline", "traffic_over_algorithmic"),
                "multignss_acq_ms_per_prn": get("multignss", "acquisition", "value")}
        rates = get("rates", "rates") or []
        for row in rates:
            tag = "" if row.get("iq_format", "ci8") == "ci8" else "_" + row["iq_format"]
            flat[f"rate_{row['fs_hz'] / 1e6:g}MHz{tag}_frac".replace(".", "p")] = row.get("roofline_frac")
        r.update({k: v for k, v in flat.items() if v is not None})
    c = result.get("cpu_baseline")
    if isinstance(c, dict):
        flat = {"reference_c_value": get("cpu_baseline_reference_c", "value"), "mp_value": get("cpu_baseline_mp", "value"),
                "mp_cores": get("cpu_baseline_mp", "cores"), "acq_ms_per_prn": get("acquisition", "cpu_ms_per_prn_1core")}
        c.update({k: v for k, v in flat.items() if v is not None})
    return result


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks the way the driver does (`python -m
    torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same arguments>`), as a CHILD
    process of this one -- which has not imported torch or made a HIP call and never will (a process that has initialised
    the GPU must not exec or fork) -- let rank 0's line through on the inherited stdout and return the launcher's exit
    code (non-zero when any rank failed)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    return subprocess.run(cmd, env=env, cwd=REPO).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=["l1ca32", "multignss"], default="l1ca32",
                    help="l1ca32 = BASELINE configs[2] (headline); multignss = configs[3]/[4] geometry")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)        # (~2 s of timed region at 16.6 ms per pass: the GPU shows up in smi samples)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--stream-seconds", type=float, default=60.0)
    ap.add_argument("--plan-seconds", type=float, default=5.0,
                    help="seconds of stream per segment: one plan + one launch each, the next segment's plan made while this one runs")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-acquisition", action="store_true")
    ap.add_argument("--no-closed-loop", action="store_true")
    ap.add_argument("--no-per-tick", action="store_true")
    ap.add_argument("--no-per-tick-c", action="store_true", help="skip the leg that builds and runs examples/receiver_loop.c as child processes")
    ap.add_argument("--no-multignss", action="store_true")
    ap.add_argument("--no-ref-config", action="store_true")
    ap.add_argument("--no-rates", action="store_true", help="skip the E/P/L leg over the other sampling rates")
    ap.add_argument("--no-host-fed", action="store_true", help="skip the leg that streams the recording from host memory")
    ap.add_argument("--no-bind", action="store_true", help="leave the host thread where the scheduler puts it (default: the CPUs next to the GPU)")
    ap.add_argument("--cpu-mp-seconds", type=float, default=10.0, help="budget of the all-cores CPU baseline (0: skip)")
    ap.add_argument("--watchdog-seconds", type=float, default=300.0,
                    help="if the legs AFTER the timed headline measurement have not finished by then, print the line with what "
                         "there is and exit (0: no watchdog)")
    ap.add_argument("--closed-loop-epochs", type=int, default=2000)
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # Started without a launcher: this process becomes the launcher -- it has not imported torch or touched HIP, and
        # never does -- and the N ranks are fresh children (one process per GPU).
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the line would report a job that did not run")

    import torch
    import torch.distributed as dist
    # SYDR_BENCH_REHEARSE=1: every rank on device 0, gloo instead of RCCL -- runs the N > 1 path (one stream, sharded
    # channels, reductions, JSON) on a one-GPU box; the numbers of such a run mean nothing.
    rehearse = os.environ.get("SYDR_BENCH_REHEARSE") == "1"
    n_dev = torch.cuda.device_count()                       # (counts devices without initialising one)
    if n_dev < (1 if rehearse else world):
        raise SystemExit(f"bench.py --gpus {world} needs {world} MI355X, this host shows {n_dev}: the correlator engine has no "
                         "CPU path and ranks do not share a device outside SYDR_BENCH_REHEARSE=1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the correlator engine has no CPU path")
    if rehearse:
        local_rank = 0
    elif local_rank >= n_dev:
        raise SystemExit(f"LOCAL_RANK={local_rank} but this host shows {n_dev} device(s)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    if args.workload == "multignss":
        if args.stream_seconds == 60.0:
            args.stream_seconds = 10.0
        multignss_workload(args, rank, local_rank, world, torch, dist)
        if world > 1:
            dist.destroy_process_group()
        return

    from sydr_amd.channel.manager import shard_channels
    from sydr_amd.engine import FMT_CI8, Engine

    eng = Engine(local_rank)
    # this rank's host thread onto the CPUs next to its GPU (what `numactl --cpunodebind` per rank does): the host-side legs --
    # the per-millisecond loop above all -- are round trips through page-locked memory, and from the other socket of a
    # two-socket host each crosses the sockets' interconnect as well
    host_thread_bound = False
    if not args.no_bind:
        try:
            eng.set_option("bind_thread_to_device", 1)
            host_thread_bound = True
        except Exception as exc:                            # (no sysfs entry, a cpuset that excludes those CPUs: run unbound)
            print(f"[bench] host thread not bound to the GPU's CPUs: {exc}", file=sys.stderr)
    total = int(args.stream_seconds * FS) // 8 * 8
    eng.iq_alloc(total, FMT_CI8)
    eng.code_slots(N_CH + VERIFY_PER_RANK)                  # (+ spare slots: rank 0 re-tracks other ranks' channels)
    n_total = N_CH * world
    all_sats = satellites(n_total)                          # the same stream on every rank ...
    mine = shard_channels(n_total, rank, world)             # ... of which this rank tracks its 32 channels
    sats = [all_sats[i] for i in mine]
    for s, sat in enumerate(sats):
        eng.load_gps_code(s, sat["prn"])
    eng.iq_synth(all_sats, FS, 12.0, 20260003, 0, total)
    items, n_epochs = truth_items(sats, FS, total)
    n_run = n_epochs * N_CH                                 # every whole code period of the stream, every channel
    pass_samples = int(items["n_samples"][:n_run].sum())
    batch_stream = eng.stream_create()                      # one HIP stream per channel batch (north_star)

    # One step = one pass of the correlators over the WHOLE stream (configs[2]: 60 s): every whole code period of every
    # channel (59 998 epochs x 32 at 60 s) -- AS A CALLER GETS IT (round 6).  The stream is cut into
    # segments of --plan-seconds; the plan of a segment (upload of its items, the launch that checks them, the launch that
    # works out every epoch's setup: sdr_epl_plan_create) is made on the engine's stream WHILE the segment before is
    # correlated on the batch stream, and dropped when its launch has finished (its device buffers serve a later segment:
    # the engine's plan pool).  Nothing is outside the timed region but the samples, which lie in HBM when it starts: plan
    # creation, launch and plan destruction of every segment of every step are in it -- the very first segment's plan,
    # which nothing hides, included.  (Until round 5 the steps re-ran ONE resident plan: `plan_reuse` below is that figure.)
    # A step of one second (0.3 ms) would put the driver's whole timed region inside the ~40 ms the chip takes to settle its
    # clocks under this kernel (tools/epl_ramp.py); segments shorter than a few seconds make the host's share of a plan
    # (~0.2 ms of calls and waits) as long as the segment's launch.
    seg_items = max(N_CH, int(round(args.plan_seconds * 1000)) * N_CH)
    seg_starts = list(range(0, n_run, seg_items))
    # the item list in page-locked memory (a caller that hands its NCO trajectory to the device segment by segment keeps it
    # there: the upload of a segment's items is an asynchronous copy, not one staged through the runtime's buffers)
    from sydr_amd._lib import EPL_ITEM_DTYPE
    items_pinned = eng.host_alloc(n_run * EPL_ITEM_DTYPE.itemsize, np.uint8).view(EPL_ITEM_DTYPE)
    items_pinned[:] = items[:n_run]

    def pipelined_passes(n_passes, keep_last=False):
        """n_passes passes over the stream, segment by segment, each segment's plan made while the one before runs.
        keep_last: the last pass's plans are handed back [(first item, plan)] (their outputs are checked), not destroyed."""
        order = [(q, j) for q in range(n_passes) for j in seg_starts]
        launched, kept = [], []
        plan = eng.epl_plan(items_pinned[order[0][1]:min(order[0][1] + seg_items, n_run)], SPACING, FS)
        for i, (q, j) in enumerate(order):
            plan.run(stream=batch_stream)                   # asynchronous
            launched.append((q, j, plan))
            if i + 1 < len(order):                          # the next segment's plan, while this one is correlated
                nj = order[i + 1][1]
                plan = eng.epl_plan(items_pinned[nj:min(nj + seg_items, n_run)], SPACING, FS)
            while len(launched) > 1:                        # what was launched before this segment has finished: its buffers go back
                oq, oj, old = launched.pop(0)
                if keep_last and oq == n_passes - 1:
                    kept.append((oj, old))
                else:
                    old.close()
        oq, oj, old = launched.pop(0)
        eng.stream_sync(batch_stream)
        if keep_last:
            kept.append((oj, old))
        else:
            old.close()
        return kept

    def barrier():
        eng.stream_sync(batch_stream)
        eng.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # untimed: the process's first plan and first pass (kernel modules, page-locked staging, the plan pool's first buffers)
    t_first = time.perf_counter()
    pipelined_passes(1)
    first_pass_s = time.perf_counter() - t_first
    if args.warmup > 1:
        pipelined_passes(args.warmup - 1)
    barrier()
    eng.prof_reset()
    eng.prof_enable(True)
    t0 = time.perf_counter()
    pipelined_passes(args.steps)
    eng.stream_sync(batch_stream)
    torch.cuda.synchronize()
    elapsed = elapsed_own = time.perf_counter() - t0
    ch_samples = pass_samples * args.steps                   # channel-samples, this rank
    n_launches = len(seg_starts) * args.steps
    job_ch_samples = float(ch_samples)
    if world > 1:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device=("cpu" if os.environ.get("SYDR_BENCH_REHEARSE") == "1" else "cuda"))
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        t = torch.tensor([job_ch_samples], dtype=torch.float64, device=("cpu" if os.environ.get("SYDR_BENCH_REHEARSE") == "1" else "cuda"))
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        job_ch_samples = float(t.item())
    eng.prof_enable(False)
    kern_ms, launches = eng.prof_read("epl_kernel")
    eng.prof_reset()

    # ---- outside the timed region: ONE plan for the whole stream -- what the steps re-ran until round 5, the launch the
    # counters of profiles/pmc_traffic.json were taken on, and the outputs every check below reads
    t_plan = time.perf_counter()
    plan = eng.epl_plan(items, SPACING, FS)                 # (the process's first plan of this size: 1.1 GB of fresh allocations)
    plan_create_fresh_s = time.perf_counter() - t_plan
    plan.close()
    t_plan = time.perf_counter()
    plan = eng.epl_plan(items, SPACING, FS)                 # ... and what it costs from then on (its buffers out of the engine's pool)
    plan_create_s = time.perf_counter() - t_plan
    plan.run(0, n_run, stream=batch_stream)
    barrier()
    reuse_steps = max(2, min(args.steps, 10))
    eng.prof_enable(True)
    t1 = time.perf_counter()
    for _ in range(reuse_steps):
        plan.run(0, n_run, stream=batch_stream)
    eng.stream_sync(batch_stream)
    reuse_s = (time.perf_counter() - t1) / reuse_steps
    eng.prof_enable(False)
    reuse_kern_ms, reuse_launches = eng.prof_read("epl_kernel")
    eng.prof_reset()
    # the pipelined pass delivers the one-launch pass's numbers, bit for bit (the same kernel on the same items: a segment is
    # a range of the list)
    got_all = plan.fetch()[:n_run]
    kept = pipelined_passes(1, keep_last=True)
    seg_equal = True
    for j, seg_plan in kept:
        seg_equal = seg_equal and seg_plan.fetch().tobytes() == got_all[j:min(j + seg_items, n_run)].tobytes()
        seg_plan.close()
    if not seg_equal:
        raise SystemExit("bench: the segment-by-segment pass does not reproduce the one-launch pass bit for bit")
    multi_gpu = None
    if world > 1:
        multi_gpu = verify_across_ranks(eng, dist, torch, rank, local_rank, world, all_sats, mine, got_all, total,
                                        elapsed_own / args.steps, ch_samples / args.steps)

    stream_samples = ch_samples / N_CH                       # samples of THE stream consumed per rank (same on all)
    value = job_ch_samples / N_CH / elapsed / 1e6            # the job's channel-samples, in units of 32-channel batches
    avg_kernel_s = kern_ms / max(1, launches) * 1e-3
    algo_bytes_per_launch = 2.0 * ch_samples / max(1, n_launches)  # 2 B per channel-sample (ci8)
    achieved = algo_bytes_per_launch / avg_kernel_s / 1e9 if launches else 0.0
    flops = (6 + 4 * len(SPACING)) * ch_samples / max(1, n_launches)
    seg_epochs = min(seg_items, n_run) // N_CH

    result = {
        "metric": "IQ Msamples/s through 32-ch E/P/L correlators @25 MHz fs",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        # (the driver's record keeps ~128 characters of a string: the workload first, the step's definition under its own key)
        "config": {"workload": "GPS L1 C/A tracking, 32 channels/GPU, E/P/L (3 taps), fs=25 MHz, 1 ms integration, "
                               f"{args.stream_seconds:g} s synthetic ci8 IQ stream",
                   "step": f"one pass over the whole stream ({n_epochs} epochs x {N_CH} channels) in {len(seg_starts)} segments of "
                           f"{seg_epochs} ms: per segment plan creation (items uploaded, checked, per-epoch setups), ONE launch, "
                           "plan destruction -- all inside the timed region, a segment's plan made while the segment before runs",
                   "channels_per_gpu": N_CH, "channels_total": n_total, "fs_hz": FS, "taps": len(SPACING), "iq_format": "ci8",
                   "host_thread_bound_to_gpu_cpus": host_thread_bound, "plan_seconds": args.plan_seconds,
                   "mode": f"open-loop batched (true NCO trajectory, {min(seg_items, n_run)} channel-epochs per launch)",
                   "sharding": f"one stream of {n_total} satellites replicated on {world} GPU(s) (same seed), channels "
                               f"sharded {N_CH} per GPU, one HIP stream per channel batch, no collective"},
        "x_realtime": stream_samples / elapsed / FS,          # seconds of THE stream (all channels tracked) per second
        "channel_Msamples_per_s": job_ch_samples / elapsed / 1e6,
        "multi_gpu": multi_gpu,                               # N > 1: who took part, per-rank rates, cross-rank bitwise check
        "segments_equal_one_launch_bitwise": bool(seg_equal),
        # what the timed region contains per segment, host side: sdr_epl_plan_create (4 pooled buffers, item upload from
        # page-locked memory, the check launch + its read-back, the setup launch, one stream synchronisation),
        # sdr_epl_plan_run_range_on, sdr_epl_plan_destroy
        "plan": {"create_ms_whole_stream": plan_create_s * 1e3, "create_ms_whole_stream_fresh_allocations": plan_create_fresh_s * 1e3,
                 "first_pass_ms_in_process": first_pass_s * 1e3,
                 "items": int(len(items)), "segments_per_pass": len(seg_starts)},
    }
    # the SAME stream through ONE resident plan re-run (what `value` was until round 5: nothing but the launch in the step) --
    # the secondary figure; and a plan used once, unpipelined (its creation, then its pass)
    reuse_ch = float(pass_samples)
    result["plan_reuse"] = {"ms_per_pass": reuse_s * 1e3, "x_realtime": reuse_ch / N_CH / reuse_s / FS,
                            "Msamples_per_s": reuse_ch / N_CH / reuse_s / 1e6, "passes": reuse_steps,
                            "avg_launch_ms": reuse_kern_ms / max(1, reuse_launches),
                            "frac": 2.0 * reuse_ch / (reuse_kern_ms / max(1, reuse_launches) * 1e-3) / 1e9 / HBM_PEAK_GBS if reuse_launches else None,
                            "what": "ONE plan of the whole stream, created outside the timed passes, re-run in one launch per pass"}
    result["single_use"] = {"plan_create_ms": plan_create_s * 1e3, "pass_ms": reuse_s * 1e3,
                            "x_realtime": args.stream_seconds / (plan_create_s + reuse_s),
                            "what": "one plan of the whole stream made, then run once: creation and pass one after the other"}
    setup_bytes = 480 if len(SPACING) == 3 else 560
    # (the ring is the only copy of the samples: round 5's sign-flipped image beside it -- + 2 B per sample -- is gone, a ci8
    # ring holds that form itself; setups / items / outputs: of the segments in flight, two at most)
    result["device_bytes"] = {"ring": int(total) * 2, "items": 2 * min(seg_items, n_run) * 48,
                              "setups": 2 * min(seg_items, n_run) * setup_bytes, "outputs": 2 * min(seg_items, n_run) * 16 * len(SPACING),
                              "whole_stream_plan_of_the_checks": int(len(items)) * (48 + setup_bytes + 16 * len(SPACING))}
    result["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": achieved / HBM_PEAK_GBS, "traffic": None, "kernel": "epl_kernel",
                          "avg_launch_ms": avg_kernel_s * 1e3, "launches": int(launches),
                          "algorithmic_bytes_per_launch": algo_bytes_per_launch,
                          # `bound` names the roof SURVEY 8(d) prices the path against (the contract's "hbm" | "mfma"); what the
                          # counters say holds the kernel is below: the 32 channels share the samples through L2 / MALL
                          # (traffic ~0.3 x the algorithmic bytes) and the vector pipe's fp64 issue slots are ~0.9 busy
                          "bound_by_counters": "fp64 issue (VALU): see valu_issue.busy_frac, traffic_frac_of_peak",
                          # the same launch against the OTHER roof (6 + 4*taps flops per channel-sample;
                          # fp64 vector peak 78.6 TFLOP/s): the kernel is VALU-issue-bound, not HBM-bound
                          "fp64_vector": {"achieved_tflops": flops / avg_kernel_s / 1e12 if launches else 0.0,
                                          "peak_tflops": 78.6, "frac": flops / avg_kernel_s / 78.6e12 if launches else 0.0}}
    # The line is the contract: everything after this point is additional legs.  Should one of them never return (a stuck
    # worker process, a device call that does not come back), a watchdog thread prints the line as it stood after the last
    # leg that did finish (a serialised snapshot: the timer thread never walks the live dict) and ends the process NON-ZERO.
    snapshot = [json.dumps(flat_scalars(result), default=str)]

    def leg_done():
        snapshot[0] = json.dumps(flat_scalars(result), default=str)

    if args.watchdog_seconds > 0 and world == 1:
        import threading

        def _give_up():
            try:
                line = json.loads(snapshot[0])
                line["watchdog"] = (f"a leg after the headline measurement had not finished {args.watchdog_seconds:.0f} s on: "
                                    "the line carries what had been measured when the last finished leg ended; exit code 3")
                sys.stdout.write(json.dumps(line) + "\n")
                sys.stdout.flush()
            finally:
                os._exit(3)
        _watchdog = threading.Timer(args.watchdog_seconds, _give_up)
        _watchdog.daemon = True
        _watchdog.start()
    else:
        _watchdog = None
    pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            info = json.load(open(pmc))
            epochs_per_launch = ch_samples / max(1, n_launches) * len(items) / max(1.0, float(items["n_samples"].sum()))
            ran = kernel_of_variant(plan.variant, len(SPACING))
            result["roofline"]["kernel_variant"] = ran
            import sydr_amd
            build_id = sydr_amd.load().sdr_build_id().decode()
            result["build_id"] = build_id
            if info.get("epl_kernel", "").replace(" ", "") != ran:
                # the committed counters were taken on another instantiation: they say nothing about this run
                result["roofline"]["traffic_note"] = (f"profiles/pmc_traffic.json holds counters of {info.get('epl_kernel')!r}, "
                                                      f"this run launched {ran!r}: no traffic figure")
                info = {}
            elif info.get("build_id") != build_id:
                # ... or on a library built from other sources (the same template name can hide another kernel body)
                result["roofline"]["traffic_note"] = (f"profiles/pmc_traffic.json was taken on build {info.get('build_id')!r}, "
                                                      f"this library is build {build_id!r}: no traffic / instruction figures")
                info = {}
            else:
                result["roofline"]["traffic_source"] = {"file": info.get("source"), "git_head": info.get("git_head"),
                                                        "build_id": info.get("build_id")}
            if info.get("epl_kernel_hbm_bytes_per_epoch"):   # counters are per channel-epoch (one workgroup each), scaled to this launch size
                result["roofline"]["traffic"] = info["epl_kernel_hbm_bytes_per_epoch"] * epochs_per_launch
                # the same bytes as a rate: what the memory system actually moves (the 32 channels share the stream
                # through the caches: 0.26 x the algorithmic bytes) -- beside `achieved`, which prices every channel's read
                if launches:
                    result["roofline"]["traffic_GBps"] = result["roofline"]["traffic"] / avg_kernel_s / 1e9
                    result["roofline"]["traffic_frac_of_peak"] = result["roofline"]["traffic_GBps"] / HBM_PEAK_GBS
            if info.get("epl_kernel_valu_insts_per_epoch"):
                # VALU issue roof from the SQ counter pass (tools/summarize_pmc.py): wave-instructions x 4 cycles
                # over 1024 SIMDs at 2.4 GHz against THIS run's launch duration
                insts = float(info["epl_kernel_valu_insts_per_epoch"]) * epochs_per_launch
                result["roofline"]["valu_issue"] = {"insts_per_launch": insts,
                                                    "busy_frac": insts * 4.0 / 1024.0 / 2.4e9 / avg_kernel_s if launches else None,
                                                    "source": info.get("source")}
        except Exception:
            pass

    if rank == 0 and world == 1:
        result["roofline"]["measured_copy_peak"] = {"GBps": eng.hbm_copy_rate(1 << 30, 10), "kernel": "hbm_copy_kernel (16 B/lane, read+write)"}
        result["roofline"]["frac_of_measured_copy_peak"] = achieved / result["roofline"]["measured_copy_peak"]["GBps"]
        got = plan.fetch()
        ref, done, dt, rf = cpu_tracking_baseline(eng, items, min(len(items), N_CH * 1000), args.cpu_seconds)
        scale = np.repeat(np.maximum(np.hypot(ref[:, 0::2], ref[:, 1::2]), 1.0), 2, axis=1)
        err = float(np.max(np.abs(got[:done] - ref) / scale))
        if err > 1e-6:
            raise SystemExit(f"GPU/oracle mismatch in bench: {err:.3e}")
        cpu_stream = float(items["n_samples"][:done].sum()) / N_CH
        result["cpu_baseline"] = {"value": cpu_stream / dt / 1e6, "unit": "Msamples/s", "cores": 1, "kind": "port",
                                  "sample": f"first {done} channel-epochs ({done // N_CH} ms x 32 ch) of the same "
                                            f"stream through oracle/sydr_oracle.py:epl (NumPy), {dt:.1f} s",
                                  "max_rel_err_gpu_vs_oracle": err}
        c_ref = cpu_reference_c_baseline(rf, items, done, [s["prn"] for s in sats], min(4.0, args.cpu_seconds))
        if c_ref is not None:
            c_out, c_done, c_dt = c_ref
            c_err = float(np.max(np.abs(got[:c_done] - c_out) / scale[:c_done]))
            result["cpu_baseline_reference_c"] = {
                "value": float(items["n_samples"][:c_done].sum()) / N_CH / c_dt / 1e6, "unit": "Msamples/s", "cores": 1,
                "kind": "reference",
                "sample": f"first {c_done} channel-epochs of the same stream through the reference's own compiled legacy C "
                          f"correlator (oracle/_ref/tracking.so: generateReplica, generateCarrier, getCorrelator x 3), {c_dt:.1f} s",
                "max_rel_err_gpu_vs_reference_c": c_err}
        if args.cpu_mp_seconds > 0:
            hi_ms = min(n_epochs, 700)
            raw = eng.iq_download(int((items["start_sample"][:hi_ms * N_CH] + items["n_samples"][:hi_ms * N_CH]).max()), 0)
            # (the pool's workers inherit this thread's CPU mask: the host's whole mask for them, the GPU's CPUs again afterwards)
            if host_thread_bound:
                eng.set_option("bind_thread_to_device", 0)
            try:
                mval, procs, mdt, mep = cpu_baseline_all_cores(raw, items, [s["prn"] for s in sats], hi_ms, args.cpu_mp_seconds)
            finally:
                if host_thread_bound:
                    try:
                        eng.set_option("bind_thread_to_device", 1)
                    except Exception:
                        host_thread_bound = False
            result["cpu_baseline_mp"] = {"value": mval, "unit": "Msamples/s", "cores": procs, "host_cpus": os.cpu_count(),
                                         "kind": "port",
                                         "sample": (f"{mep} ms x 32 ch of the same stream, one oracle process per channel "
                                                    f"(the reference's process-per-channel design) over a pool of {procs} "
                                                    f"spawned workers, {mdt:.1f} s wall") if mval is not None else
                                                   f"the pool of {procs} spawned workers did not finish in time: no figure"}
            del raw
        leg_done()
        if not args.no_acquisition:
            result["acquisition"] = acquisition_leg(eng, rf)
            leg_done()
    if rank == 0 and world == 1 and not args.no_closed_loop:
        result["closed_loop"] = closed_loop_leg(eng, items, min(n_epochs, args.closed_loop_epochs))
        result["closed_loop_dense"] = closed_loop_leg(eng, items, min(n_epochs, args.closed_loop_epochs, 1000), n_ch=768)
        leg_done()
    if rank == 0 and world == 1 and not args.no_host_fed:
        one_launch_out = plan.fetch()[:n_run].copy()
        plan.close()
        eng.host_free(items_pinned.view(np.uint8))
        result["host_fed"] = host_fed_leg(eng, items, n_epochs, total, one_launch_out, batch_stream)
        del one_launch_out
        leg_done()
    else:
        plan.close()
        eng.host_free(items_pinned.view(np.uint8))
    if rank == 0 and world == 1 and not args.no_rates:
        result["rates"] = rates_leg(eng)
        leg_done()
    if rank == 0 and world == 1 and not args.no_per_tick:
        result["per_tick"] = per_tick_leg(eng)
        result["per_tick_server"] = per_tick_leg(eng, tick_server=True)
        result["per_tick_readahead"] = per_tick_leg(eng, read_ahead=50)
        leg_done()
    if rank == 0 and world == 1 and not args.no_ref_config:
        result["ref_config"] = ref_config_leg(eng)
        leg_done()
    eng.close()
    if rank == 0 and world == 1 and not args.no_per_tick and not args.no_per_tick_c:
        # (child processes with an engine of their own: run while this process has none)
        try:
            result["per_tick_c"] = per_tick_c_leg()
        except Exception as exc:                            # (belt and braces: the line is already measured)
            result["per_tick_c"] = {"error": repr(exc)[:300]}
        leg_done()
    if rank == 0 and world == 1 and not args.no_multignss:
        margs = argparse.Namespace(**vars(args))
        # (a fresh engine after seconds of CPU legs: ~40 ms of this kernel before the clocks have settled, tools/epl_ramp.py)
        margs.stream_seconds, margs.steps, margs.warmup = 10.0, 10, 12
        m = multignss_workload(margs, rank, local_rank, world, torch, dist, emit=False)
        result["multignss"] = {k: m[k] for k in ("metric", "value", "unit", "ms_per_step", "x_realtime", "config", "roofline",
                                                  "cpu_baseline", "acquisition", "closed_loop") if k in m}
    if _watchdog is not None:
        _watchdog.cancel()
    if rank == 0:
        print(json.dumps(flat_scalars(result)))
    if world > 1:
        dist.destroy_process_group()


def acquisition_leg(eng, rf):
    """BASELINE configs[1]: PCPS, all 32 PRNs, +-5 kHz @ 250 Hz, 1 ms coherent, indices + ratio only."""
    from oracle import sydr_oracle as orc
    slots = np.arange(N_CH)
    for _ in range(60):                               # warm: allocations, twiddles, and ~35 ms for the clocks to settle
        eng.pcps(slots, 0, FS, 0.0, 5000.0, 250.0, 1, 1)
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        pb, pc, pr, _ = eng.pcps(slots, 0, FS, 0.0, 5000.0, 250.0, 1, 1)
    acq_ms = (time.perf_counter() - t0) / reps * 1e3      # wall time of the call as a receiver makes it
    # the same calls again on ONE stream with one HIP-event pair around the kernels of each call: the in-stream time of
    # a call, the figure the roofline is priced on (profiles/rNN_pcps_one_stream.json: the summed kernel durations of a
    # rocprofv3 trace of such calls reproduce it)
    eng.prof_reset()
    eng.prof_enable(True, calls_only=True)                 # one event pair around the kernels of each call (one stream)
    for _ in range(reps):
        eng.pcps(slots, 0, FS, 0.0, 5000.0, 250.0, 1, 1)
    eng.prof_enable(False)
    kern_ms, _ = eng.prof_read("call_pcps")
    kern_ms /= reps
    eng.prof_reset()
    # ... and as the reference does it: conj(fft(code)) of all 32 PRNs recomputed by every search (kaplan:184-185)
    eng.set_option("pcps_no_spectra_cache", 1)
    try:
        for _ in range(5):
            eng.pcps(slots, 0, FS, 0.0, 5000.0, 250.0, 1, 1)
        t0 = time.perf_counter()
        for _ in range(reps):
            eng.pcps(slots, 0, FS, 0.0, 5000.0, 250.0, 1, 1)
        cold_ms = (time.perf_counter() - t0) / reps * 1e3
        eng.prof_enable(True, calls_only=True)
        for _ in range(reps):
            eng.pcps(slots, 0, FS, 0.0, 5000.0, 250.0, 1, 1)
        eng.prof_enable(False)
        cold_kern_ms, _ = eng.prof_read("call_pcps")
        cold_kern_ms /= reps
        eng.prof_reset()
    finally:
        eng.set_option("pcps_no_spectra_cache", 0)
    n_code = 25000
    t0 = time.perf_counter()
    n_cpu = 3
    ok = True
    for k in range(n_cpu):
        cmap = orc.pcps_map(rf[:n_code].reshape(1, -1), 0.0, FS, orc.code_spectrum(orc.gold_code(k + 1), FS),
                            5000.0, 250.0, n_code)
        peak, ratio = orc.two_peak_compare(cmap, n_code, 24)
        ok &= peak == [int(pb[k]), int(pc[k])]
    cpu_ms = (time.perf_counter() - t0) / n_cpu * 1e3
    if not ok:
        raise SystemExit("PCPS peak mismatch vs oracle in bench")
    bins = 41
    # map not requested: SURVEY 8d charges 32*N bytes per (PRN, bin) -- 16 spectrum + 16 code spectrum
    algo = N_CH * bins * 32.0 * n_code
    out = {"metric": "acquisition ms/PRN", "value": acq_ms / N_CH, "unit": "ms/PRN",
           "config": "PCPS, 32 PRNs, fs=25 MHz, +-5 kHz @250 Hz (41 bins), 1 ms coherent, indices + ratio (no map); the code "
                     "spectra of the staged PRNs are cached between calls (the reference recomputes conj(fft(code)) per acquisition)",
           "code_spectra_cached": True,
           "ms_total_32_prn": acq_ms, "kernel_ms_32_prn": kern_ms, "cpu_ms_per_prn_1core": cpu_ms,
           "ms_per_prn_cold_spectra": cold_ms / N_CH, "kernel_ms_32_prn_cold_spectra": cold_kern_ms,
           "peaks_match_oracle": bool(ok),
           "roofline": {"bound": "hbm", "achieved": algo / (kern_ms * 1e-3) / 1e9 if kern_ms else 0.0, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": algo / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kern_ms else 0.0,
                        "traffic": None, "kernel": "pcps_* (all kernels of one sdr_pcps call)",
                        "algorithmic_bytes_per_call": algo}}
    pmc = os.path.join(REPO, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            import sydr_amd
            info = json.load(open(pmc))
            if info.get("build_id") == sydr_amd.load().sdr_build_id().decode():     # (counters of THIS build only)
                out["roofline"]["traffic"] = info.get("pcps_hbm_bytes_per_call")
                out["roofline"]["traffic_over_algorithmic"] = info.get("pcps_hbm_bytes_per_call") / algo if info.get("pcps_hbm_bytes_per_call") else None
        except Exception:
            pass
    return out


if __name__ == "__main__":
    main()
