"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol the
header declares, struct layouts agree between C, ctypes and NumPy, and -- without a GPU --
the engine refuses to start instead of falling back to a CPU path."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import REPO
from sydr_amd import _lib

HEADER = os.path.join(REPO, "include", "sydr_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/sydr_amd.h but not exported"
    # and the binding covers exactly the declared surface
    assert sorted(_lib.exported_symbols()) == names


def test_dynamic_symbol_table_is_the_header():
    """`nm -D --defined-only`: the header's functions, all of type T, and nothing else (no C++-mangled internals, no kernel
    handles) -- sydr/c_functions/Makefile:1-12's convention of one .so whose bound symbols are its whole surface."""
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    table = [line.split() for line in out.splitlines() if line.strip()]
    exported = sorted(t[-1] for t in table)
    assert exported == declared_symbols(), sorted(set(exported) ^ set(declared_symbols()))
    assert {t[-2] for t in table} == {"T"}, sorted({t[-2] for t in table})


def test_only_gfx950_code_objects():
    out = subprocess.run(["strings", "-a", _lib.LIB_PATH], capture_output=True, text=True).stdout
    targets = set(re.findall(r"amdgcn-amd-amdhsa--(gfx[0-9a-z]+)", out))
    assert targets == {"gfx950"}, targets


def test_struct_layouts():
    assert C.sizeof(_lib.EplItem) == 48 == _lib.EPL_ITEM_DTYPE.itemsize
    for name, (_, off) in _lib.EPL_ITEM_DTYPE.fields.items():
        assert getattr(_lib.EplItem, name).offset == off
    assert C.sizeof(_lib.SynthSat) == 40
    assert C.sizeof(_lib.TrackEpoch) == _lib.TRACK_EPOCH_DTYPE.itemsize
    for name, (_, off) in _lib.TRACK_EPOCH_DTYPE.fields.items():
        assert getattr(_lib.TrackEpoch, name).offset == off
    assert C.sizeof(_lib.TrackState) == 8 + 8 + 15 * 8 + 6 * 4 + 8 + 2 * 4
    assert C.sizeof(_lib.LoopCfg) == 8 + 8 + 2 * 8 * 8 + 16 * 8 + 16 + 8
    assert _lib.TRACK_STATE_DTYPE.itemsize == C.sizeof(_lib.TrackState) and _lib.LOOP_CFG_DTYPE.itemsize == C.sizeof(_lib.LoopCfg)


def test_struct_sizes_agree_with_the_c_compiler(tmp_path):
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "sydr_amd.h"\nint main(void){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
                   "sizeof(sdr_epl_item),sizeof(sdr_synth_sat),sizeof(sdr_track_state),sizeof(sdr_loop_cfg),"
                   "sizeof(sdr_track_epoch),sizeof(sdr_tick_update),sizeof(sdr_tick_mirror),offsetof(sdr_tick_mirror,n_ran),"
                   "offsetof(sdr_tick_mirror,max_unread));return 0;}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)])
    sizes = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert sizes == [C.sizeof(_lib.EplItem), C.sizeof(_lib.SynthSat), C.sizeof(_lib.TrackState),
                     C.sizeof(_lib.LoopCfg), C.sizeof(_lib.TrackEpoch), C.sizeof(_lib.TickUpdate), C.sizeof(_lib.TickMirror),
                     _lib.TickMirror.n_ran.offset, _lib.TickMirror.max_unread.offset]
    assert _lib.TICK_UPDATE_DTYPE.itemsize == C.sizeof(_lib.TickUpdate) == 24


def test_doppler_grid_length_host_helper():
    lib = _lib.load()
    for rng, step in ((5000.0, 250.0), (5000.0, 100.0), (5000.0, 300.0), (7000.0, 125.0), (10.0, 3.0)):
        assert lib.sdr_pcps_bins(rng, step) == len(np.arange(-rng, rng + 1, step))


def test_no_cpu_fallback_without_gpu():
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible here")
    lib = _lib.load()
    h = C.c_void_p()
    rc = lib.sdr_engine_create(0, C.byref(h))
    assert rc == -2 and not h.value
    assert b"no CPU fallback" in lib.sdr_last_error()
    from sydr_amd.engine import Engine
    with pytest.raises(_lib.SdrError):
        Engine(0)


def test_product_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's CPU-baseline / check legs may touch oracle/: not the package,
    not the examples, not the tools (whose oracle-backed stress drivers therefore live under tests/)."""
    for top in ("sydr_amd", "examples", "tools", "include"):
        for root, _, files in os.walk(os.path.join(REPO, top)):
            for f in files:
                if f.endswith((".py", ".hip", ".h", ".c", ".sh")):
                    text = open(os.path.join(root, f)).read()
                    assert "sydr_oracle" not in text and "from oracle" not in text and "import oracle" not in text, f
                    assert "test_host_layer" not in text and "fake_engine" not in text, f   # (they would pull the oracle in)


def _build_c_example(tmp_path, name="acquire_track"):
    exe = tmp_path / name
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(REPO, "include"),
                           os.path.join(REPO, "examples", name + ".c"), "-L", os.path.join(REPO, "sydr_amd"),
                           "-lsydr_amd", "-lm", "-Wl,-rpath," + os.path.join(REPO, "sydr_amd"), "-o", str(exe)])
    return exe


def test_header_is_plain_c_and_a_c_client_links(tmp_path):
    """The boundary is a C-ABI: the header compiles as C99 and a client written in C (examples/acquire_track.c)
    links against the library with nothing but the header."""
    _build_c_example(tmp_path)
    _build_c_example(tmp_path, "receiver_loop")


@pytest.mark.gpu
def test_c_receiver_loop_ticks_through_the_mirrored_call(tmp_path):
    """examples/receiver_loop.c: the reference receiver's per-millisecond loop from plain C -- sdr_iq_upload_begin +
    sdr_bank_tick_mirrored per tick, 32 channels @ 25 MHz from acquisition on: every channel ends on its satellite's
    Doppler, an epoch per channel and tick, navigation bits counted by the call."""
    out = subprocess.check_output([str(_build_c_example(tmp_path, "receiver_loop")), "300"], text=True, timeout=120)
    m = re.search(r"(\d+) ticks, (\d+) epochs, (\d+) navigation bits, (\d+) of 32 channels on their Doppler", out)
    assert m, out
    ticks, epochs, bits, locked = (int(g) for g in m.groups())
    assert ticks == 300 and locked == 32 and epochs >= 32 * 297 and bits >= 20, out
    assert re.search(r"us per tick", out), out
    # the same loop answered by the resident tick server (sdr_set_option "tick_server"): the same channels in the same
    # states (what it prints about them is equal line for line), every tick a request the server answered
    served = subprocess.check_output([str(_build_c_example(tmp_path, "receiver_loop")), "300", "server"], text=True, timeout=120)
    strip = lambda text: [l for l in text.splitlines() if "us per tick" not in l and not l.startswith("tick server:")]
    assert strip(served) == strip(out), served
    m = re.search(r"tick server: (\d+) requests answered, (\d+) server\(s\) started$", served, re.M)
    assert m and int(m.group(1)) >= 290 and int(m.group(2)) == 1, served      # (the server starts after eight steady ticks)


@pytest.mark.gpu
def test_c_client_acquires_and_tracks(tmp_path):
    out = subprocess.check_output([str(_build_c_example(tmp_path))], text=True, timeout=120)
    found = [int(m) for m in re.findall(r"PRN (\d+): bin .* -> track", out)]
    assert found == [1, 3, 6, 8], out                     # the four synthesised satellites, none of the absent ones
    carriers = {int(p): float(f) for p, f in re.findall(r"PRN (\d+): carrier\s+([-+0-9.]+) Hz", out)}
    for prn, doppler in ((1, 1750.0), (3, -2500.0), (6, 4000.0), (8, -750.0)):
        assert abs(carriers[prn] - doppler) < 25.0, out
