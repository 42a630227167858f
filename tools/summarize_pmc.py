#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs (one pass per counter group, as the gfx950 guide
prescribes) into a per-kernel summary and profiles/pmc_traffic.json (read by bench.py).

    python tools/summarize_pmc.py gpurun_out/prof_r02a r02a
HBM bytes per launch = 2 * FETCH_SIZE*1024 (gfx950: FETCH_SIZE reads exactly half of a 16-B/lane
streaming read, MI355X_MICROARCH.md section HBM) + WRITE_SIZE*1024.

Kernel variants are kept apart (`epl_kernel<0, 3, 26, 24>` is the headline launch, `<0, 5, ...>` the
multi-GNSS one); the PCPS figure is the sum over every kernel of one sdr_pcps call of the headline search (dispatches
walked in order, a call ends with its one ratio_kernel dispatch; calls at other rates are left out)."""
import collections
import csv
import glob
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
force = "--force" in sys.argv[3:]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# The counters belong to the library that was profiled (build_id.txt, written by tools/profile_bench.sh from sdr_build_id()):
# if the sources in the tree hash to something else now -- a kernel was edited after the profile was taken -- the summary would
# describe another build than the one bench.py runs, so nothing is written.
sys.path.insert(0, repo)
from sydr_amd import _lib as _sydr_lib
try:
    profiled_build = open(os.path.join(src, "build_id.txt")).read().strip()
except OSError:
    profiled_build = ""
tree_build = _sydr_lib.source_build_id()
if profiled_build != tree_build and not force:
    sys.exit(f"summarize_pmc: the profile in {src} was taken on build {profiled_build or '(unknown)'}, the sources in the tree are "
             f"build {tree_build}: re-take the profile (tools/profile_round.sh) instead of summarising a stale one (--force overrides)")
PCPS = ("fft4_rows_kernel", "fft4_cols_kernel", "fast25k", "fused25k", "fused10k", "fastn::", "mag_acc_kernel", "peak_finish_kernel", "argmax_part_kernel", "fft_pass_kernel", "argmax", "ratio_kernel", "peak_", "second_peak",
        "chirp", "upsample_batch_kernel", "mix_", "twiddle_kernel")


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"((?:fast25k::|fused25k::|fused10k::|fastn::)?[A-Za-z0-9_]+)(<[^>]*>)?", name)
    base, targs = m.group(1), m.group(2) or ""
    if base in ("epl_kernel", "epl2_kernel", "track_kernel", "fft4_rows_kernel", "fft4_cols_kernel", "fft_pass_kernel", "fastn::cols_kernel",
                "fastn::rows_kernel", "fused25k::ifft_max_kernel", "fused25k::ifft_second_kernel"):   # (<1>: N = 25 000, <2>: N = 50 000)
        return base + targs.replace(", ", ",")
    return base[:48]


agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k in sorted(agg):
    for c in sorted(agg[k]):
        v = agg[k][c]
        rows.append((k, c, len(v), sum(v) / len(v), min(v), max(v)))
out_csv = os.path.join(repo, "profiles", f"{tag}_pmc_summary.csv")
with open(out_csv, "w") as f:
    f.write("kernel,counter,dispatches,mean_per_dispatch,min,max\n")
    for r in rows:
        f.write(",".join(f'"{x}"' if isinstance(x, str) and "," in x else str(x) for x in r) + "\n")


def mean(v):
    return sum(v) / len(v)


import subprocess
try:
    head = subprocess.run(["git", "-C", repo, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", repo, "status", "--porcelain", "--", "sydr_amd", "bench.py"], capture_output=True,
                                text=True).stdout.strip())
except OSError:
    head, dirty = "", False
info = {"correction": "FETCH_SIZE doubled (gfx950, 16-B/lane streaming reads)", "source": f"profiles/{tag}_pmc_summary.csv",
        "git_head": head + ("+uncommitted" if dirty else ""), "build_id": profiled_build,
        "note": "counters belong to the kernel VARIANT named here; bench.py prints them only for a run of that variant"}
# the headline launch: ci8, 3 taps; the full 32000-epoch launches are the dispatches with the most waves
head = [k for k in agg if k.startswith("epl_kernel<0,3,")]
if head:
    k = max(head, key=lambda k: len(agg[k].get("FETCH_SIZE", [])))
    epl = agg[k]

    def full(counter):  # only the full-size launches (bench steps), not the parity checks' small ones
        v = epl.get(counter, [])
        if not v:
            return None
        big = [x for x in v if x >= 0.5 * max(v)]
        return mean(big)
    if full("FETCH_SIZE") is not None and full("WRITE_SIZE") is not None:
        fetch, write = full("FETCH_SIZE") * 1024, full("WRITE_SIZE") * 1024
        info.update({"epl_kernel": k, "epl_kernel_hbm_bytes_per_launch": 2 * fetch + write,
                     "epl_fetch_size_bytes_raw": fetch, "epl_write_size_bytes": write,
                     "epl_workload": "bench.py launch of the PMC passes: 32 ch x ~25000 samples ci8 per epoch (50 KB algorithmic bytes per channel-epoch)"})
    for counter, key in (("SQ_INSTS_VALU", "epl_kernel_valu_insts_per_launch"), ("SQ_INSTS_SALU", "epl_kernel_salu_insts_per_launch"),
                         ("SQ_INSTS_LDS", "epl_kernel_lds_insts_per_launch")):
        if full(counter) is not None:
            info[key] = full(counter)
    # one single-wave workgroup per channel-epoch: SQ_WAVES of the profiled launch = its channel-epochs; bench.py scales
    # the per-epoch figures to whatever launch size it runs
    if full("SQ_WAVES"):
        epochs = full("SQ_WAVES")
        info["epl_kernel_epochs_per_profiled_launch"] = epochs
        for key in ("hbm_bytes", "valu_insts", "salu_insts", "lds_insts"):
            if f"epl_kernel_{key}_per_launch" in info:
                info[f"epl_kernel_{key}_per_epoch"] = info[f"epl_kernel_{key}_per_launch"] / epochs
# the multi-GNSS launches (configs 4-5: five taps, four epochs of a channel per workgroup)
multi = [k for k in agg if k.startswith("epl_kernel<0,5,") and agg[k].get("FETCH_SIZE") and agg[k].get("WRITE_SIZE")]
if multi:
    k = max(multi, key=lambda k: mean(agg[k]["FETCH_SIZE"]))
    m = agg[k]
    big = lambda v: mean([x for x in v if x >= 0.5 * max(v)])
    info.update({"multignss_kernel": k,
                 "multignss_hbm_bytes_per_launch": 2 * big(m["FETCH_SIZE"]) * 1024 + big(m["WRITE_SIZE"]) * 1024})
    if m.get("SQ_WAVES"):
        info["multignss_waves_per_profiled_launch"] = big(m["SQ_WAVES"])
        info["multignss_hbm_bytes_per_wave"] = info["multignss_hbm_bytes_per_launch"] / big(m["SQ_WAVES"])
        for counter, key in (("SQ_INSTS_VALU", "valu"), ("SQ_INSTS_LDS", "lds")):
            if m.get(counter):
                info[f"multignss_{key}_insts_per_wave"] = big(m[counter]) / big(m["SQ_WAVES"])
# The acquisition figure: the kernels of ONE sdr_pcps call of the headline search (N = 25 000: the calls that contain a
# fast25k:: kernel -- the bench also runs searches at other rates, which share the small kernels).  Dispatches are walked
# in order per counter pass; a call ends with its one ratio_kernel dispatch -- or, where the one-workgroup-per-transform
# kernels ran, with the second sweep, whose last workgroup per PRN divides the two peaks (round 5).
def pcps_calls(counter, marker=("fast25k", "fused25k::ifft_max_kernel<1>")):
    per_kernel, calls = collections.defaultdict(float), 0
    for path in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
        rows_ = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
        rows_.sort(key=lambda r: int(r["Dispatch_Id"]))
        cur = []
        for r in rows_:
            k = short(r["Kernel_Name"])
            if not any(k.startswith(p) for p in PCPS):
                continue
            cur.append((k, float(r["Counter_Value"])))
            if k in ("ratio_kernel", "peak_finish_kernel", "fused10k::peaks_kernel") or k.startswith("fused25k::ifft_second_kernel"):   # (the call's last kernel: map-free / with a map / 10 MHz fused / fused second sweep)
                if any(name.startswith(marker) for name, _ in cur):
                    calls += 1
                    for name, v in cur:
                        per_kernel[name] += v
                cur = []
    return per_kernel, calls


fetch_k, calls = pcps_calls("FETCH_SIZE")
write_k, calls_w = pcps_calls("WRITE_SIZE")
if calls and calls_w:
    per_kernel = {k: {"read": 2 * fetch_k.get(k, 0.0) * 1024 / calls, "write": write_k.get(k, 0.0) * 1024 / calls_w}
                  for k in sorted(set(fetch_k) | set(write_k))}
    total = sum(v["read"] + v["write"] for v in per_kernel.values())
    info.update({"pcps_hbm_bytes_per_call": total, "pcps_calls_profiled": calls, "pcps_per_kernel": per_kernel,
                 "pcps_workload": "sdr_pcps: 32 PRNs x 41 bins x 25000 samples, no map (1.05e9 algorithmic bytes)"})
# ... and of the searches at the other rates bench.py runs (their register-resident kernels carry N1 = N / 200 as a template
# argument: 50 -> the reference's shipped 10 MHz, 250 -> 50 MHz)
for marker, key, what in ((("fastn::cols_kernel<50,", "fused10k::search_kernel"), "pcps_10mhz", "ref_config leg: 32 PRNs x 34 bins x 10 blocks x 10000 samples, indices + ratio (no map)"),
                          (("fastn::cols_kernel<250,", "fused25k::ifft_max_kernel<2>"), "pcps_50mhz", "multignss leg: 32 PRNs x 41 bins x 50000 samples, no map")):
    f_k, c_f = pcps_calls("FETCH_SIZE", marker)
    w_k, c_w = pcps_calls("WRITE_SIZE", marker)
    if c_f and c_w:
        info[f"{key}_hbm_bytes_per_call"] = sum(2 * v * 1024 / c_f for v in f_k.values()) + sum(v * 1024 / c_w for v in w_k.values())
        info[f"{key}_calls_profiled"] = c_f
        info[f"{key}_workload"] = what
# the two-chips-per-lane kernel of the ref_config tracking leg (one single-wave workgroup per channel-epoch)
two = [k for k in agg if k.startswith("epl2_kernel<") and agg[k].get("FETCH_SIZE") and agg[k].get("WRITE_SIZE") and agg[k].get("SQ_WAVES")]
if two:
    k = max(two, key=lambda k: mean(agg[k]["SQ_WAVES"]))
    m = agg[k]
    big = lambda v: mean([x for x in v if x >= 0.5 * max(v)])
    info.update({"epl2_kernel": k, "epl2_kernel_hbm_bytes_per_epoch": (2 * big(m["FETCH_SIZE"]) + big(m["WRITE_SIZE"])) * 1024 / big(m["SQ_WAVES"])})
    if m.get("SQ_INSTS_VALU"):
        info["epl2_kernel_valu_insts_per_epoch"] = big(m["SQ_INSTS_VALU"]) / big(m["SQ_WAVES"])
json.dump(info, open(os.path.join(repo, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(info, indent=1))

# ---- the round's other artefacts: kernel statistics, bench lines, one-stream acquisition profile
import shutil
prof = os.path.join(repo, "profiles")
for pattern, name in (("stats/*/*kernel_stats.csv", f"{tag}_bench_kernel_stats.csv"),
                      ("pcps_one_stream/*/*kernel_stats.csv", f"{tag}_pcps_one_stream_kernel_stats.csv"),
                      ("pcps_one_stream_50/*/*kernel_stats.csv", f"{tag}_pcps50_one_stream_kernel_stats.csv")):
    # (a traced child process writes a stats file of its own into the same directory: the bench's is the largest)
    hits = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getsize, reverse=True)
    if hits:
        shutil.copyfile(hits[0], os.path.join(prof, name))
for fname, name in (("bench.json", f"{tag}_bench.json"), ("bench_plain_run.json", f"{tag}_bench_plain_run.json")):
    path = os.path.join(src, fname)
    if os.path.exists(path) and os.path.getsize(path):
        shutil.copyfile(path, os.path.join(prof, name))
for log_name, stats_name, out_name, bench_key in (
        ("pcps_one_stream.log", f"{tag}_pcps_one_stream_kernel_stats.csv", f"{tag}_pcps_one_stream.json", ("acquisition",)),
        ("pcps_one_stream_50.log", f"{tag}_pcps50_one_stream_kernel_stats.csv", f"{tag}_pcps50_one_stream.json", ("multignss", "acquisition"))):
    log = os.path.join(src, log_name)
    stats = os.path.join(prof, stats_name)
    if not (os.path.exists(log) and os.path.exists(stats)):
        continue
    line = [l for l in open(log) if l.startswith("{")][-1]
    rec = json.loads(line)
    total_ns = sum(float(r["TotalDurationNs"]) for r in csv.DictReader(open(stats))
                   if any(p in r["Name"] for p in PCPS))
    rec["rocprof_kernel_ms_per_call"] = total_ns / 1e6 / rec["calls"]
    rec["agreement"] = rec["rocprof_kernel_ms_per_call"] / rec["hip_event_kernel_ms_per_call"]
    # (the event pair of the traced process also spans what the tracer adds between a call's kernels; the figure bench.py
    # reports comes from an untraced process: held against that one too)
    plain = os.path.join(src, "bench_plain_run.json")
    try:
        leg = json.load(open(plain))
        for k in bench_key:
            leg = leg[k]
        rec["bench_kernel_ms_32_prn_untraced"] = leg["kernel_ms_32_prn"]
        rec["agreement_with_untraced_bench"] = rec["rocprof_kernel_ms_per_call"] / rec["bench_kernel_ms_32_prn_untraced"]
    except Exception:
        pass
    json.dump(rec, open(os.path.join(prof, out_name), "w"), indent=1)
    print(json.dumps(rec))
