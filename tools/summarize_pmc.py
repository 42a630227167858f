#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection CSVs (one pass per counter group, as the gfx950 guide
prescribes) into a per-kernel summary and profiles/pmc_traffic.json (read by bench.py).

    python tools/summarize_pmc.py gpurun_out/prof_r01b r01
HBM bytes per launch = 2 * FETCH_SIZE*1024 (gfx950: FETCH_SIZE reads exactly half of a 16-B/lane
streaming read, MI355X_MICROARCH.md section HBM) + WRITE_SIZE*1024."""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    for key in ("epl_kernel", "track_kernel", "fft4_rows_kernel", "fft4_cols_kernel", "fft_pass_kernel", "argmax_part", "peak_finish", "synth_kernel"):
        if key in name:
            return key + ("<inv>" if key == "fft_pass_kernel" and ", true," in name else "")
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:48]


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
        f.write(",".join(str(x) for x in r) + "\n")
epl = agg.get("epl_kernel", {})
if "FETCH_SIZE" in epl and "WRITE_SIZE" in epl:
    fetch = sum(epl["FETCH_SIZE"]) / len(epl["FETCH_SIZE"]) * 1024
    write = sum(epl["WRITE_SIZE"]) / len(epl["WRITE_SIZE"]) * 1024
    info = {"epl_kernel_hbm_bytes_per_launch": 2 * fetch + write, "fetch_size_bytes_raw": fetch,
            "write_size_bytes": write, "correction": "FETCH_SIZE doubled (gfx950, 16-B/lane streaming reads)",
            "workload": "bench.py step: 32 ch x 1000 epochs x ~25000 samples ci8 (1.6e9 algorithmic bytes)",
            "source": f"profiles/{tag}_pmc_summary.csv"}
    if "SQ_INSTS_VALU" in epl:
        info["valu_insts_per_launch"] = sum(epl["SQ_INSTS_VALU"]) / len(epl["SQ_INSTS_VALU"])
    json.dump(info, open(os.path.join(repo, "profiles", "pmc_traffic.json"), "w"), indent=1)
    print(json.dumps(info, indent=1))
print(open(out_csv).read())
