#!/bin/bash
# Where an epoch's instructions go at the low front-end rates (tools/rate_by_fs.py: 4 MHz per-sample core, 10 MHz two chips per
# lane): SQ counters of the E/P/L launches, one counter group per pass (rocprofv3 --pmc alone), per wave = per channel-epoch.
#   tools/pmc_rate.sh <tag> [MHz ...]      -> gpurun_out/pmc_rate_<tag>/summary.txt
set -u
TAG=${1:-a}; shift
RATES=${@:-4 10 25}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_rate_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SMEM"
G2="SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64"
for r in $RATES; do
  python3 "$ROOT/tools/rate_by_fs.py" $r > "$OUT/plain_$r.txt" 2>&1
  n=0
  for G in "$G1" "$G2"; do
    n=$((n+1))
    echo "rate $r group $n: $G" >> "$OUT/progress.txt"
    timeout -k 10 150 rocprofv3 --pmc $G --output-format csv -d "$OUT/r${r}_g$n" -- python3 "$ROOT/tools/rate_by_fs.py" $r > "$OUT/r${r}_g$n.log" 2>&1 || echo "  (pass failed or timed out)" >> "$OUT/progress.txt"
  done
done
python3 - "$OUT" $RATES <<'PY' > "$OUT/summary.txt"
import csv, glob, collections, sys
out, rates = sys.argv[1], sys.argv[2:]
table = collections.defaultdict(dict)
for r in rates:
    print(open(f"{out}/plain_{r}.txt").read().strip().splitlines()[-1])
    for path in glob.glob(f"{out}/r{r}_g*/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for row in csv.DictReader(open(path)):
            if "epl" in row["Kernel_Name"] and "setup" not in row["Kernel_Name"]:
                agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for c, v in agg.items():
            big = [x for x in v if x >= 0.5 * max(v)] or v      # (the full-size launches)
            table[c][r] = sum(big) / len(big)
print("counter (per launch)".ljust(44), *[f"{r} MHz".rjust(16) for r in rates])
for c in sorted(table):
    print(c.ljust(44), *[f"{table[c].get(r, float('nan')):16.4g}" for r in rates])
print("per wave (= per channel-epoch)".ljust(44))
for c in sorted(table):
    if c.startswith("SQ_INSTS") or c in ("SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY"):
        print(("  " + c).ljust(44), *[f"{table[c].get(r, float('nan')) / table['SQ_WAVES'].get(r, float('nan')):16.1f}" for r in rates])
PY
cat "$OUT/summary.txt"
