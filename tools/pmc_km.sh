#!/bin/bash
# What differs between the straight-line E/P/L kernels of neighbouring block lengths (tools/rate_by_km.py: KM.5 samples per
# chip): SQ / TA / TCP / TD counters of the epl_kernel launches, one counter group per pass (rocprofv3 --pmc alone).
#   tools/pmc_km.sh <tag> [KM ...]      -> gpurun_out/pmc_km_<tag>/summary.txt
set -u
TAG=${1:-a}; shift
KMS=${@:-19 21 24}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_km_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"
G2="TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"
# (a TA_* group -- TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum -- never came
# back on this pool: the pass sat for seven minutes until the runner killed it.  Left out; every pass under its own timeout.)
G4="TCP_TAGRAM0_REQ_sum TCP_TAGRAM1_REQ_sum TCP_TAGRAM2_REQ_sum TCP_TAGRAM3_REQ_sum"
G5="TD_TC_STALL_sum TD_TD_BUSY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
G6="SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"
for km in $KMS; do
  python3 "$ROOT/tools/rate_by_km.py" $km > "$OUT/plain_$km.txt" 2>&1
  n=0
  for G in "$G1" "$G2" "$G4" "$G5" "$G6"; do
    n=$((n+1))
    echo "KM $km group $n: $G" >> "$OUT/progress.txt"
    timeout -k 10 120 rocprofv3 --pmc $G --output-format csv -d "$OUT/km${km}_g$n" -- python3 "$ROOT/tools/rate_by_km.py" $km > "$OUT/km${km}_g$n.log" 2>&1 || echo "  (pass failed or timed out)" >> "$OUT/progress.txt"
  done
done
python3 - "$OUT" $KMS <<'PY' > "$OUT/summary.txt"
import csv, glob, collections, sys
out, kms = sys.argv[1], sys.argv[2:]
table = collections.defaultdict(dict)
for km in kms:
    print(open(f"{out}/plain_{km}.txt").read().strip().splitlines()[-1])
    for path in glob.glob(f"{out}/km{km}_g*/*/*counter_collection.csv"):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if "epl_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in agg.items():
            big = [x for x in v if x >= 0.5 * max(v)] or v      # (the full-size launches)
            table[c][km] = sum(big) / len(big)
print("counter".ljust(44), *[f"KM={k}".rjust(16) for k in kms])
for c in sorted(table):
    print(c.ljust(44), *[f"{table[c].get(k, float('nan')):16.4g}" for k in kms])
PY
cat "$OUT/summary.txt"
