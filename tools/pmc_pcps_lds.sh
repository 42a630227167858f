#!/bin/bash
# LDS-side counters of the fused PCPS search (fused25k::ifft_max_kernel): is the LDS pipe, and are bank conflicts, what the
# barrier-separated phases wait for?  Separate --pmc passes, the program directly behind `--`.
#   tools/pmc_pcps_lds.sh <tag> [fs_mhz]   -> gpurun_out/pmc_pcps_lds_<tag>/summary.txt
set -u
TAG=${1:-x}
FS=${2:-25}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_pcps_lds_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
PROG="$ROOT/tools/pcps_one_stream.py"
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d "$OUT/lds" -- python3 "$PROG" $FS > "$OUT/lds.log" 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d "$OUT/sq" -- python3 "$PROG" $FS > "$OUT/sq.log" 2>&1
python3 - "$OUT" <<'PY' > "$OUT/summary.txt"
import collections, csv, glob, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = n.split("(")[0][:60]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if "fused" not in k:
        continue
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:24s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
cat "$OUT/summary.txt"
