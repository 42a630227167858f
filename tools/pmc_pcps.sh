#!/bin/bash
# SQ / LDS / memory counters of one acquisition (separate --pmc passes, no tracing) -> gpurun_out/pmc_pcps_<tag>/summary.txt
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_pcps_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq" -- python3 "$ROOT/tools/pcps_breakdown.py" > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d "$OUT/sq2" -- python3 "$ROOT/tools/pcps_breakdown.py" > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/tools/pcps_breakdown.py" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/tools/pcps_breakdown.py" > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY' > "$OUT/summary.txt"
import collections, csv, glob, os, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(os.path.join(sys.argv[1], "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        n = n.split("(")[0][:44]
        agg[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    if not any(x in k for x in ("fft4", "argmax", "peak_finish")):
        continue
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:24s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
PY
cat "$OUT/summary.txt"
