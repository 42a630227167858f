#!/bin/bash
# PMC passes over one warm stream of map-free acquisitions at 25 MHz (the register-resident 125 x 200 kernels):
#   tools/pmc_pcps_fast.sh <tag>     ->  gpurun_out/pmc_<tag>/
set -u
TAG=${1:-pcpsfast}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$ROOT/tools/pcps_stream.py" 20 > "$OUT/$name.log" 2>&1; }
run a GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU
run b SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR
# (a pass of eight TCC_* counters did not finish in seven minutes on this pool and was killed: SQ counters only)
python3 - "$OUT" <<'PY'
import sys, glob, csv, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "fast25k" not in k: continue
        k = "cols" if "cols_kernel" in k else "rows"
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(k, {c: round(v / n[(k, c)]) for c, v in d.items()})
PY
