#!/bin/bash
# SQ instruction counters + plain timing of the E/P/L kernel for two builds on one box:
#   tools/pmc_ab.sh <tagA> <libA.so|-> <tagB> <libB.so|->       ("-" = the regular build)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_ab
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
one() {
  tag=$1; lib=$2
  if [ "$lib" != "-" ]; then export SYDR_AMD_LIB=$ROOT/$lib; else unset SYDR_AMD_LIB; fi
  python3 "$ROOT/tools/epl_scaling.py" --only ${ONLY:-25000} --reps 40 > "$OUT/$tag.time.log" 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY \
     --output-format csv -d "$OUT/$tag" -- python3 "$ROOT/tools/epl_scaling.py" --only ${ONLY:-25000} > "$OUT/$tag.pmc.log" 2>&1
  python3 - "$OUT/$tag" "$tag" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(float)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "epl_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
w = acc.get("SQ_WAVES", 1.0)
print(sys.argv[2], {k: round(v / w, 1) for k, v in sorted(acc.items())})
PY
  grep "ms/launch" "$OUT/$tag.time.log"
}
one "$1" "$2"
one "$3" "$4"
one "$1" "$2"
