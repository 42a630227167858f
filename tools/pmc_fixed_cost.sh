#!/bin/bash
# SQ instruction counters of epl_kernel per wave (= per epoch) at two epoch lengths of one code step: the difference is
# the per-sample part, the rest the per-epoch part.   tools/pmc_fixed_cost.sh [step] [n_full]
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
STEP=${1:-0.04092}
NFULL=${2:-24987}
OUT=$ROOT/gpurun_out/pmc_fixed
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in $NFULL $((NFULL / 2)); do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD \
     --output-format csv -d "$OUT/n$n" -- python3 "$ROOT/tools/epl_fixed_cost.py" --step $STEP --only-n $n > "$OUT/n$n.log" 2>&1
  python3 - "$OUT/n$n" "$n" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "epl_kernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, a in acc.items():
    w = a.get("SQ_WAVES", 1.0)
    print("n =", sys.argv[2], k, {c: round(v / w, 1) for c, v in sorted(a.items())})
PY
done
