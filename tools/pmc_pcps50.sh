#!/bin/bash
# L2 / fabric counters of the fused PCPS search at 50 MHz (tools/pcps_one_stream.py 50), one counter group per pass:
#   tools/pmc_pcps50.sh <tag>   -> gpurun_out/pmc_pcps50_<tag>/{fetch,write,tcc}/...counter_collection.csv
set -u
TAG=${1:-a}
FS=${2:-50}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_pcps${FS}_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/tools/pcps_one_stream.py" $FS > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/tools/pcps_one_stream.py" $FS > "$OUT/write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/tcc" -- python3 "$ROOT/tools/pcps_one_stream.py" $FS > "$OUT/tcc.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k, {c: round(sum(v) / len(v), 1) for c, v in agg[k].items()}, "launches", max(len(v) for v in agg[k].values()))
PY
