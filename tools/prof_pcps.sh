#!/bin/bash
# rocprofv3 kernel statistics of one acquisition (tools/pcps_breakdown.py) -> gpurun_out/prof_pcps_<tag>/
set -u
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_pcps_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/tools/pcps_breakdown.py" > "$OUT/run.log" 2>&1
cat "$OUT/run.log" | tail -8
f=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg_us {float(r["AverageNs"])/1e3:9.2f} total_ms {float(r["TotalDurationNs"])/1e6:8.3f}')
PY
