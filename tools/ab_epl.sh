#!/bin/bash
# A/B of libsydr_amd builds on the headline correlator launch: tools/ab_epl.sh <suffix> [<suffix> ...]
# (sydr_amd/libsydr_amd<suffix>.so; "" = the product build).  Same box, same process order, 3 rounds each.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for round in 1 2; do
  for v in "" "$@"; do
    r=$(SYDR_AMD_LIB=$ROOT/sydr_amd/libsydr_amd$v.so python3 $ROOT/bench.py --steps 40 --no-acquisition --no-closed-loop --no-per-tick --no-multignss --cpu-mp-seconds 0 --cpu-seconds 0.5 --stream-seconds 20 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['avg_launch_ms'],4), round(d['ms_per_step'],4))")
    echo "variant '$v': kernel_ms step_ms = $r"
  done
done
