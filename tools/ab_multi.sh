#!/bin/bash
# same-box comparison of several library builds: tools/ab_multi.sh <fs> <rounds> lib1 lib2 ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
FS=$1; R=$2; shift 2
for i in $(seq 1 $R); do
  for L in "$@"; do
    echo -n "$(basename $L) "
    SYDR_AMD_LIB=$L python3 $ROOT/tools/pcps_one_stream.py $FS | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms %.4f wall_ms %.4f' % (d['hip_event_kernel_ms_per_call'], d['wall_ms_per_call_one_stream']))"
  done
done
