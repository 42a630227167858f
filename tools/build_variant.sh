#!/bin/bash
# A/B builds of one translation unit:  tools/build_variant.sh <tag> <unit> [-DFLAG ...]  ->  tools/scratch/var/lib_<tag>.so
# (the other objects come from the regular build in sydr_amd/csrc; select a variant with SYDR_AMD_LIB=<path>)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; UNIT=$2; shift 2
mkdir -p "$ROOT/tools/scratch/var"
cd "$ROOT/sydr_amd/csrc"
EXTRA=""; { [ "$UNIT" = track_dense ] || [ "$UNIT" = pcps_fused ]; } && [ -z "$KEEP_LICM" ] && EXTRA="-mllvm -disable-machine-licm"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-result $EXTRA "$@" -c $UNIT.hip -o /tmp/var_${TAG}_$UNIT.o
OBJS=""
for u in engine codes epl epl_straight pcps pcps_fused track track_dense schedule; do
  if [ $u = $UNIT ]; then OBJS="$OBJS /tmp/var_${TAG}_$UNIT.o"; else OBJS="$OBJS $u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o "$ROOT/tools/scratch/var/lib_$TAG.so" $OBJS
echo "built tools/scratch/var/lib_$TAG.so"
