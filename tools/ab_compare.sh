#!/bin/bash
# Same-box A/B of two builds of the library:  tools/ab_compare.sh <libA.so> <libB.so> <rounds> -- <command...>
# Runs the command alternately with each library installed (box-to-box variance on the pool is ~5 %).
A=$1; B=$2; R=$3; shift 4
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cp "$ROOT/sydr_amd/libsydr_amd.so" /tmp/ab_orig.so
for i in $(seq 1 "$R"); do
  cp "$A" "$ROOT/sydr_amd/libsydr_amd.so"; echo "[A] $("$@" 2>&1 | tail -1)"
  cp "$B" "$ROOT/sydr_amd/libsydr_amd.so"; echo "[B] $("$@" 2>&1 | tail -1)"
done
cp /tmp/ab_orig.so "$ROOT/sydr_amd/libsydr_amd.so"
