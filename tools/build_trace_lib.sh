#!/bin/bash
# Builds the debug variant of the library (per-workgroup / per-phase clocks) into tools/libsydr_trace.so
# without touching the product build.  On the GPU box:  cp tools/libsydr_trace.so sydr_amd/libsydr_amd.so
set -e
cd "$(dirname "$0")/../sydr_amd/csrc"
mkdir -p /tmp/sdr_trace_build
make build_id.h          # (engine.hip includes it; generated, not tracked)
for f in engine codes epl epl_straight pcps pcps_fused track track_dense schedule; do
  EXTRA=""
  # as the Makefile builds them: 256 / 168 registers per lane and nothing may spill -- a trace of a spilling kernel is not a
  # trace of the shipped one
  { [ $f = track_dense ] || [ $f = pcps_fused ]; } && EXTRA="-mllvm -disable-machine-licm"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fvisibility=hidden $EXTRA -DSDR_TRACE_WG -DSDR_TRACE_TRACK -I../../include -c $f.hip -o /tmp/sdr_trace_build/$f.o &
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,--version-script=exports.map /tmp/sdr_trace_build/*.o -o ../../tools/libsydr_trace.so
echo built tools/libsydr_trace.so
