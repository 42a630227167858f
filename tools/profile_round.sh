#!/bin/bash
# Everything a round's profiles/ entry needs, on the GPU box:  tools/profile_round.sh <tag>
#   1. tools/profile_bench.sh <tag>      rocprofv3 kernel statistics of the default bench command + separate PMC passes
#   2. the same bench command WITHOUT the profiler (what the driver's own run measures)
#   3. one-stream acquisitions under --kernel-trace --stats (tools/pcps_one_stream.py), at 25 and at 50 MHz
# then, back in the build container:  python tools/summarize_pmc.py gpurun_out/prof_<tag> <tag>
set -u
TAG=${1:-r00}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
bash "$ROOT/tools/profile_bench.sh" "$TAG"
cd "$ROOT" && python3 bench.py --steps 20 --warmup 5 > "$OUT/bench_plain_run.json" 2> "$OUT/bench_plain_run.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pcps_one_stream" -- python3 "$ROOT/tools/pcps_one_stream.py" > "$OUT/pcps_one_stream.log" 2>&1
find "$OUT/pcps_one_stream" -name "*kernel_trace.csv" -size +8M -delete
tail -1 "$OUT/pcps_one_stream.log"
# ... and the same at N = 50 000 (configs 4-5's rate: the multignss leg's acquisition)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/pcps_one_stream_50" -- python3 "$ROOT/tools/pcps_one_stream.py" 50 > "$OUT/pcps_one_stream_50.log" 2>&1
find "$OUT/pcps_one_stream_50" -name "*kernel_trace.csv" -size +8M -delete
tail -1 "$OUT/pcps_one_stream_50.log"
tail -c 400 "$OUT/bench_plain_run.json"
# ... and the closed-loop kernels' counters (tools/pmc_track.sh: track_kernel dense 768 / 256 channels, cluster 32 x 8)
bash "$ROOT/tools/pmc_track.sh" "$TAG" > "$OUT/pmc_track.log" 2>&1
# ... and the fused PCPS kernel's LDS side (bank conflicts / LDS busy: tools/lds_conflicts_fused.py is the model they pin)
bash "$ROOT/tools/pmc_pcps_lds.sh" "$TAG" 25 > "$OUT/pmc_pcps_lds.log" 2>&1
cp -r "$ROOT/gpurun_out/pmc_pcps_lds_$TAG" "$OUT/pmc_pcps_lds" 2>/dev/null
