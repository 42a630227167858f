#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + separate PMC passes of bench.py.
# Usage: tools/profile_bench.sh <tag>      -> gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
# the build the counters belong to (tools/summarize_pmc.py refuses them when the tree has moved on)
(cd "$ROOT" && python3 -c "import sydr_amd; print(sydr_amd.load().sdr_build_id().decode())") > "$OUT/build_id.txt"
cd /tmp && export TMPDIR=/tmp
# the driver's command with short CPU legs and WITHOUT the process-per-channel CPU baseline: its 32 spawned workers each start
# under the profiler's preloaded library, and one run of the round hung in their shutdown (the untraced run of profile_round.sh keeps it)
# (no C per-tick leg under the profiler: gcc re-execs, and a traced child writes a second stats file; bench.py also skips the leg by itself when LD_PRELOAD / ROCP* is set)
ARGS="--steps 20 --warmup 5 --cpu-seconds 2 --cpu-mp-seconds 0 --no-per-tick-c"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 "$ROOT/bench.py" $ARGS > "$OUT/bench_stats.log" 2>&1
PARGS="--steps 2 --warmup 1 --stream-seconds 6 --cpu-seconds 0.2 --cpu-mp-seconds 0 --no-closed-loop --no-per-tick --no-rates"   # (the ref_config and multignss legs run: their kernels get counters too)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOT/bench.py" $PARGS > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOT/bench.py" $PARGS > "$OUT/bench_pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOT/bench.py" $PARGS > "$OUT/bench_pmc_sq.log" 2>&1
grep -h '^{"metric' "$OUT"/bench_stats.log | tail -1 > "$OUT/bench.json"
find "$OUT" -name "*.csv" | head -20
# keep only the small summaries (traces of 6M-block synth launches are small too, but cap anyway)
find "$OUT" -name "*kernel_trace.csv" -size +8M -delete
