#!/bin/bash
# Counters of the closed-loop kernels (track_kernel: dense form 768 ch = three 256-thread workgroups per CU, 256 ch = one per
# CU, cluster form 32 ch x 8) -- separate --pmc passes, the program directly behind `--`, no tracing in the counter passes; one
# --kernel-trace pass for durations and the kernels' register / scratch footprint.
#   tools/pmc_track.sh <tag>   -> gpurun_out/prof_<tag>/pmc_track_*/...   (tools/summarize_pmc.py folds pmc_* into the summary)
set -u
TAG=${1:-r00}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
[ -s "$OUT/build_id.txt" ] || (cd "$ROOT" && python3 -c "import sydr_amd; print(sydr_amd.load().sdr_build_id().decode())") > "$OUT/build_id.txt"
cd /tmp && export TMPDIR=/tmp
PROG="$ROOT/tools/closed_dense_only.py"
CH="768 256 32"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/track_stats" -- python3 "$PROG" $CH > "$OUT/track_stats.log" 2>&1 || echo "(stats pass ended non-zero)"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_track_sq" -- python3 "$PROG" $CH > "$OUT/track_pmc_sq.log" 2>&1 || echo "(pass ended non-zero: the profiler crashes in its own teardown after the files are written)"
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d "$OUT/pmc_track_sq2" -- python3 "$PROG" $CH > "$OUT/track_pmc_sq2.log" 2>&1 || echo "(pass ended non-zero: the profiler crashes in its own teardown after the files are written)"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_FLAT SQ_INSTS_FLAT_LDS_ONLY SQ_INSTS_GDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_FMA_F64 --output-format csv -d "$OUT/pmc_track_sq3" -- python3 "$PROG" $CH > "$OUT/track_pmc_sq3.log" 2>&1 || true
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_track_fetch" -- python3 "$PROG" $CH > "$OUT/track_pmc_fetch.log" 2>&1 || echo "(pass ended non-zero: the profiler crashes in its own teardown after the files are written)"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_track_write" -- python3 "$PROG" $CH > "$OUT/track_pmc_write.log" 2>&1 || echo "(pass ended non-zero: the profiler crashes in its own teardown after the files are written)"
find "$OUT/track_stats" -name "*kernel_trace.csv" -size +8M -delete
tail -3 "$OUT/track_stats.log"
