#!/bin/bash
# Regenerate the judged profile set of one build on the GPU box (run through gpurun from the repo root):
#
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02_a'
#
# 1. plain `bench.py` (the bench line),
# 2. `rocprofv3 --kernel-trace --stats -- python3 bench.py` (per-kernel time; its own bench line is kept beside it),
# 3. two PMC passes of the same command, kernel-trace only (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc is never
#    combined with --stats / sys traces on this pool),
# 4. tools/pmc_summary.py -> profiles/<tag>_pmc_summary.json, stamped with the kernel-source hash bench.py checks.
# Everything is written under gpurun_out/<tag>/ and the summaries are copied into profiles/ (commit those).
set -eo pipefail
TAG=${1:?tag, e.g. r02_a}
ARGS=${2:-"--steps 30 --warmup 5"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT" "$ROOT/profiles"
export TMPDIR=/tmp
cd /tmp

python3 "$ROOT/bench.py" $ARGS > "$OUT/bench.json" 2> "$OUT/bench.err"
echo "[profile] bench done: $(cut -c1-200 "$OUT/bench.json")"

rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o "$TAG" -- python3 "$ROOT/bench.py" $ARGS --no-cpu-baseline --no-pcie --no-secondary --no-host-probe --no-power-probe \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
echo "[profile] kernel-trace stats done"

rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-secondary --no-host-probe --no-power-probe \
    > /dev/null 2> "$OUT/rocprof_fetch.err"
echo "[profile] FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-pcie --no-secondary --no-host-probe --no-power-probe \
    > /dev/null 2> "$OUT/rocprof_write.err"
echo "[profile] WRITE_SIZE pass done"

STATS=$(find "$OUT/stats" -name "*kernel_stats.csv" | head -1)
FETCH=$(find "$OUT/pmc_fetch" -name "*counter_collection.csv" | head -1)
WRITE=$(find "$OUT/pmc_write" -name "*counter_collection.csv" | head -1)
python3 "$ROOT/tools/pmc_summary.py" "$TAG" "$STATS" "$FETCH" "$WRITE"
cp "$STATS" "$ROOT/profiles/${TAG}_vitb32_kernel_stats.csv"
cp "$OUT/bench.json" "$ROOT/profiles/${TAG}_vitb32_bench.json"
cp "$OUT/bench_under_rocprof.json" "$ROOT/profiles/${TAG}_vitb32_bench_under_rocprof.json"
# the other two BASELINE configs the default bench line carries as `secondary` (configs[2] RN50 + hierarchy, configs[4] ViT-L/14 + CoOp
# OM training step): kernel-trace statistics and the two PMC passes of each (PROFILE_SECONDARY=0 skips them)
profile_config() {   # <name> <bench args...>
    local NAME=$1; shift
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${NAME}_stats" -o "$TAG" -- python3 "$ROOT/bench.py" "$@" \
        > "$OUT/${NAME}_bench_under_rocprof.json" 2> "$OUT/${NAME}_rocprof_stats.err"
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/${NAME}_pmc_fetch" -o fetch -- python3 "$ROOT/bench.py" "$@" > /dev/null 2> "$OUT/${NAME}_rocprof_fetch.err"
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/${NAME}_pmc_write" -o write -- python3 "$ROOT/bench.py" "$@" > /dev/null 2> "$OUT/${NAME}_rocprof_write.err"
    local S=$(find "$OUT/${NAME}_stats" -name "*kernel_stats.csv" | head -1)
    local F=$(find "$OUT/${NAME}_pmc_fetch" -name "*counter_collection.csv" | head -1)
    local W=$(find "$OUT/${NAME}_pmc_write" -name "*counter_collection.csv" | head -1)
    python3 "$ROOT/tools/pmc_summary.py" "$TAG" "$S" "$F" "$W" "$NAME"
    cp "$S" "$ROOT/profiles/${TAG}_${NAME}_kernel_stats.csv"
    cp "$OUT/${NAME}_bench_under_rocprof.json" "$ROOT/profiles/${TAG}_${NAME}_bench_under_rocprof.json"
    echo "[profile] $NAME done"
}
if [ "${PROFILE_SECONDARY:-1}" != "0" ]; then
    profile_config rn50 --arch RN50 --nodes 20842 --steps 8 --warmup 2 --no-pcie --no-cpu-baseline --no-power-probe --secondary
    profile_config train_l14 --mode train --arch ViT-L/14 --n-ctx 16 --batch 256 --steps 2 --warmup 1 --secondary
fi
# profiles/ on the box is outside gpurun_out/: ship copies back through it
mkdir -p "$OUT/profiles" && cp "$ROOT/profiles/${TAG}"_* "$OUT/profiles/"
echo "[profile] summaries in gpurun_out/$TAG/profiles/ (copy into profiles/ and commit)"
