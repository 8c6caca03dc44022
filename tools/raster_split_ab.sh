#!/bin/bash
# Dev tool (round 5): raster / cache-policy arms of the tower kernels - step time (bench runs, two rounds interleaved) and FETCH_SIZE per
# launch of the tower kernels per arm.  An arm = "tag ENV=.. ENV=..".
#   gpurun -- 'bash tools/raster_split_ab.sh gpurun_out/raster "base HGR_GEMM_NSPLIT=0" "split HGR_GEMM_NSPLIT=2" ...'
set -eo pipefail
ROOT=$PWD; OUT=$PWD/$1; shift
ARMS=("$@")
mkdir -p "$OUT"; export TMPDIR=/tmp
B="--no-cpu-baseline --no-pcie --no-secondary --no-host-probe --no-power-probe"
for r in 1 2; do
  for arm in "${ARMS[@]}"; do
    read -r -a w <<< "$arm"; tag=${w[0]}
    env "${w[@]:1}" python3 bench.py --steps 30 --warmup 5 $B > "$OUT/${tag}_$r.json" 2> "$OUT/${tag}_$r.err"
    python3 - "$OUT/${tag}_$r.json" "$tag" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["ms_per_step"], d["step_ms"]["p50"], {k: v["avg_us"] for k, v in d["roofline"].get("by_shape", {}).items()}, flush=True)
PY
  done
done
cd /tmp
for arm in "${ARMS[@]}"; do
  read -r -a w <<< "$arm"; tag=${w[0]}
  # (env before rocprofv3, not after "--": the profiled program itself must follow the "--")
  env "${w[@]:1}" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_$tag" -o fetch -- python3 "$ROOT/bench.py" --steps 10 --warmup 3 $B > /dev/null 2> "$OUT/pmc_$tag.err"
  python3 - "$(find "$OUT/pmc_$tag" -name '*counter_collection.csv' | head -1)" "$tag" <<'PY'
import csv,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=r["Kernel_Name"].replace("void ","").replace("(anonymous namespace)::","").replace("hgr_gemm::","").split("(")[0]
    d[n].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(2*1024*sum(v)/len(v)/1e6,1) for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:4]}, "read MB/launch", flush=True)
PY
  rm -rf "$OUT/pmc_$tag"
done
