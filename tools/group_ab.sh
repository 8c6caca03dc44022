#!/bin/bash
# Dev tool (round 5): raster group by shape (HGR_GEMM_GROUP=0) against the fixed 4, all three BASELINE configurations, interleaved.
set -eo pipefail
OUT=$PWD/$1; mkdir -p "$OUT"
line() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], d["ms_per_step"], d["step_ms"]["p50"], d["roofline"].get("frac"), flush=True)
PY
}
for r in 1 2; do
  for g in 4 0; do
    HGR_GEMM_GROUP=$g python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pcie --no-secondary --no-host-probe --no-power-probe > "$OUT/vit_g${g}_$r.json" 2> "$OUT/vit_g${g}_$r.err"; line "$OUT/vit_g${g}_$r.json" "vitb32 group=$g"
    HGR_GEMM_GROUP=$g python3 bench.py --arch RN50 --nodes 20842 --steps 15 --warmup 3 --no-pcie --no-cpu-baseline --no-power-probe --secondary > "$OUT/rn_g${g}_$r.json" 2> "$OUT/rn_g${g}_$r.err"; line "$OUT/rn_g${g}_$r.json" "rn50 group=$g"
    HGR_GEMM_GROUP=$g python3 bench.py --mode train --arch ViT-L/14 --n-ctx 16 --batch 256 --steps 3 --warmup 1 --secondary > "$OUT/tr_g${g}_$r.json" 2> "$OUT/tr_g${g}_$r.err"; line "$OUT/tr_g${g}_$r.json" "train group=$g"
  done
done
