"""Round 5: times the LayerNorm-consumer GEMM of the loaded library (HGR_LIB = an ablation build of hgr_gemm_duo.hip, -DHGR_DUO_ABL=n: the
experiment switch is described in profiles/NOTES.md, round 5, and is not in the tree) on three shapes; run through tools/lib_ab.sh."""
import sys, json, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from tools.ws_bench import LnC, Plain, LnP, timeit, set_ws
set_ws(0)
M = 25600
cases = {"fc": LnC(M, 3072, 768, True), "qkv": LnC(M, 2304, 768, False), "big": LnC(M, 3072, 3072, False)}
for _ in range(3):
    for c in cases.values(): c.run(0)
torch.cuda.synchronize()
out = {}
for n, c in cases.items():
    t = [timeit(lambda: c.run(0), 20) for _ in range(6)]
    out[n] = round(min(t), 1)
print(os.environ.get("HGR_LIB", "tree").split("/")[-1], json.dumps(out))
