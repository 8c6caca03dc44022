"""Round 5: times gemm_nt_p8 (forced through hgr_gemm_set_p8) of the loaded library (HGR_LIB = an ablation build, -DP8_ABL=n: profiles/NOTES.md,
round 5; not in the tree) on c_fc and a K = 3072 shape; run through tools/lib_ab.sh."""
import sys, json, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from hgr_net_amd import _lib
from tools.p8_bench import Case
from tools.ws_bench import timeit
_lib.load().hgr_gemm_set_p8(1)
cases = {"fc": Case(25600, 3072, 768, True), "big": Case(25600, 3072, 3072, False)}
for _ in range(3):
    for c in cases.values(): c.run(1)
torch.cuda.synchronize()
out = {n: round(min(timeit(lambda: c.run(1), 20) for _ in range(6)), 1) for n, c in cases.items()}
print(os.environ.get("HGR_LIB", "tree").split("/")[-1], json.dumps(out))
