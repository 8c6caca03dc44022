"""Round 6: times the tower GEMM shapes of the ViT-B/32 step with their real epilogues on the loaded library (HGR_LIB = a variant of
hgr_gemm_duo.hip built by tools/build_duo_variants.sh) and prints a checksum of every output, so that variants can be compared for
bits as well as for time; run through tools/lib_ab.sh (processes interleaved on one box)."""
import sys, json, os, zlib
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
from tools.ws_bench import LnC, Plain, LnP, timeit, set_ws
from hgr_net_amd._lib import EPI_NONE
set_ws(0)
M = 25600
cases = {"fc": LnC(M, 3072, 768, True), "qkv": LnC(M, 2304, 768, False), "out": LnP(M, 768, 768), "proj": LnP(M, 768, 3072),
         "patch": Plain(25088, 768, 3072, EPI_NONE)}
def crc(t):
    return zlib.crc32(t.contiguous().view(torch.uint8).cpu().numpy().tobytes()) & 0xffffffff
sums = {}
for n, c in cases.items():
    if hasattr(c, "reset"): c.reset()
    c.run(0)
    torch.cuda.synchronize()
    if isinstance(c, LnP): sums[n] = [crc(c.xh[0]), crc(c.xl[0]), crc(c.st[0])]
    else: sums[n] = [crc(c.out[0])]
for _ in range(3):
    for c in cases.values(): c.run(0)
torch.cuda.synchronize()
out = {}
for n, c in cases.items():
    t = [timeit(lambda: c.run(0), 20) for _ in range(6)]
    out[n] = [round(min(t), 1), round(sorted(t)[3], 1)]
print(os.environ.get("HGR_LIB", "tree").split("/")[-1], json.dumps(out), json.dumps(sums))
