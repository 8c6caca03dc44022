#!/usr/bin/env python3
"""Dev tool (round 5): GPU power and clock while the tower GEMMs run - is the step power-limited?  One process launches c_fc-shaped
LayerNorm-consumer GEMMs back to back for a few seconds per arm (gemm_nt_duo, gemm_nt_p8, idle) while this process polls rocm-smi."""
import json, subprocess, sys, time, threading
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from hgr_net_amd import _lib
from tools.p8_bench import Case


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        c = d[sorted(d)[0]]
        keep = {k: v for k, v in c.items() if any(t in k.lower() for t in ("power", "sclk", "mclk", "fclk", "junction", "edge"))}
        return keep
    except Exception as e:  # noqa
        return {"error": str(e)[:200]}


def arm(name, fn, seconds=6.0):
    samples, stop = [], False

    def poll():
        while not stop:
            samples.append(smi())
            time.sleep(0.25)
    th = threading.Thread(target=poll)
    th.start()
    t0 = time.time()
    n = 0
    while time.time() - t0 < seconds:
        if fn is None:
            time.sleep(0.05)
        else:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
    dt = time.time() - t0
    stop = True
    th.join()
    print(json.dumps({"arm": name, "launches": n, "us_per_launch": round(dt / max(n, 1) * 1e6, 1) if n else None, "samples": samples[1::3][:8]}), flush=True)


if __name__ == "__main__":
    print(json.dumps({"smi_idle": smi()}), flush=True)
    c = Case(25600, 3072, 768, True)
    arm("idle", None, 3.0)
    arm("c_fc gemm_nt_duo", lambda: c.run(0))
    arm("c_fc gemm_nt_p8", lambda: c.run(1))
    big = Case(25600, 3072, 3072, False)
    arm("K=3072 gemm_nt_duo", lambda: big.run(0))
    arm("K=3072 gemm_nt_p8", lambda: big.run(1))
    _lib.load().hgr_gemm_set_p8(2)
