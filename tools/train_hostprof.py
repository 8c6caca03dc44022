#!/usr/bin/env python3
"""Dev tool: cProfile of the host side of one OM training step (where do the Python microseconds go)."""
import sys, cProfile, pstats, io, runpy, os
sys.argv = ["train_bench.py"] + sys.argv[1:]
import torch
pr = cProfile.Profile()
src = open(os.path.join(os.path.dirname(__file__), "train_bench.py")).read().replace("for it in range(4):", "for it in range(5):\n    if it == 3: pr.enable()\n    if it == 4: pr.disable()")
exec(compile(src, "train_bench.py", "exec"), {"__name__": "__main__", "pr": pr, "__file__": os.path.join(os.path.dirname(__file__), "train_bench.py")})
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
