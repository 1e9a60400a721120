#!/usr/bin/env python3
"""Wall time of the posterior phase (three LL passes + the per-cell formulae + 32 MB of results to the host) at a bench workload."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import WORKLOADS
from cellector_amd import Cellector
w = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
N, L, d = WORKLOADS[w]
g = Cellector(0)
g.set_option("keep_coo", 0)
g.load_synthetic(L, N, d, seed=4, min_alt=4, min_ref=4)
for _ in range(4):
    s = g.em_iteration(5.0)
torch.cuda.synchronize()
for k in range(3):
    t0 = time.perf_counter()
    p = g.posteriors()
    print(w, "posteriors: %.2f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
g.close()
