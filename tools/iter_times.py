#!/usr/bin/env python3
"""Wall time of every single EM iteration of the default bench workload (cfg4), with what the iteration changed: shows where the
first iterations of a run differ from the converged ones.  usage: python3 tools/iter_times.py [n_iter] [--opt k=v ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import WORKLOADS
from cellector_amd import Cellector

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 30
N, L, d, pct = WORKLOADS[os.environ.get("CELLECTOR_BENCH_WORKLOAD", "cfg4")]
g = Cellector(0, stream=torch.cuda.current_stream().cuda_stream)
g.set_option("keep_coo", 0)
for kv in [a for a in sys.argv[1:] if "=" in a]:
    k, v = kv.split("=")
    g.set_option(k, int(v))
g.set_option("synth_continue_pct", pct)
g.ingest_synthetic(L, N, d, seed=4, minority_fraction=0.05, doublet_fraction=0.0)
g.ingest_finish(4, 4)
torch.cuda.synchronize()
for i in range(n_iter):
    t0 = time.perf_counter()
    s = g.em_iteration(5.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    print(f"iter {i:2d}: {ms:7.3f} ms  new {s.n_new_excluded:6d} rescued {s.n_rescued:6d} excluded {s.n_excluded:6d} filtered {s.n_loci_filtered:4d} change {s.any_change}")
g.close()
