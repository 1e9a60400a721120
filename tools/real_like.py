#!/usr/bin/env python3
"""EM iteration time on shapes like real 10x runs (few thousand cells, 10^5..10^6 loci): python3 tools/real_like.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from cellector_amd import Cellector

for N, L, d in ((3000, 100000, 0.03), (8000, 300000, 0.02), (20000, 500000, 0.015), (60000, 1000000, 0.01)):
    g = Cellector(0, stream=torch.cuda.current_stream().cuda_stream)
    g.set_option("keep_coo", 0)
    t0 = time.perf_counter()
    g.load_synthetic(L, N, d, seed=4, minority_fraction=0.05, min_alt=4, min_ref=4)
    torch.cuda.synchronize()
    t_load = time.perf_counter() - t0
    dm = g.dims()
    for _ in range(3):
        g.em_iteration(5.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        s = g.em_iteration(5.0)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    g.posteriors()
    torch.cuda.synchronize()
    t_post = (time.perf_counter() - t0) * 1e3
    e = g.engine_info()
    print(f"N={N} L={L} d={d}: loci_used={dm.loci_used} nnz={dm.nnz_used} ingest {t_load*1e3:.0f} ms, iteration {ms:.3f} ms ({dm.nnz_used/ms/1e6:.0f} M entries/ms), posteriors {t_post:.2f} ms, excluded {s.n_excluded}")
    g.close()
