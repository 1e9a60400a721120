#!/usr/bin/env python3
"""GPU box: time of the order statistics inside an EM iteration, plain ctx (select over all keys) against a ctx with a
one-rank RCCL communicator (sharded select: six digit-histogram levels, each followed by a real ncclAllReduce of 48 KB).
  python tools/select_ab.py [cells loci]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cellector_amd import Cellector, ffi  # noqa: E402

N, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1_000_000, 200_000)
out = {}
for name in ("plain", "one_rank_rccl_sharded", "one_rank_rccl_gather"):
    if name != "plain":
        os.environ["CELLECTOR_COMM_SELFTEST"] = "1"
    g = Cellector(0)
    if name != "plain":
        g.comm_init_rank(ffi.comm_unique_id(), 1, 0)
        os.environ.pop("CELLECTOR_COMM_SELFTEST", None)
        g.set_option("sharded_select", 1 if name.endswith("sharded") else 0)
    g.set_option("timing", 1)
    g.load_synthetic(L, N, 0.01, seed=4, minority_fraction=0.05)
    for _ in range(3):
        g.em_iteration(5.0)
    g.reset_timing()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        s = g.em_iteration(5.0)
    dt = (time.perf_counter() - t0) / K
    ms, n = g.kernel_time(ffi.K_SELECT) if hasattr(ffi, "K_SELECT") else g.kernel_time(2)
    out[name] = {"ms_per_iteration": dt * 1e3, "select_ms": ms / max(n, 1), "threshold": s.threshold}
    g.close()
assert len({v["threshold"] for v in out.values()}) == 1
print(json.dumps({"cells": N, "loci": L, **out}))
