"""GPU: the product shard engine (HipShardEngine) under ShardedCellector, and bench.py's contract on a tiny case."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_hip_shard_engine_single_rank_equals_plain_context(hip_lib_path):
    import torch
    from cellector_amd import Cellector, synth
    from cellector_amd.sharded import HipShardEngine, ShardedCellector
    L, N, d = 1200, 900, 0.1
    coo = synth.generate_coo(L, N, d, seed=3, minority_fraction=0.07)
    sc = ShardedCellector(HipShardEngine(0), None)
    sc.load(lambda e: e.ingest_coo(L, N, *coo), N)
    s1 = sc.run(5.0, 30)
    with Cellector(0) as g:
        g.load_coo(L, N, *coo)
        s2 = g.run(5.0, 30)
        assert [(a.threshold, a.n_new_excluded, a.n_excluded) for a in s1] == [(b.threshold, b.n_new_excluded, b.n_excluded) for b in s2]
        assert np.array_equal(sc.e.excluded(), g.excluded())
        p1, p2 = sc.e.posteriors(), g.posteriors()
        assert np.array_equal(p1["posterior"], p2["posterior"])
    # the exchange buffers are torch tensors on the device: NORM holds every cell's normalised LL
    norm = sc.e.buf[1].cpu().numpy()
    assert np.array_equal(norm, sc.e.cell_outputs()["normalized"])
    torch.cuda.synchronize()


def test_bench_contract_on_cfg1():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "hbm"
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
