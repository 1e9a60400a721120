#!/usr/bin/env python3
"""One-GPU rehearsal of the per-rank work of an R-way cell sharding (R = 1, 2, 4, 8): rank 0's shard of the BASELINE
matrix is loaded alone and iterated; the NORM exchange is emulated by a device copy of the own slice into the other
ranks' slices (same bytes as the all-gather), the LOCUS all-reduce is left out.  NOT a multi-GPU result: it shows how
the per-rank compute and the fixed per-iteration costs (order statistics over all cells, per-locus tables, launches,
the one host synchronisation) scale, i.e. the ceiling of the strong scaling that bench.py --gpus R can reach."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg4")
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--opt", action="append", default=[], help="key=value context option (repeatable)")
    args = ap.parse_args()
    import torch
    from bench import WORKLOADS
    from cellector_amd import Cellector, ffi
    N, L_total, density, _pct = WORKLOADS[args.workload]
    dev = torch.device("cuda", 0)
    out = []
    for R in [int(x) for x in args.ranks.split(",")]:
        per = (N + R - 1) // R
        g = Cellector(0, stream=torch.cuda.current_stream().cuda_stream)
        g.set_option("keep_coo", 0)
        for kv in args.opt:
            k, v = kv.split("=")
            g.set_option(k, int(v))
        g.set_shard(0, per)
        x_pass1 = torch.zeros(5 * L_total, dtype=torch.float64, device=dev)
        g.bind_exchange_buffer(ffi.XCHG_PASS1, x_pass1.data_ptr(), x_pass1.numel())
        g.ingest_synthetic(L_total, N, density, seed=4, minority_fraction=0.05, doublet_fraction=0.0)
        x_pass1 *= R  # the other ranks' shards look like this one (locus filter and alpha/beta then behave like the full matrix)
        g.ingest_finish(4, 4)
        L = g.dims().loci_used
        x_norm = torch.zeros(R * per, dtype=torch.float64, device=dev)
        g.bind_exchange_buffer(ffi.XCHG_NORM, x_norm.data_ptr(), x_norm.numel())

        def step():
            g.em_begin()
            if R > 1:
                x_norm.view(R, per)[1:] = x_norm[:per]   # stands in for the all-gather of the NORM slices
            g.em_threshold(5.0)
            return g.em_finish()

        g.set_option("norm_zero", 0)  # (the other ranks' slices are overwritten by the stand-in copy below)
        for _ in range(args.warmup):
            step()
        # timed WITHOUT event pairs (each one idles the queue ~6 us: 5 pairs per iteration are 6 % of an 8-rank iteration); the
        # per-kernel breakdown comes from a second loop
        times = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(args.steps):
                s = step()
            torch.cuda.synchronize(); times.append((time.perf_counter() - t0) / args.steps * 1e3)
        ms = sorted(times)[1]
        g.set_option("timing", 1); g.reset_timing()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        k = {n: (lambda a: a[0] / max(a[1], 1))(g.kernel_time(i)) for n, i in
             (("cell_pass", ffi.K_CELL_LL), ("tile_ll", ffi.K_TILE_LL), ("locus_pass", ffi.K_LOCUS_STATS), ("select", ffi.K_SELECT))}
        out.append({"ranks": R, "cells_per_rank": per, "nnz_rank": int(g.dims().nnz_used), "loci_used": int(L),
                    "ms_per_step": ms, "ms_per_step_runs": times, "kernels_ms": k, "n_excluded": int(s.n_excluded)})
        g.close()
        del x_norm, x_pass1
        torch.cuda.empty_cache()
    base = out[0]["ms_per_step"]
    for o in out:
        o["compute_only_speedup_vs_1"] = base / o["ms_per_step"]
    print(json.dumps({"rehearsal": "one GPU, rank 0's shard only, no collectives: NOT a multi-GPU measurement",
                      "workload": args.workload, "results": out}))


if __name__ == "__main__":
    main()
