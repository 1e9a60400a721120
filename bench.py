#!/usr/bin/env python3
"""Benchmark of the hot path: full EM iterations (compute_new_excluded, main.rs:308-347) over a synthetic
variant x cell matrix resident in HBM.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

A step = one EM iteration over the whole matrix: alpha/beta update, per-cell log-likelihood pass (with the
reference's expected-log-pmf column), exact median/quartiles over all cells, exclusion flags, per-locus
statistics pass, locus filter.  Cells are sharded across ranks (strong scaling: the matrix is fixed, BASELINE
cfg4 = 1M cells x 200k loci at 1%); the two exchanges per iteration are all-reduces over RCCL.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (cells, loci, density, continue_pct)  — BASELINE.json configs; an entry's total is 1 + Geometric(1 - pct/100)
    "cfg1": (1_000, 2_000, 0.10, 30),
    "cfg2": (50_000, 50_000, 0.01, 30),
    "cfg3": (200_000, 100_000, 0.01, 30),
    "cfg4": (1_000_000, 200_000, 0.01, 30),
    # not BASELINE configs: the same shapes with DEEP coverage (totals 1 + Geometric(0.4): 13 % of the entries have
    # alt+ref > 4 instead of 0.8 %), to show what the count distribution does to the table-driven engine
    "cfg3-deep": (200_000, 100_000, 0.01, 60),
    "cfg4-deep": (1_000_000, 200_000, 0.01, 60),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (≈6.3 TB/s achievable)
HBM_ACHIEVABLE_GBS = 6300.0  # ... the rate a plain streaming read reaches on this part


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=os.environ.get("CELLECTOR_BENCH_WORKLOAD", "cfg4"), choices=list(WORKLOADS))
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-entries", type=float, default=4e7,
                    help="size of the CPU-baseline sample in matrix entries (~1 us/entry/pass on one core)")
    ap.add_argument("--engine", type=int, default=2, choices=[1, 2],
                    help="2 = table-driven tiled passes (default), 1 = CSR/CSC kernels evaluating every entry")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI, the real path) or gloo: REHEARSAL ONLY — exchanges staged through the host "
                         "so that several ranks can share one GPU (--same-device) on a 1-GPU box")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--exchange", default="lib", choices=["lib", "torch"],
                    help="N > 1: lib = the library's own RCCL communicator (cellector_comm_init_rank; ncclAllGather / ncclAllReduce "
                         "enqueued by the C++ loop, the default), torch = torch.distributed collectives on the bound exchange buffers")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=INT",
                    help="extra cellector_set_option pairs (ablations), e.g. --opt overlap=1")
    ap.add_argument("--no-expected", action="store_true",
                    help="skip the expected_log_likelihood diagnostic column (NOT the reference-equivalent step)")
    return ap.parse_args()


def roofline(args, kernel, launch_ms, units, b_alg, achieved, traffic, traffic_src, layout_bytes, lds_lookup_bytes,
             lds_stage_bytes, lds_peak_gbs, sq):
    """The dominant kernel against its ceilings.  `achieved` / `frac` are SURVEY 8(d)'s figure — ALGORITHMIC bytes (the
    reference-shaped 8 B per entry) over the launch time — which says how fast the kernel does the reference's work, not how
    busy the HBM is: engine 2 streams a 2.67 B/entry re-encoding, so the figure can exceed 1.  The honest utilisations
    are beside it: HBM from the bytes the layout holds and from the PMC counters, LDS from the lookup bytes."""
    t = launch_ms * 1e-3
    out = {"kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
           "frac_is": "algorithmic bytes (SURVEY 8(d): 8 B per entry + row pointers + outputs + alpha/beta) / launch time / 8 TB/s"
                      " = rate of reference-equivalent work, NOT an HBM utilisation (it can exceed 1: the kernel reads the same"
                      " entries from a 16-bit tiled layout of ~2.7 B each, see hbm_frac_measured / hbm_frac_layout and bound)",
           "algorithmic_bytes_per_launch": int(b_alg), "launch_ms": launch_ms, "entries_per_launch": int(units),
           "launch_ms_is": "mean of HIP event pairs recorded on the launch stream around every fourth launch inside the timed region",
           "traffic": traffic,
           "traffic_is": ("HBM bytes per launch from rocprofv3 PMC passes of this command, committed as " + str(traffic_src)
                          + " (not measured inside this run)") if traffic else None,
           "layout_bytes_per_launch": int(layout_bytes),
           "hbm_frac_layout": layout_bytes / t / 1e9 / HBM_PEAK_GBS if t > 0 else None,
           "hbm_frac_measured": traffic / t / 1e9 / HBM_PEAK_GBS if (traffic and t > 0) else None,
           # the same bytes (PMC traffic when known, else the layout's size) against what a streaming read achieves (6.3 TB/s)
           "frac_achievable": (traffic or layout_bytes) / t / 1e9 / HBM_ACHIEVABLE_GBS if t > 0 else None}
    if args.engine == 2:
        lds_frac = lds_lookup_bytes / t / 1e9 / lds_peak_gbs if t > 0 else None
        out["bound"] = "lds"
        out["lds"] = {"lookup_bytes_per_launch": int(lds_lookup_bytes), "table_staging_bytes_per_launch": int(lds_stage_bytes),
                      "peak": lds_peak_gbs, "unit": "GB/s",
                      "frac": lds_frac,  # conflict-free share of the LDS pipe's time that the lookups alone need
                      "sq_counters": sq,  # measured: LDS pipe busy / bank-conflict share / wait share (profiles/*_sq_tile.csv)
                      "note": "random 8-byte lookups: the measured LDS-busy share is ~2.6x the conflict-free figure (bank "
                              "conflicts); ablations of the kernel alone (DESIGN 3.3b): 1.0 ms without lookups and table re-staging (a wave's own "
                              "instruction stream and row loads, 4 waves/SIMD), + 0.22 re-staging, + 0.27 lookups of which 0.11 conflicts"}
        if out["frac"] > 1.0:
            out["frac_note"] = "above 1 because the tiled layout moves 1/3 of the algorithmic bytes; see hbm_frac_measured / lds"
    else:
        out["bound"] = "f64-alu"  # engine 1 evaluates ~300 f64 instructions per entry
    return out


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    from cellector_amd import Cellector, ffi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X: there is no CPU fallback for the hot path")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    N, L_total, density, continue_pct = WORKLOADS[args.workload]
    per = (N + world - 1) // world
    cb, ce = min(N, rank * per), min(N, (rank + 1) * per)

    def make_ctx():
        g = Cellector(local_rank, stream=torch.cuda.current_stream().cuda_stream)
        g.set_option("keep_coo", 0)
        g.set_option("engine", args.engine)
        g.set_option("compute_expected", 0 if args.no_expected else 1)
        for kv in args.opt:
            k, v = kv.split("=")
            g.set_option(k, int(v))
        g.set_option("synth_continue_pct", continue_pct)
        return g

    g = make_ctx()
    # N > 1: the exchanges run inside the library over its own RCCL communicator (rank 0's id is broadcast with the launcher's
    # process group); if RCCL cannot be set up on any rank, every rank falls back to torch.distributed collectives on the
    # bound exchange buffers and the line says so.
    lib_comm = world > 1 and args.exchange == "lib" and args.backend == "nccl"
    comm_note = None
    if lib_comm:
        ok, why = 1, ""
        # (every rank takes part in the broadcast whatever happened on rank 0: byte 128 says whether the id is valid)
        uid = torch.zeros(129, dtype=torch.uint8, device=dev)
        if rank == 0:
            try:
                raw = bytearray(ffi.comm_unique_id()) + bytearray([1])
                uid.copy_(torch.frombuffer(raw, dtype=torch.uint8))
            except Exception as e:  # noqa: BLE001
                ok, why = 0, str(e)
        dist.broadcast(uid, 0)
        uid_h = uid.cpu().numpy()
        if int(uid_h[128]) != 1:
            ok, why = 0, why or "rank 0 could not make a communicator id"
        else:
            try:
                g.comm_init_rank(bytes(uid_h[:128].tobytes()), world, rank)
            except Exception as e:  # noqa: BLE001
                ok, why = 0, str(e)
        flag = torch.tensor([float(ok)], dtype=torch.float64, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if flag.item() < 1.0:
            comm_note = "library communicator unavailable (%s): torch.distributed exchanges" % (why or "another rank failed")
            lib_comm = False
            g.close()
            g = make_ctx()
    if not lib_comm:
        g.set_shard(cb, ce)
        g.set_option("norm_zero", 0)  # the NORM slices are all-gathered below, not summed: no need to clear the others' first

    def allreduce(t, op=None):
        if world > 1:
            kw = {} if op is None else {"op": op}
            if args.backend == "nccl":
                dist.all_reduce(t, **kw)
            else:  # rehearsal: stage through the host
                h = t.cpu()
                dist.all_reduce(h, **kw)
                t.copy_(h)

    # ---- ingest (untimed setup): device-side generation, pass-1 exchange, CSR/CSC build
    t_setup = time.time()
    if lib_comm:
        g.ingest_synthetic(L_total, N, density, seed=args.seed, minority_fraction=0.05, doublet_fraction=0.0)
        g.ingest_finish(4, 4)  # (all-reduces PASS1 itself)
        dm = g.dims()
        L = dm.loci_used
        assert (dm.cell_begin, dm.cell_end) == (cb, ce)
    else:
        x_pass1 = torch.zeros(5 * L_total, dtype=torch.float64, device=dev)
        g.bind_exchange_buffer(ffi.XCHG_PASS1, x_pass1.data_ptr(), x_pass1.numel())
        g.ingest_synthetic(L_total, N, density, seed=args.seed, minority_fraction=0.05, doublet_fraction=0.0)
        allreduce(x_pass1)
        g.ingest_finish(4, 4)
        dm = g.dims()
        L = dm.loci_used
        x_norm = torch.zeros(world * per, dtype=torch.float64, device=dev)  # padded: every rank owns `per` slots
        x_locus = torch.zeros(5 * L + 8, dtype=torch.float64, device=dev)
        g.bind_exchange_buffer(ffi.XCHG_NORM, x_norm.data_ptr(), x_norm.numel())
        g.bind_exchange_buffer(ffi.XCHG_LOCUS, x_locus.data_ptr(), x_locus.numel())
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup

    nnz_local = dm.nnz_used
    nnz_t = torch.tensor([float(nnz_local)], dtype=torch.float64, device=dev)
    allreduce(nnz_t)
    nnz_total = int(nnz_t.item())

    def exchange_norm():
        # in-place all-gather of the equal-sized contiguous cell slices (half the bytes of a sum all-reduce)
        if world == 1:
            return
        if args.backend == "nccl":
            dist.all_gather_into_tensor(x_norm, x_norm[rank * per:(rank + 1) * per])
        else:
            h = x_norm.cpu()
            dist.all_gather_into_tensor(h, h[rank * per:(rank + 1) * per].clone())
            x_norm.copy_(h)

    def step():
        if lib_comm or world == 1:
            return g.em_iteration(5.0)  # the whole (distributed) iteration inside the library
        g.em_begin()
        exchange_norm()
        g.em_threshold(5.0)
        allreduce(x_locus)
        return g.em_finish()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    # HIP events around the dominant kernel only, every fourth launch of the timed region (an event pair idles the queue ~10 us:
    # 2 % of an iteration when recorded around every launch)
    g.set_option("timing", 3)
    g.reset_timing()
    fence()
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = step()
    fence()
    elapsed = time.perf_counter() - t0
    g.set_option("timing", 0)
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    allreduce(el, dist.ReduceOp.MAX)
    elapsed = float(el.item())
    ms_per_step = elapsed / args.steps * 1e3
    # the same K-step loop four more times (reported beside the headline, which stays the FIRST loop's elapsed / K): a timed
    # region of tens of milliseconds moves by percent with one queue hiccup, and min / median make that visible
    repeats = [ms_per_step]
    for _ in range(4):
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        e1 = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        allreduce(e1, dist.ReduceOp.MAX)
        repeats.append(float(e1.item()) / args.steps * 1e3)
    # every rank ran the same exact selection on the same all-gathered keys and saw the same all-reduced counters: the
    # iteration summaries must agree to the bit (checked, reported in the line)
    ranks_agree = True
    if world > 1:
        mine = torch.tensor([last.threshold, last.median, last.iqr, float(last.n_excluded), float(last.any_change)],
                            dtype=torch.float64, device=dev)
        lo, hi = mine.clone(), mine.clone()
        allreduce(lo, dist.ReduceOp.MIN)
        allreduce(hi, dist.ReduceOp.MAX)
        ranks_agree = bool(torch.equal(lo, hi))

    # ---- the dominant kernel's duration: HIP events on the launch stream, recorded inside the timed region
    dom_id = ffi.K_TILE_LL if args.engine == 2 else ffi.K_CELL_LL
    dom_total_ms, dom_n = g.kernel_time(dom_id)
    # ---- the other groups of kernels (breakdown only): the same steps once more, untimed, with every event pair on
    g.set_option("timing", 1)
    g.reset_timing()
    for _ in range(args.steps):
        step()
    fence()
    g.set_option("timing", 0)
    ll_ms, ll_n = g.kernel_time(ffi.K_CELL_LL)
    lo_ms, lo_n = g.kernel_time(ffi.K_LOCUS_STATS)
    se_ms, se_n = g.kernel_time(ffi.K_SELECT)
    ti_ms, ti_n = g.kernel_time(ffi.K_TILE_LL)
    ll_avg = ll_ms / max(ll_n, 1)
    n_loc = ce - cb
    # ---- the posterior phase (phases are timed separately): calculate_posteriors = three log-likelihood passes
    # with three alpha/beta sets over all loci + the per-cell finalize, on the device (no output copies) and as the host sees
    # it (4 x 8 B per cell copied back); untimed for the headline figure
    post_ms = {}
    for name, fetch in (("device", False), ("with_output_copy", True)):
        g.posteriors(fetch=fetch)
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            g.posteriors(fetch=fetch)
        fence()
        post_ms[name] = (time.perf_counter() - t1) / 3 * 1e3
    info = g.engine_info()
    # Dominant kernel and the entries ONE launch of it processes on this rank:
    #   engine 2: k_tile_ll, the regular entries (1 <= alt+ref <= 4); engine 1: k_cell_ll, all entries.
    # Algorithmic bytes: 8 B/entry (u32 locus + u16 alt + u16 ref) + u64 row pointers
    #   + f64 LL and u32 loci-used out per cell + alpha,beta read once per locus.
    if args.engine == 2:
        dom_kernel, dom_ms, dom_units = "k_tile_ll", dom_total_ms / max(dom_n, 1), info.nnz_regular
    else:
        dom_kernel, dom_ms, dom_units = "k_cell_ll", dom_total_ms / max(dom_n, 1), nnz_local
    b_pass = dom_units * 8 + (n_loc + 1) * 8 + n_loc * 12 + L * 16
    achieved = b_pass / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
    # What the kernel really streams (engine 2): the 16-bit tiles + slice headers, the per-group partial sums it writes,
    # the chunk tables once from HBM (they are re-staged into LDS out of L2 for every column of cell blocks), and what
    # its lookups ask of the LDS pipe: `tile_lookups` lookups (entries + row padding) x 2 x 8-byte reads.
    props = torch.cuda.get_device_properties(dev)
    n_cu, clock_hz = props.multi_processor_count, getattr(props, "clock_rate", 2_400_000) * 1e3  # (kHz; MI355X: 2.4 GHz)
    # MI355X_MICROARCH.md, LDS table: a conflict-free ds_read_b64 wave-instruction takes 2 LDS cycles = 256 B/clk/CU
    # (128 B/clk/CU for 4-byte reads and the ds_read2 forms); the lookups are ds_read_b64
    lds_peak_gbs = n_cu * 256.0 * clock_hz / 1e9
    reads_per_lookup = 1 if args.no_expected else 2
    if args.engine == 2:
        t_pad = info.cell_blocks * 1024
        table_bytes = info.locus_chunks * 640 * 18 * 8
        layout_bytes = info.tile_bytes + info.chunk_groups * t_pad * 8 * reads_per_lookup + table_bytes
        lds_lookup_bytes = info.tile_lookups * 8 * reads_per_lookup
        n_cols = (info.cell_blocks + 3) // 4
        lds_stage_bytes = n_cols * info.locus_chunks * 640 * 18 * 8  # every column re-stages its groups' tables (L2 -> LDS)
    else:
        layout_bytes, lds_lookup_bytes, lds_stage_bytes = b_pass, 0, 0

    out = None
    if rank == 0:
        traffic, traffic_src, sq = None, None, None
        tr_path = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tr_path):
            try:
                tr = json.load(open(tr_path))
                key = f"{args.workload}:n{world}:engine{args.engine}"
                if key in tr:
                    traffic = tr[key]["hbm_bytes_per_launch"]
                    traffic_src = tr[key].get("source")
                    sq = tr[key].get("sq")
            except Exception:
                traffic = None
        out = {
            "metric": "cell x locus log-likelihood evals/sec over full EM iterations (EM iters/sec in em_iters_per_s)",
            "value": nnz_total / (elapsed / args.steps),
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "ms_per_step_repeats": {"n": len(repeats), "min": min(repeats), "median": sorted(repeats)[len(repeats) // 2],
                                    "all": repeats, "note": "the K-step loop five times; ms_per_step is the first"},
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}: {N} cells x {L_total} loci, density {density}, seed {args.seed}, "
                                   f"5% minority, entry totals 1 + Geometric({1 - continue_pct / 100:.1f}); cells sharded "
                                   "contiguously across ranks",
                       "loci_used": int(L), "nnz_used": nnz_total,
                       "step": "one EM iteration = alpha/beta + cell LL pass"
                               + ("" if args.no_expected else " (+expected-log-pmf)")
                               + " + exact quartiles + flags + locus pass + locus filter",
                       "parallelism": (f"cells/{world}, " + ("RCCL all-gather + all-reduce per iteration inside the library "
                                                                             "(cellector_comm_init_rank)" if lib_comm else
                                                                             "torch.distributed exchanges on the bound buffers")
                                       if world > 1 else "1 gpu")
                                      + ("" if args.backend == "nccl" else " (gloo host-staged REHEARSAL, not a result)")
                                      + (f"; {comm_note}" if comm_note else "")},
            "em_iters_per_s": args.steps / elapsed,
            "dense_cells_x_loci_per_s": float(N) * float(L) / (elapsed / args.steps),
            "posterior_phase_ms": post_ms,
            "kernels_ms_note": "breakdown from a second, untimed run of the same steps with every event pair recorded; "
                               "roofline.launch_ms is from the timed region",
            "kernels_ms": {"cell_pass": ll_avg, "tile_ll": ti_ms / max(ti_n, 1), "locus_pass": lo_ms / max(lo_n, 1),
                           "select": se_ms / max(se_n, 1)},
            "ll_pass_evals_per_s": nnz_local * world / (ll_avg * 1e-3) if ll_avg > 0 else None,
            "roofline": roofline(args, dom_kernel, dom_ms, dom_units, b_pass, achieved, traffic, traffic_src, layout_bytes,
                                 lds_lookup_bytes, lds_stage_bytes, lds_peak_gbs, sq),
            "engine": {"engine": args.engine, "nnz_regular": int(info.nnz_regular),
                       "nnz_overflow": int(info.nnz_overflow), "tile_bytes": int(info.tile_bytes)},
            "setup_s": t_setup,
            "last_iteration": {"n_excluded": int(last.n_excluded), "threshold": last.threshold,
                               "any_change": int(last.any_change), "ranks_agree": ranks_agree},
        }

    # ---- CPU baseline: the oracle (port of the reference's single-threaded path) on a bounded cell sample
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import binding as ob
        avg_row = max(1.0, nnz_local / max(n_loc, 1))
        n_sample = int(max(1, min(n_loc, args.cpu_sample_entries / avg_row)))
        rp, ent = g.csr_rows(0, n_sample)
        lc = g.locus_counts()
        o = ob.Oracle.from_csr(L, rp, ent, lc)
        ob.set_threads(1)
        t1 = time.perf_counter()
        o.em_iteration(5.0)
        cpu_s = time.perf_counter() - t1
        out["cpu_baseline"] = {
            "value": float(len(ent)) / cpu_s, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": f"first {n_sample} cells ({len(ent)} entries) of the same matrix with the full matrix's "
                      f"alpha/beta, one EM iteration of the single-threaded C oracle in {cpu_s:.1f} s",
            "host_cpus": os.cpu_count(),
        }
        out["gpu_over_cpu_1core"] = out["value"] / out["cpu_baseline"]["value"]
        # the reference is single-threaded; the same port with its per-cell loop on all host cores, for scale
        nt = ob.host_threads(cap=256)
        if nt > 1:
            ob.set_threads(nt)
            o.em_iteration(5.0)  # spin up the pool
            t1 = time.perf_counter()
            o.em_iteration(5.0)
            cpu_mt = time.perf_counter() - t1
            ob.set_threads(1)
            out["cpu_baseline_all_cores"] = {"value": float(len(ent)) / cpu_mt, "unit": "evals/s", "cores": nt, "kind": "port",
                                             "sample": f"same sample, per-cell loop on {nt} threads (OpenMP), {cpu_mt:.2f} s"}
        o.close()
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out))
    g.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
