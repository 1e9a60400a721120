"""N > 1 host logic on CPU: cellector_amd.sharded.ShardedCellector over torch.distributed (gloo, world_size 2)
with a TEST-ONLY engine backed by the oracle.  Checks that two cell shards + the three sum-exchanges reproduce the
single-process oracle run (same thresholds, same exclusion set, same locus tallies, same posteriors).

The product engine (HipShardEngine) implements the same interface on the GPU; its shard arithmetic is covered on one
GPU by tests/test_gpu_parity.py::test_two_shards_equal_single_shard."""
import os
import socket
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

L_TOTAL, N_TOTAL, DENSITY, SEED = 900, 501, 0.12, 6


class OracleShardEngine:
    """Same interface as cellector_amd.sharded.HipShardEngine, computing with the CPU oracle (tests only)."""

    def __init__(self):
        import torch
        from oracle import binding as ob
        self.torch, self.ob = torch, ob

    def set_shard(self, cb, ce):
        self.cb, self.ce = cb, ce

    def ingest_coo(self, total_loci, total_cells, lo, ce, al, re):
        self.TL, self.N = total_loci, total_cells
        keep = (ce >= self.cb) & (ce < self.ce)
        self.coo = [x[keep] for x in (lo, ce, al, re)]
        lo, ce, al, re = self.coo
        p1 = np.zeros((5, total_loci))
        np.add.at(p1[0], lo, re > 0); np.add.at(p1[1], lo, al > 0)
        np.add.at(p1[2], lo, re); np.add.at(p1[3], lo, al); np.add.at(p1[4], lo, 1)
        self.pass1 = self.torch.from_numpy(p1.reshape(-1).copy())
        return self.pass1

    def ingest_finish(self, min_alt, min_ref, norm_len=0):
        p1 = self.pass1.numpy().reshape(5, self.TL)
        used = (p1[0] >= min_ref) & (p1[1] >= min_alt)
        self.locus_ids = np.nonzero(used)[0]
        self.L = len(self.locus_ids)
        to_used = np.full(self.TL, -1)
        to_used[self.locus_ids] = np.arange(self.L)
        lo, ce, al, re = self.coo
        k = to_used[lo] >= 0
        li, cl, al, re = to_used[lo][k], (ce[k] - self.cb).astype(np.int64), al[k], re[k]
        order = np.argsort(cl, kind="stable")  # per-cell lists in file order
        li, cl, al, re = li[order], cl[order], al[order], re[order]
        nloc = self.ce - self.cb
        row_ptr = np.zeros(nloc + 1, np.uint64)
        row_ptr[1:] = np.cumsum(np.bincount(cl, minlength=nloc))
        packed = li.astype(np.uint64) | (al.astype(np.uint64) << np.uint64(32)) | (re.astype(np.uint64) << np.uint64(48))
        self.s_ref, self.s_alt, self.n_ent = p1[2][used], p1[3][used], p1[4][used]
        self.o = self.ob.Oracle.from_csr(self.L, row_ptr, packed, np.stack([self.s_ref, self.s_alt], 1))
        self.li, self.al, self.re, self.cl = li, al, re, cl
        self.mask = np.ones(self.L, np.uint8)
        self.flags = np.zeros(nloc, np.uint8)
        self.locus = self.torch.zeros(5 * self.L + 8, dtype=self.torch.float64)
        self.norm = self.torch.zeros(max(self.N, norm_len), dtype=self.torch.float64)
        return types.SimpleNamespace(total_cells=self.N, total_loci=self.TL, loci_used=self.L, cell_begin=self.cb,
                                     cell_end=self.ce, nnz_used=len(li))

    def _planes(self):
        return self.locus.numpy()[:5 * self.L].reshape(5, self.L)

    def em_begin(self):
        pl = self._planes()
        self.alpha, self.beta = (self.s_alt + 1.0) - pl[3], (self.s_ref + 1.0) - pl[4]
        self.ll, self.ell, self.nl = self.o.cell_log_likelihoods(self.alpha, self.beta, self.mask)
        self.norm.zero_()
        self.norm.numpy()[self.cb:self.ce] = np.where(self.nl > 0, self.ll / np.where(self.nl > 0, self.nl, 1), 0.0)
        return self.norm

    def em_threshold(self, iqr):
        x = self.norm.numpy()[:self.N]
        self.median = self.ob.median(x)
        q1, q3 = self.ob.quantile(x, 0.25), self.ob.quantile(x, 0.75)
        self.iqr = q3 - q1
        self.thr = q1 - iqr * self.iqr
        self.new = (x[self.cb:self.ce] < self.thr).astype(np.uint8)
        st = self.o.locus_stats(self.alpha, self.beta, self.mask, self.new)
        buf = self.locus.numpy()
        buf[:] = 0
        pl = self._planes()
        pl[0], pl[1], pl[2] = st["contrib_min"], st["contrib_maj"], st["cells_min"]
        mino = self.new[self.cl] != 0  # mask-agnostic minority tallies
        pl[3] = np.bincount(self.li[mino], self.al[mino], self.L)
        pl[4] = np.bincount(self.li[mino], self.re[mino], self.L)
        buf[5 * self.L + 0] = int(((self.new == 1) & (self.flags == 0)).sum())
        buf[5 * self.L + 1] = int(((self.new == 0) & (self.flags == 1)).sum())
        buf[5 * self.L + 2] = int(self.new.sum())
        return self.locus

    def em_finish(self):
        buf = self.locus.numpy()
        pl = self._planes()
        per_cell = np.where(pl[2] != 0, pl[0] / np.where(pl[2] != 0, pl[2], 1), 0.0)
        filt = per_cell < -80.0
        self.mask_used = self.mask.copy()
        self.mask[filt] = 0
        self.flags = self.new
        n_new, n_res, n_exc = (int(buf[5 * self.L + k]) for k in range(3))
        self.n_excluded = n_exc
        return types.SimpleNamespace(any_change=int(n_new > 0 or n_res > 0), n_new_excluded=n_new, n_rescued=n_res,
                                     n_excluded=n_exc, n_loci_filtered=int(filt.sum()), median=self.median,
                                     iqr=self.iqr, threshold=self.thr)

    def excluded(self):
        return self.flags

    def posteriors(self):
        pl = self._planes()
        a_maj, b_maj = (self.s_alt + 1.0) - pl[3], (self.s_ref + 1.0) - pl[4]
        a_min, b_min = (self.s_alt + 1.0) - (self.s_alt - pl[3]), (self.s_ref + 1.0) - (self.s_ref - pl[4])
        mf0 = (self.n_excluded + 1.0) / (self.N + 1.0)
        a_dbl, b_dbl = (a_maj - 1.0) * mf0 + (a_min - 1.0) + 1.0, (b_maj - 1.0) * mf0 + (b_min - 1.0) + 1.0
        mf = max(mf0, 0.01)
        a_maj, b_maj = (a_maj - 1.0) * mf + 1.0, (b_maj - 1.0) * mf + 1.0
        l_min = self.o.cell_log_likelihoods(a_min, b_min)[0]
        l_maj = self.o.cell_log_likelihoods(a_maj, b_maj)[0]
        l_dbl = self.o.cell_log_likelihoods(a_dbl, b_dbl)[0]
        lse = self.ob.lib().orc_logsumexp
        lpd, lpm, lpM = np.log(self.N / 1000.0 / 100.0 * max(mf, 0.1)), np.log(mf), np.log(1.0 - mf)
        post = np.empty(len(l_min))
        for i in range(len(l_min)):
            num = lpm + l_min[i]
            den = lse(lse(num, lpM + l_maj[i]), lpd + l_dbl[i])
            post[i] = np.exp(num - den)
        return dict(posterior=post, ll_majority=l_maj, ll_minority=l_min)


def _worker(rank, world, port, outdir):
    import torch.distributed as dist
    from cellector_amd import synth
    from cellector_amd.sharded import ShardedCellector, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    coo = synth.generate_coo(L_TOTAL, N_TOTAL, DENSITY, seed=SEED, minority_fraction=0.08)
    sc = ShardedCellector(OracleShardEngine(), dist)
    dims = sc.load(lambda e: e.ingest_coo(L_TOTAL, N_TOTAL, *coo), N_TOTAL)
    assert (dims.cell_begin, dims.cell_end) == shard_range(N_TOTAL, rank, world)
    summaries = sc.run(5.0, max_iter=30)
    post = sc.e.posteriors()
    excl = sc.gather_cells(sc.e.excluded(), np.uint8)
    posterior = sc.gather_cells(post["posterior"])
    np.savez(os.path.join(outdir, f"rank{rank}.npz"), excluded=excl, posterior=posterior,
             thresholds=np.array([s.threshold for s in summaries]), medians=np.array([s.median for s in summaries]),
             n_new=np.array([s.n_new_excluded for s in summaries]), n_exc=np.array([s.n_excluded for s in summaries]),
             loci_used=dims.loci_used, alt_min=sc.e._planes()[3], cells_min=sc.e._planes()[2],
             contrib_min=sc.e._planes()[0], mask=sc.e.mask)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_ranges_cover_all_cells():
    from cellector_amd.sharded import shard_range
    for n, w in ((501, 2), (1000, 8), (7, 8), (1_000_000, 8)):
        r = [shard_range(n, k, w) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n and all(a[1] == b[0] for a, b in zip(r, r[1:]))


def test_two_gloo_ranks_reproduce_the_single_process_run(tmp_path, oracle_lib):
    import torch.multiprocessing as mp
    from cellector_amd import synth
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    coo = synth.generate_coo(L_TOTAL, N_TOTAL, DENSITY, seed=SEED, minority_fraction=0.08)
    o = oracle_lib.Oracle.from_coo(L_TOTAL, N_TOTAL, *coo)
    ref = o.run(5.0, max_iter=30)
    po = o.posteriors()
    lo_ = o.locus_outputs()
    r0, r1 = (np.load(os.path.join(tmp_path, f"rank{k}.npz")) for k in range(world))
    for r in (r0, r1):
        assert int(r["loci_used"]) == o.loci_used
        assert len(r["thresholds"]) == len(ref) and len(ref) >= 2
        np.testing.assert_allclose(r["thresholds"], [s.threshold for s in ref], rtol=0, atol=1e-12)
        np.testing.assert_allclose(r["medians"], [s.median for s in ref], rtol=0, atol=1e-12)
        assert r["n_new"].tolist() == [s.n_new_excluded for s in ref]
        assert r["n_exc"][-1] == o.excluded().sum()
        assert np.array_equal(r["excluded"], o.excluded()) and r["excluded"].sum() > 10
        np.testing.assert_allclose(r["posterior"], po["posterior"], rtol=0, atol=1e-9)
        assert np.array_equal(r["alt_min"], lo_["alt_min"].astype(float))      # exact: integer tallies
        assert np.array_equal(r["cells_min"], lo_["cells_min"].astype(float))
        np.testing.assert_allclose(r["contrib_min"], lo_["contrib_min"], rtol=0, atol=1e-9)  # f64 sums: order differs
        assert np.array_equal(r["mask"], o.loci_mask())
    # both ranks hold identical global state after the exchanges
    for k in ("thresholds", "excluded", "alt_min", "contrib_min"):
        assert np.array_equal(r0[k], r1[k])
