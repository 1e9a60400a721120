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


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0, 0]], ids=["2shards", "5shards"])
def test_multi_device_ctx_equals_single_ctx_and_oracle(devices, oracle_lib, hip_lib_path, tmp_path):
    """cellector_create_multi: ONE ctx over several shards, exchanges inside the library.  With a device listed more than
    once the shards are logical shards on one GPU (device-side sums instead of RCCL): the same sharded code path on a
    one-GPU box.  Every entry point must answer for the whole matrix, in global cell order, like a single-device ctx —
    checked against one and against the oracle, from COO and from text files."""
    from cellector_amd import Cellector, ffi, synth
    import test_gpu_parity as T
    L, N, d = 1500, 1201, 0.1   # 1201 cells: shards of unequal size, the last one ragged
    coo = synth.generate_coo(L, N, d, seed=4, minority_fraction=0.06, doublet_fraction=0.01)
    a_path, r_path = synth.write_mtx_pair(str(tmp_path), L, N, *coo)
    # (bank_order = 0: rows in file order, the layout whose per-cell sums do not depend on the sharding — this test compares bits;
    #  the default order is compared within rounding in test_nnz_balanced_partition_on_a_skewed_matrix)
    single = Cellector(0)
    single.set_option("bank_order", 0)
    single.load_coo(L, N, *coo)
    for source in ("coo", "mtx"):
        m = Cellector(devices=devices)
        m.set_option("bank_order", 0)
        # exchange point 2 both ways per shard count: digit histograms, or NORM all-gather + select over all keys
        m.set_option("sharded_select", 1 if source == "coo" else 0)
        if source == "coo":
            m.load_coo(L, N, *coo)
        else:
            # the text is tokenised once and its entries routed to the shards; with GPUs to spare the ref file is tokenised
            # on a second one meanwhile — forced here so that a one-GPU box runs that path too (5-shard case)
            if len(devices) > 2:
                os.environ["CELLECTOR_MULTI_REF_HELPER"] = "1"
                m.set_option("parse_window", 4096)  # ... through the windowed parser (files of 1 GB and more take it)
            try:
                m.load_mtx(a_path, r_path)
            finally:
                os.environ.pop("CELLECTOR_MULTI_REF_HELPER", None)
        o = oracle_lib.Oracle.from_coo(L, N, *coo)
        T._check_matrix(m, o)                      # dims, locus ids/counts, entries per cell, CSR rows across the shards
        dm = m.dims()
        assert (dm.cell_begin, dm.cell_end, dm.nnz_used) == (0, N, single.dims().nnz_used)
        rp, ent = m.csr_rows(395, 1003)            # a range that straddles shard boundaries
        rs, es = single.csr_rows(395, 1003)
        assert np.array_equal(rp, rs) and np.array_equal(ent, es)
        if source == "coo":
            s_iters = []
        for it in range(30):
            sm, so = m.em_iteration(5.0), o.em_iteration(5.0)
            T._check_iteration(m, o, sm, so)       # against the oracle
            if source == "coo":
                s1 = single.em_iteration(5.0)
                s_iters.append((s1, single.cell_outputs(), single.locus_outputs(), single.excluded()))
            s1, c1, l1, e1 = s_iters[it]
            assert (sm.any_change, sm.n_new_excluded, sm.n_rescued, sm.n_excluded, sm.n_loci_filtered, sm.n_near_threshold) == \
                   (s1.any_change, s1.n_new_excluded, s1.n_rescued, s1.n_excluded, s1.n_loci_filtered, s1.n_near_threshold)
            assert (sm.median, sm.iqr, sm.threshold) == (s1.median, s1.iqr, s1.threshold)
            cm = m.cell_outputs()
            for k in c1:
                assert np.array_equal(cm[k], c1[k]), k
            assert np.array_equal(m.excluded(), e1)
            lm = m.locus_outputs()
            for k in l1:
                if k.startswith("contrib"):
                    np.testing.assert_allclose(lm[k], l1[k], rtol=0, atol=1e-8)
                else:
                    assert np.array_equal(lm[k], l1[k]), k
            if not so.any_change:
                break
        T._check_posteriors(dict(ffi=ffi), m, o)
        if source == "coo":
            p1, t1 = single.posteriors(), single.final_allele_tallies()
        pm, tm = m.posteriors(), m.final_allele_tallies()
        for k in p1:
            assert np.array_equal(pm[k], p1[k]), k
        for k in t1:
            assert np.array_equal(tm[k], t1[k]), k
        rng = np.random.default_rng(1)
        alpha, beta = rng.uniform(0.5, 900.0, dm.loci_used), rng.uniform(0.5, 900.0, dm.loci_used)
        for x, y in zip(m.cell_log_likelihoods(alpha, beta), single.cell_log_likelihoods(alpha, beta)):
            assert np.array_equal(x, y)
        info = m.engine_info()
        assert info.nnz_regular + info.nnz_overflow == dm.nnz_used
        # what a multi-device ctx does not offer says so
        for call in (lambda: m.set_shard(0, 10), lambda: m.em_begin(), lambda: m.exchange_buffer(ffi.XCHG_NORM),
                     lambda: m.set_stream(0)):
            with pytest.raises(ffi.CellectorError):
                call()
        m.close(); o.close()
    single.close()


def _skewed_coo(L, N, seed):
    """10 % of the cells (the first ones: the worst case for equal ranges) carry 50 % of the entries"""
    rng = np.random.default_rng(seed)
    dens = np.where(np.arange(N) < N // 10, 0.18, 0.02)
    p, q = rng.uniform(0.05, 0.95, L), rng.uniform(0.05, 0.95, L)
    minority = rng.random(N) < 0.08
    lo, ce, al, re = [], [], [], []
    for l in range(L):
        cells = np.nonzero(rng.random(N) < dens)[0].astype(np.uint32)
        tot = rng.geometric(0.7, len(cells)).astype(np.uint32)
        a = rng.binomial(tot, np.where(minority[cells], q[l], p[l])).astype(np.uint32)
        lo.append(np.full(len(cells), l, np.uint32)); ce.append(cells); al.append(a); re.append(tot - a)
    return tuple(np.concatenate(x) for x in (lo, ce, al, re))


def test_nnz_balanced_partition_on_a_skewed_matrix(oracle_lib, hip_lib_path, tmp_path):
    """contiguous cell ranges "balanced by nnz, not by cell count".  The ingest of a multi-device ctx cuts the
    ranges from the entries per cell — COO, routed text and split text ingest alike: per-shard entry counts within 5 % of
    each other on a matrix whose first tenth of the cells holds half of the entries (equal counts: 2.6x between shards) —
    and the run is the single-device run (labels, thresholds) whatever the ranges; explicit ranges (cellector_set_partition)
    and option balance = 0 give what they say."""
    from cellector_amd import Cellector, ffi, synth
    import test_gpu_parity as T
    L, N, n = 900, 6000, 4
    coo = _skewed_coo(L, N, seed=12)
    epc = np.bincount(coo[1], minlength=N)
    assert 0.45 < epc[: N // 10].sum() / epc.sum() < 0.55
    a_path, r_path = synth.write_mtx_pair(str(tmp_path), L, N, *coo)
    single = Cellector(0)
    single.load_coo(L, N, *coo)
    runs = single.run(5.0, 30)
    want = (single.excluded(), [(s.threshold, s.n_excluded) for s in runs], single.posteriors())

    def shard_nnz(m):
        b = m.partition()
        assert len(b) == n + 1 and b[0] == 0 and b[-1] == N and np.all(np.diff(b.astype(np.int64)) >= 0)
        cs = np.concatenate(([0], np.cumsum(m.entries_per_cell())))
        return b, np.diff(cs[b.astype(np.int64)])

    def same_run(m):
        got = m.run(5.0, 30)
        assert np.array_equal(m.excluded(), want[0])
        assert [s.n_excluded for s in got] == [x[1] for x in want[1]]
        np.testing.assert_allclose([s.threshold for s in got], [x[0] for x in want[1]], rtol=1e-12, atol=0)
        p = m.posteriors()
        np.testing.assert_allclose(p["posterior"], want[2]["posterior"], rtol=0, atol=1e-9)

    for source in ("coo", "mtx", "mtx-split"):
        m = Cellector(devices=[0] * n)
        if source == "mtx-split":
            os.environ["CELLECTOR_MULTI_SPLIT"] = "1"
            m.set_option("parse_window", 4096)
        try:
            m.load_coo(L, N, *coo) if source == "coo" else m.load_mtx(a_path, r_path)
        finally:
            os.environ.pop("CELLECTOR_MULTI_SPLIT", None)
        b, nz = shard_nnz(m)
        assert nz.sum() == single.dims().nnz_used
        assert nz.max() <= 1.05 * nz.min(), (source, b, nz)
        assert b[1] < N // 8  # (the heavy cells: the first range is short)
        o = oracle_lib.Oracle.from_coo(L, N, *coo)
        T._check_matrix(m, o)
        o.close()
        same_run(m)
        m.close()
    # equal counts on request: the first shard then holds most of the heavy cells
    m = Cellector(devices=[0] * n)
    m.set_option("balance", 0)
    m.load_coo(L, N, *coo)
    b, nz = shard_nnz(m)
    assert np.array_equal(b, [0, 1500, 3000, 4500, 6000]) and nz.max() > 2 * nz.min()
    same_run(m)
    m.close()
    # explicit ranges, an empty one among them
    m = Cellector(devices=[0] * n)
    m.set_partition([0, 100, 100, 4000, N])
    m.load_mtx(a_path, r_path)
    b, nz = shard_nnz(m)
    assert np.array_equal(b, [0, 100, 100, 4000, N]) and nz[1] == 0
    same_run(m)
    with pytest.raises(ffi.CellectorError):
        m.set_partition([0, 1, 2, 3, N])  # (after the ingest)
    m.close()
    m = Cellector(devices=[0] * n)
    for bad in ([0, 10, 5, 20, N], [1, 2, 3, 4, N], [0, 1, N]):
        with pytest.raises(ffi.CellectorError):
            m.set_partition(bad)
    m.set_partition([0, 10, 20, 30, N - 1])  # does not cover the matrix: refused at the ingest
    with pytest.raises(ffi.CellectorError):
        m.load_coo(L, N, *coo)
    m.close()
    single.close()


def test_split_text_ingest_every_shard_parses_a_byte_range(hip_lib_path, tmp_path):
    """From three devices on, every GPU of a multi-device ctx tokenises 1/n of BOTH files' bytes (window ranges), the shards
    line the files up by global line number, zip, validate and route every entry to the shard that owns its cell
    (kernels_parse.hip: ingest_stage_mtx_split).  CELLECTOR_MULTI_SPLIT=1 runs it on logical shards of one GPU with tiny windows:
    alt and ref ranges never cut at the same lines (different header, different digit counts), the last line is unterminated,
    late shards get no window at all for a short file.  The staged matrix (file order inside every cell), the EM results and the
    error reports must be those of a single-device ctx."""
    from cellector_amd import Cellector, ffi, synth
    L, N = 700, 901
    lo, ce, al, re = synth.generate_coo(L, N, 0.05, seed=23, minority_fraction=0.08)
    al = al.copy(); al[::7] += 12345  # longer tokens in one file than in the other
    a_path, r_path = synth.write_mtx_pair(str(tmp_path), L, N, lo, ce, al, re)
    with open(a_path, "rb") as f:
        body = f.read()
    with open(a_path, "wb") as f:
        f.write(body.rstrip(b"\n"))
    ref_lines = open(r_path).read().split("\n", 3)
    open(r_path, "w").write(ref_lines[0] + "\n% a longer comment line than the alt file has, to shift every window\n" + ref_lines[2] + "\n" + ref_lines[3])
    single = Cellector(0)
    single.set_option("bank_order", 0)  # (rows in file order: thresholds are compared bit for bit across shardings below)
    single.load_mtx(a_path, r_path, 2, 2)
    rs, es = single.csr_rows(0, N)
    it_single = [single.em_iteration(5.0) for _ in range(3)]
    os.environ["CELLECTOR_MULTI_SPLIT"] = "1"
    try:
        for n_shards, window in ((3, 4096), (5, 512), (7, 1 << 15), (2, 640)):
            m = Cellector(devices=[0] * n_shards)
            m.set_option("bank_order", 0)
            m.set_option("parse_window", window)
            m.load_mtx(a_path, r_path, 2, 2)
            dm, ds = m.dims(), single.dims()
            assert (dm.total_loci, dm.total_cells, dm.loci_used, dm.nnz_used) == (ds.total_loci, ds.total_cells, ds.loci_used, ds.nnz_used)
            assert np.array_equal(m.locus_ids(), single.locus_ids())
            assert np.array_equal(m.locus_counts(), single.locus_counts())
            rm, em = m.csr_rows(0, N)
            assert np.array_equal(rm, rs) and np.array_equal(em, es), (n_shards, window)
            for s1 in it_single:
                sm = m.em_iteration(5.0)
                assert (sm.threshold, sm.median, sm.iqr, sm.n_excluded, sm.n_loci_filtered) == \
                       (s1.threshold, s1.median, s1.iqr, s1.n_excluded, s1.n_loci_filtered)
            m.close()
        # a file that is not locus-major, and a ref file shorter than the alt file (the zip stops there, load_data.rs:190)
        rng = np.random.default_rng(5)
        perm = rng.permutation(len(lo))
        b_alt, b_ref = synth.write_mtx_pair(str(tmp_path / "perm"), L, N, lo[perm], ce[perm], al[perm], re[perm])
        keep = len(lo) - 777
        with open(b_ref) as f:
            lines = f.read().split("\n")
        open(b_ref, "w").write("\n".join(lines[:3 + keep]) + "\n")
        single2 = Cellector(0)
        single2.load_mtx(b_alt, b_ref, 2, 2)
        m = Cellector(devices=[0] * 4)
        m.set_option("parse_window", 2048)
        m.load_mtx(b_alt, b_ref, 2, 2)
        assert m.dims().nnz_used == single2.dims().nnz_used
        r1, e1 = single2.csr_rows(0, N)
        r2, e2 = m.csr_rows(0, N)
        assert np.array_equal(r1, r2) and np.array_equal(e1, e2)
        s1, s2 = single2.em_iteration(5.0), m.em_iteration(5.0)
        assert (s1.threshold, s1.n_excluded) == (s2.threshold, s2.n_excluded)
        single2.close()
        # errors: same status and entry number as the single-device parser, from whichever shard meets them; nobody hangs
        many = "".join(f"1 {1 + i % 3} 1\n" for i in range(1500))
        cases = {"float": (many + "1 1 1.0\n" + many, 3, 1500), "two_tokens": (many + many + "1 1\n", 3, 3000),
                 "index0": (many + "0 1 1\n" + many, 1, 1500), "cell_range": ("1 4 1\n" + many + many, 1, 0),
                 "count_range": (many + many + "1 1 70000\n", 1, 3000), "first_of_two": (many + "x\n" + many + "y\n", 3, 1500)}
        for name, (text, status, where) in cases.items():
            bad = tmp_path / f"split_{name}.mtx"
            bad.write_text("%%MatrixMarket\n%\n2 3 0\n" + text)
            with pytest.raises(ffi.CellectorError) as ei:
                m.load_mtx(str(bad), str(bad), 1, 1)
            assert ei.value.status == status and f"entry {where}" in str(ei.value), (name, str(ei.value))
        # ... and the ctx is still usable afterwards
        m.load_mtx(a_path, r_path, 2, 2)
        rm, em = m.csr_rows(0, N)
        assert np.array_equal(rm, rs) and np.array_equal(em, es)
        m.close()
    finally:
        os.environ.pop("CELLECTOR_MULTI_SPLIT", None)
    single.close()


def test_more_shards_than_cells_and_errors(hip_lib_path, tmp_path):
    """shards without cells, and an error inside one shard must come back as an error (not hang the others)"""
    from cellector_amd import Cellector, ffi, synth
    coo = synth.generate_coo(60, 3, 0.9, seed=2, minority_fraction=0.3)
    m, single = Cellector(devices=[0] * 5), Cellector(0)
    m.load_coo(60, 3, *coo, 1, 1)
    single.load_coo(60, 3, *coo, 1, 1)
    for _ in range(3):
        sm, s1 = m.em_iteration(5.0), single.em_iteration(5.0)
        assert (sm.threshold, sm.n_excluded) == (s1.threshold, s1.n_excluded)
        assert np.array_equal(m.cell_outputs()["ll"], single.cell_outputs()["ll"])
    with pytest.raises(ffi.CellectorError, match="couldn't open file"):
        m.load_mtx(str(tmp_path / "nope.mtx"), str(tmp_path / "nope.mtx"))
    m.close(); single.close()
    with pytest.raises(ffi.CellectorError):
        Cellector(devices=[0, 99])


def test_rccl_transport_single_rank_selftest(hip_lib_path):
    """The RCCL transport (cellector_comm_init_rank: one process per GPU) with ONE rank: librccl is loaded, a communicator
    made from cellector_comm_unique_id's bytes, and the three exchanges run as real ncclAllReduce / ncclAllGather calls on the
    ctx's stream — all a one-GPU box can execute of that path.  Results must equal a ctx without a communicator."""
    from cellector_amd import Cellector, ffi, synth
    L, N, d = 900, 700, 0.12
    coo = synth.generate_coo(L, N, d, seed=9, minority_fraction=0.08)
    os.environ["CELLECTOR_COMM_SELFTEST"] = "1"
    try:
        g = Cellector(0)
        uid = ffi.comm_unique_id()
        assert len(uid) == 128 and any(uid)
        g.comm_init_rank(uid, 1, 0)
    finally:
        os.environ.pop("CELLECTOR_COMM_SELFTEST", None)
    with pytest.raises(ffi.CellectorError):
        g.set_shard(0, 10)                      # the communicator decides the cell range
    g.load_coo(L, N, *coo)
    with Cellector(0) as p:
        p.load_coo(L, N, *coo)
        sg, sp = g.run(5.0, 30), p.run(5.0, 30)
        assert [(a.threshold, a.n_excluded, a.any_change) for a in sg] == [(b.threshold, b.n_excluded, b.any_change) for b in sp]
        cg, cp = g.cell_outputs(), p.cell_outputs()
        for k in cg:
            assert np.array_equal(cg[k], cp[k]), k
        lg, lp = g.locus_outputs(), p.locus_outputs()
        for k in lg:
            assert np.array_equal(lg[k], lp[k]), k
        assert np.array_equal(g.posteriors()["posterior"], p.posteriors()["posterior"])
    g.close()


def test_sharded_order_statistics_on_adversarial_keys(oracle_lib, hip_lib_path):
    """A ctx with a communicator finds the median / quartiles without gathering the keys: every shard histograms its own
    keys digit by digit and the counts are all-reduced (kernels_select.hip: select_threshold_sharded).  On keys no EM run
    produces — ties, one shared prefix, every exponent, tiny n, shards without a key — the result must be bit-identical to the
    oracle's statrs restatement and to the single-device select; through the device-side sums of logical shards and through
    RCCL with one rank."""
    from cellector_amd import Cellector, ffi
    ob = oracle_lib
    rng = np.random.default_rng(7)
    cases = [
        np.array([-0.25]), np.array([3.0, -1.0]), np.array([0.0, -0.0, -1e-310]), np.array([-1.0, -1.0, -1.0, -2.0]),
        np.full(5000, -0.5), np.where(rng.random(5000) < 0.5, -0.5, -0.75), -0.5 - np.arange(40000) * 1e-13,
        -rng.uniform(0.25, 1.0, 40000), rng.standard_normal(40000) * 10.0 ** rng.integers(-300, 300, 40000),
        np.round(rng.standard_normal(40000), 1), -np.exp(rng.standard_normal(300000) * 0.2) * 0.5,
        np.concatenate([np.zeros(100000), -rng.uniform(0.4, 0.6, 200001)]),
        np.sort(rng.standard_normal(100003)),          # sorted: every shard holds one range of values
        np.concatenate([np.full(70000, 1.5), rng.standard_normal(30000)]),  # the first shards hold nothing but one value
    ]
    single = Cellector(0)
    os.environ["CELLECTOR_COMM_SELFTEST"] = "1"
    try:
        one_rank = Cellector(0)
        one_rank.comm_init_rank(ffi.comm_unique_id(), 1, 0)
    finally:
        os.environ.pop("CELLECTOR_COMM_SELFTEST", None)
    groups = [Cellector(devices=[0] * k) for k in (2, 3, 7)]
    for keys in cases:
        keys = np.ascontiguousarray(keys, dtype=np.float64)
        q1, q3 = ob.quantile(keys, 0.25), ob.quantile(keys, 0.75)
        want = (ob.median(keys), q3 - q1, q1 - 5.0 * (q3 - q1))
        assert single.order_statistics(keys, 5.0) == want, len(keys)
        assert one_rank.order_statistics(keys, 5.0) == want, len(keys)
        for m in groups:
            assert m.order_statistics(keys, 5.0) == want, len(keys)
    for g in groups + [single, one_rank]:
        g.close()


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
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and d["roofline"]["bound"] == "lds" and d["roofline"]["hbm_frac_layout"] > 0 and d["roofline"]["lds"]["frac"] > 0
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert "workload" in d["config"] and "model" not in d["config"]
