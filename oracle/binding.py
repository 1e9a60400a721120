"""ctypes binding of the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg — never from cellector_amd/ or host/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("cellector_oracle.c", "cellector_oracle.h", "Makefile")]
    if (not force and os.path.exists(_LIB)
            and os.path.getmtime(_LIB) >= max(os.path.getmtime(s) for s in src)):
        return _LIB
    subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB


class IterSummary(C.Structure):
    _fields_ = [("any_change", C.c_int), ("n_new_excluded", C.c_uint64), ("n_rescued", C.c_uint64),
                ("n_loci_filtered", C.c_uint64), ("median", C.c_double), ("iqr", C.c_double),
                ("threshold", C.c_double), ("locus_median", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB):
        build()
    L = C.CDLL(_LIB)
    d, u64, vp, cp = C.c_double, C.c_uint64, C.c_void_p, C.c_char_p
    sig = {
        "orc_ln_gamma": (d, [d]), "orc_ln_factorial": (d, [u64]), "orc_ln_binomial": (d, [u64, u64]),
        "orc_median": (d, [vp, C.c_size_t]), "orc_quantile": (d, [vp, C.c_size_t, d]),
        "orc_binomial_pmf": (d, [d, u64, u64]), "orc_logsumexp": (d, [d, d]),
        "orc_log_beta_calc": (d, [d, d]), "orc_log_beta_binomial_pmf": (d, [d, d, d, d, d]),
        "orc_expected_log_beta_binomial_pmf": (None, [C.c_size_t, d, d, vp, vp]),
        "orc_load_mtx": (vp, [cp, cp, u64, u64, vp, C.c_size_t]),
        "orc_from_coo": (vp, [u64, u64, u64, vp, vp, vp, vp, u64, u64, vp, C.c_size_t]),
        "orc_from_csr": (vp, [u64, u64, vp, vp, vp]),
        "orc_free": (None, [vp]), "orc_dims": (None, [vp, vp, vp, vp, vp]),
        "orc_locus_ids": (None, [vp, vp]), "orc_locus_counts": (None, [vp, vp]),
        "orc_entries_per_cell": (None, [vp, vp]), "orc_row_ptr": (None, [vp, vp]),
        "orc_entries": (None, [vp, vp, vp, vp, vp]),
        "orc_em_iteration": (None, [vp, d, vp]),
        "orc_iter_cell_outputs": (None, [vp, vp, vp, vp, vp]),
        "orc_iter_locus_outputs": (None, [vp] + [vp] * 8),
        "orc_loci_mask": (None, [vp, vp]), "orc_excluded": (None, [vp, vp]),
        "orc_set_excluded": (None, [vp, vp]), "orc_alpha_betas": (None, [vp, vp, vp]),
        "orc_cell_log_likelihoods": (None, [vp] + [vp] * 6),
        "orc_posteriors": (None, [vp, vp, vp, vp, vp]),
        "orc_locus_stats": (None, [vp] + [vp] * 12),
        "orc_assignments": (None, [vp, vp, vp, d, u64, vp, vp, vp]),
        "orc_final_tallies_coo": (None, [u64, u64] + [vp] * 9),
        "orc_vcf_genotype": (None, [u64, u64, u64, u64, vp, vp, vp, vp]),
        "orc_set_threads": (None, [C.c_int]), "orc_get_threads": (C.c_int, []),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def set_threads(n):
    """Worker threads of the oracle's per-cell loop (default 1, like the single-threaded reference).  Results are
    bit-identical for any count; the large parity tests use the host's cores so that they finish in seconds."""
    lib().orc_set_threads(int(n))


def host_threads(cap=32):
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(cap, n))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """One loaded matrix + loop state (mirrors the reference's main())."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError("oracle: null handle")
        self.h = C.c_void_p(handle)
        tc, tl, lu, nz = (C.c_uint64() for _ in range(4))
        lib().orc_dims(self.h, C.byref(tc), C.byref(tl), C.byref(lu), C.byref(nz))
        self.total_cells, self.total_loci, self.loci_used, self.nnz = tc.value, tl.value, lu.value, nz.value

    # ---- constructors
    @classmethod
    def from_mtx(cls, alt_path, ref_path, min_alt=4, min_ref=4):
        err = C.create_string_buffer(512)
        h = lib().orc_load_mtx(str(alt_path).encode(), str(ref_path).encode(), min_alt, min_ref, err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return cls(h)

    @classmethod
    def from_coo(cls, total_loci, total_cells, locus0, cell0, alt, ref, min_alt=4, min_ref=4):
        arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (locus0, cell0, alt, ref)]
        err = C.create_string_buffer(512)
        h = lib().orc_from_coo(total_loci, total_cells, len(arrs[0]), *[_p(a) for a in arrs],
                               min_alt, min_ref, err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return cls(h)

    @classmethod
    def from_csr(cls, n_loci, row_ptr, packed, locus_counts):
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint64)
        packed = np.ascontiguousarray(packed, dtype=np.uint64)
        locus_counts = np.ascontiguousarray(locus_counts, dtype=np.float64)
        assert locus_counts.shape == (n_loci, 2)
        return cls(lib().orc_from_csr(len(row_ptr) - 1, n_loci, _p(row_ptr), _p(packed), _p(locus_counts)))

    def close(self):
        if self.h:
            lib().orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- data accessors
    def locus_ids(self):
        out = np.empty(self.loci_used, np.uint64); lib().orc_locus_ids(self.h, _p(out)); return out

    def locus_counts(self):
        out = np.empty((self.loci_used, 2), np.float64); lib().orc_locus_counts(self.h, _p(out)); return out

    def entries_per_cell(self):
        out = np.empty(self.total_cells, np.uint32); lib().orc_entries_per_cell(self.h, _p(out)); return out

    def row_ptr(self):
        out = np.empty(self.total_cells + 1, np.uint64); lib().orc_row_ptr(self.h, _p(out)); return out

    def entries(self):
        li = np.empty(self.nnz, np.uint32); a = np.empty(self.nnz, np.uint32)
        r = np.empty(self.nnz, np.uint32); lnc = np.empty(self.nnz, np.float64)
        lib().orc_entries(self.h, _p(li), _p(a), _p(r), _p(lnc))
        return li, a, r, lnc

    # ---- loop
    def em_iteration(self, iqr_multiple=5.0):
        s = IterSummary()
        lib().orc_em_iteration(self.h, iqr_multiple, C.byref(s))
        return s

    def cell_outputs(self):
        n = self.total_cells
        ll, ell, nl, norm = (np.empty(n, np.float64) for _ in range(4))
        lib().orc_iter_cell_outputs(self.h, _p(ll), _p(ell), _p(nl), _p(norm))
        return dict(ll=ll, expected_ll=ell, loci_used=nl, normalized=norm)

    def locus_outputs(self):
        L = self.loci_used
        cm, cj = np.empty(L, np.float64), np.empty(L, np.float64)
        ints = [np.empty(L, np.uint64) for _ in range(6)]
        lib().orc_iter_locus_outputs(self.h, _p(cm), _p(cj), *[_p(a) for a in ints])
        keys = ["cells_min", "cells_maj", "alt_min", "ref_min", "alt_maj", "ref_maj"]
        return dict(contrib_min=cm, contrib_maj=cj, **dict(zip(keys, ints)))

    def loci_mask(self):
        out = np.empty(self.loci_used, np.uint8); lib().orc_loci_mask(self.h, _p(out)); return out

    def excluded(self):
        out = np.empty(self.total_cells, np.uint8); lib().orc_excluded(self.h, _p(out)); return out

    def set_excluded(self, flags):
        flags = np.ascontiguousarray(flags, dtype=np.uint8)
        assert len(flags) == self.total_cells
        lib().orc_set_excluded(self.h, _p(flags))

    def alpha_betas(self):
        a, b = np.empty(self.loci_used, np.float64), np.empty(self.loci_used, np.float64)
        lib().orc_alpha_betas(self.h, _p(a), _p(b))
        return a, b

    def cell_log_likelihoods(self, alpha, beta, mask=None):
        n = self.total_cells
        alpha = np.ascontiguousarray(alpha, np.float64); beta = np.ascontiguousarray(beta, np.float64)
        mask = np.ones(self.loci_used, np.uint8) if mask is None else np.ascontiguousarray(mask, np.uint8)
        ll, ell, nl = (np.empty(n, np.float64) for _ in range(3))
        lib().orc_cell_log_likelihoods(self.h, _p(alpha), _p(beta), _p(mask), _p(ll), _p(ell), _p(nl))
        return ll, ell, nl

    def locus_stats(self, alpha, beta, mask, new_excluded):
        L = self.loci_used
        alpha = np.ascontiguousarray(alpha, np.float64); beta = np.ascontiguousarray(beta, np.float64)
        mask = np.ascontiguousarray(mask, np.uint8); new_excluded = np.ascontiguousarray(new_excluded, np.uint8)
        cm, cj = np.empty(L, np.float64), np.empty(L, np.float64)
        ints = [np.empty(L, np.uint64) for _ in range(6)]
        lib().orc_locus_stats(self.h, _p(alpha), _p(beta), _p(mask), _p(new_excluded), _p(cm), _p(cj),
                              *[_p(a) for a in ints])
        keys = ["cells_min", "cells_maj", "alt_min", "ref_min", "alt_maj", "ref_maj"]
        return dict(contrib_min=cm, contrib_maj=cj, **dict(zip(keys, ints)))

    def posteriors(self):
        n = self.total_cells
        p, dp, lmaj, lmin = (np.empty(n, np.float64) for _ in range(4))
        lib().orc_posteriors(self.h, _p(p), _p(dp), _p(lmaj), _p(lmin))
        return dict(posterior=p, doublet_posterior=dp, ll_majority=lmaj, ll_minority=lmin)

    def assignments(self, posterior, doublet_posterior, posterior_threshold=0.999, min_loci_used=30):
        n = self.total_cells
        pa, aa, q = np.empty(n, np.uint8), np.empty(n, np.uint8), np.empty(n, np.uint64)
        posterior = np.ascontiguousarray(posterior, np.float64)
        doublet_posterior = np.ascontiguousarray(doublet_posterior, np.float64)
        lib().orc_assignments(self.h, _p(posterior), _p(doublet_posterior), posterior_threshold,
                              min_loci_used, _p(pa), _p(aa), _p(q))
        return pa, aa, q

    def run(self, iqr_multiple=5.0, max_iter=1000):
        """cellector() outer loop, main.rs:42-46."""
        summaries = []
        for _ in range(max_iter):
            s = self.em_iteration(iqr_multiple)
            summaries.append(s)
            if not s.any_change:
                break
        return summaries


def expected_log_pmf(total, alpha, beta):
    e, v = C.c_double(), C.c_double()
    lib().orc_expected_log_beta_binomial_pmf(total, alpha, beta, C.byref(e), C.byref(v))
    return e.value, v.value


def median(x):
    x = np.ascontiguousarray(x, np.float64)
    return lib().orc_median(_p(x), len(x))


def quantile(x, tau):
    x = np.ascontiguousarray(x, np.float64)
    return lib().orc_quantile(_p(x), len(x), tau)


def final_tallies_coo(total_loci, locus0, cell0, alt, ref, excluded):
    arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (locus0, cell0, alt, ref)]
    excluded = np.ascontiguousarray(excluded, np.uint8)
    outs = [np.empty(total_loci, np.uint64) for _ in range(4)]
    lib().orc_final_tallies_coo(total_loci, len(arrs[0]), *[_p(a) for a in arrs], _p(excluded),
                                *[_p(o) for o in outs])
    return dict(zip(["alt_min", "ref_min", "alt_maj", "ref_maj"], outs))


def vcf_genotype(alt_min, ref_min, alt_maj, ref_maj):
    gmaj, gmin = C.c_uint8(), C.c_uint8()
    pmaj, pmin = C.c_double(), C.c_double()
    lib().orc_vcf_genotype(int(alt_min), int(ref_min), int(alt_maj), int(ref_maj), C.byref(gmaj),
                           C.byref(pmaj), C.byref(gmin), C.byref(pmin))
    return gmaj.value, pmaj.value, gmin.value, pmin.value
