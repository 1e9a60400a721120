"""ctypes binding of libcellector_hip.so (include/cellector_ffi.h).

Host-side plumbing only: every computation happens in the HIP library.  There is no CPU fallback — if the
library is missing or no GPU is present, construction fails loudly.
"""
import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CELLECTOR_HIP_LIB") or os.path.join(_HERE, "libcellector_hip.so")

XCHG_PASS1, XCHG_NORM, XCHG_LOCUS = 0, 1, 2
K_CELL_LL, K_LOCUS_STATS, K_SELECT, K_POSTERIOR, K_TILE_LL = 0, 1, 2, 3, 4
STATUS_NAMES = {0: "OK", 1: "EINVAL", 2: "EIO", 3: "EPARSE", 4: "ENOMEM", 5: "EDEVICE", 6: "ECOMM"}

# every entry point include/cellector_ffi.h declares: name -> (restype, argtypes)
_vp, _u64, _d, _cp, _i = C.c_void_p, C.c_uint64, C.c_double, C.c_char_p, C.c_int
SIGNATURES = {
    "cellector_create": (_i, [C.POINTER(_vp), _i]),
    "cellector_destroy": (None, [_vp]),
    "cellector_device_count": (_i, [C.POINTER(_i)]),
    "cellector_create_multi": (_i, [C.POINTER(_vp), C.POINTER(_i), _i]),
    "cellector_comm_unique_id": (_i, [_vp]),
    "cellector_comm_init_rank": (_i, [_vp, _vp, _i, _i]),
    "cellector_last_error": (_cp, [_vp]),
    "cellector_version": (_cp, []),
    "cellector_set_stream": (_i, [_vp, _vp]),
    "cellector_set_option": (_i, [_vp, _cp, C.c_int64]),
    "cellector_set_shard": (_i, [_vp, _u64, _u64]),
    "cellector_set_partition": (_i, [_vp, _vp, _i]),
    "cellector_partition": (_i, [_vp, _vp, C.POINTER(_i)]),
    "cellector_ingest_mtx": (_i, [_vp, _cp, _cp]),
    "cellector_ingest_coo": (_i, [_vp, _u64, _u64, _u64, _vp, _vp, _vp, _vp]),
    "cellector_ingest_synthetic": (_i, [_vp, _u64, _u64, _d, _u64, _d, _d]),
    "cellector_write_staged_mtx": (_i, [_vp, C.c_char_p, C.c_char_p]),
    "cellector_ingest_finish": (_i, [_vp, _u64, _u64]),
    "cellector_load_mtx": (_i, [_vp, _cp, _cp, _u64, _u64]),
    "cellector_load_coo": (_i, [_vp, _u64, _u64, _u64, _vp, _vp, _vp, _vp, _u64, _u64]),
    "cellector_dims": (_i, [_vp, _vp]),
    "cellector_locus_ids": (_i, [_vp, _vp]),
    "cellector_locus_counts": (_i, [_vp, _vp]),
    "cellector_entries_per_cell": (_i, [_vp, _vp]),
    "cellector_csr_rows": (_i, [_vp, _u64, _u64, _vp, _vp, _u64]),
    "cellector_exchange_buffer": (_i, [_vp, _i, C.POINTER(_vp), C.POINTER(_u64)]),
    "cellector_bind_exchange_buffer": (_i, [_vp, _i, _vp, _u64]),
    "cellector_em_begin": (_i, [_vp]),
    "cellector_em_threshold": (_i, [_vp, _d]),
    "cellector_em_finish": (_i, [_vp, _vp]),
    "cellector_em_iteration": (_i, [_vp, _d, _vp]),
    "cellector_iter_cell_outputs": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "cellector_iter_locus_outputs": (_i, [_vp] + [_vp] * 8),
    "cellector_loci_mask": (_i, [_vp, _vp]),
    "cellector_excluded": (_i, [_vp, _vp]),
    "cellector_alpha_betas": (_i, [_vp, _vp, _vp]),
    "cellector_cell_log_likelihoods": (_i, [_vp] + [_vp] * 6),
    "cellector_posteriors": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "cellector_final_allele_tallies": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "cellector_engine_info": (_i, [_vp, _vp]),
    "cellector_order_statistics": (_i, [_vp, _vp, _u64, _d, _vp]),
    "cellector_kernel_time": (_i, [_vp, _i, C.POINTER(_d), C.POINTER(_u64)]),
    "cellector_reset_timing": (_i, [_vp]),
}


class Dims(C.Structure):
    _fields_ = [("total_cells", _u64), ("total_loci", _u64), ("loci_used", _u64), ("cell_begin", _u64),
                ("cell_end", _u64), ("nnz_used", _u64)]


class EngineInfo(C.Structure):
    _fields_ = [("engine", _u64), ("nnz_regular", _u64), ("nnz_overflow", _u64), ("tile_bytes", _u64),
                ("cell_blocks", _u64), ("locus_chunks", _u64), ("chunk_groups", _u64), ("tile_lookups", _u64)]


class IterSummary(C.Structure):
    _fields_ = [("any_change", C.c_int32), ("n_new_excluded", _u64), ("n_rescued", _u64), ("n_excluded", _u64),
                ("n_loci_filtered", _u64), ("median", _d), ("iqr", _d), ("threshold", _d), ("n_near_threshold", _u64)]


class CellectorError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


_lib = None


def load_library(path=LIB_PATH):
    """dlopen the HIP library and attach signatures (no GPU needed for this step)."""
    global _lib
    if _lib is not None and path == LIB_PATH:
        return _lib
    # PyTorch-ROCm ships its own copy of the HIP runtime.  A process that uses both this library and torch.cuda must load
    # torch's copy FIRST (this library then binds to it by soname); the other order leaves two runtimes in the process and
    # torch.cuda fails to initialise ("No HIP GPUs are available").  CELLECTOR_NO_TORCH=1 skips this (torch-free hosts).
    if "torch" not in sys.modules and not os.environ.get("CELLECTOR_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(cellector_amd has no CPU fallback)")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if path == LIB_PATH:
        _lib = lib
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Cellector:
    """One shard of a (locus x cell) matrix on one GPU; mirrors the reference's main() flow."""

    def __init__(self, device=0, stream=None, devices=None):
        """devices=[d0, d1, ...]: ONE ctx over several GPUs (cellector_create_multi; a device listed twice = logical
        shards on that GPU); every method then works on the whole matrix, arrays in global cell order."""
        self._lib = load_library()
        h = C.c_void_p()
        if devices is not None and len(devices) > 1:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            st = self._lib.cellector_create_multi(C.byref(h), ids, len(devices))
        else:
            st = self._lib.cellector_create(C.byref(h), int(devices[0] if devices else device))
        if st != 0:
            raise CellectorError(st, "cellector_create failed (no MI355X visible? the HIP path is mandatory)")
        self.h = h
        if stream is not None:
            self._ck(self._lib.cellector_set_stream(self.h, C.c_void_p(stream)))

    def _ck(self, st):
        if st != 0:
            raise CellectorError(st, self._lib.cellector_last_error(self.h).decode(errors="replace"))

    def close(self):
        if getattr(self, "h", None):
            self._lib.cellector_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- configuration
    def set_option(self, key, value):
        self._ck(self._lib.cellector_set_option(self.h, key.encode(), int(value)))

    def set_stream(self, stream):
        """hipStream_t as an integer (e.g. torch.cuda.current_stream().cuda_stream); 0 / None = the null stream"""
        self._ck(self._lib.cellector_set_stream(self.h, C.c_void_p(stream or None)))

    def comm_init_rank(self, unique_id, n_ranks, rank):
        """attach an RCCL communicator (one process per GPU); unique_id = the 128 bytes of comm_unique_id() on rank 0"""
        buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
        self._ck(self._lib.cellector_comm_init_rank(self.h, buf, int(n_ranks), int(rank)))

    def set_shard(self, cell_begin, cell_end):
        self._ck(self._lib.cellector_set_shard(self.h, int(cell_begin), int(cell_end)))

    def set_partition(self, bounds):
        """cell ranges of the ranks of a ctx with a communicator: n_ranks + 1 boundaries (None: back to the default)"""
        if bounds is None:
            self._ck(self._lib.cellector_set_partition(self.h, None, 0))
            return
        b = np.ascontiguousarray(bounds, dtype=np.uint64)
        self._ck(self._lib.cellector_set_partition(self.h, _p(b), len(b)))

    def partition(self):
        """the cell ranges in use: array of n_ranks + 1 boundaries"""
        n = C.c_int(0)
        self._ck(self._lib.cellector_partition(self.h, None, C.byref(n)))
        out = np.zeros(n.value + 1, np.uint64)
        self._ck(self._lib.cellector_partition(self.h, _p(out), C.byref(n)))
        return out

    # ---- ingest
    def ingest_mtx(self, alt_path, ref_path):
        self._ck(self._lib.cellector_ingest_mtx(self.h, str(alt_path).encode(), str(ref_path).encode()))

    def ingest_coo(self, total_loci, total_cells, locus0, cell0, alt, ref):
        arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (locus0, cell0, alt, ref)]
        self._ck(self._lib.cellector_ingest_coo(self.h, total_loci, total_cells, len(arrs[0]), *[_p(a) for a in arrs]))

    def ingest_synthetic(self, total_loci, total_cells, density, seed=4, minority_fraction=0.05, doublet_fraction=0.0):
        self._ck(self._lib.cellector_ingest_synthetic(self.h, total_loci, total_cells, float(density), int(seed),
                                                      float(minority_fraction), float(doublet_fraction)))

    def ingest_finish(self, min_alt=4, min_ref=4):
        self._ck(self._lib.cellector_ingest_finish(self.h, int(min_alt), int(min_ref)))

    def write_staged_mtx(self, alt_path, ref_path):
        """Benchmark utility: the staged matrix as a vartrix-style text pair (needs option keep_coo=1)."""
        self._ck(self._lib.cellector_write_staged_mtx(self.h, str(alt_path).encode(), str(ref_path).encode()))

    def load_mtx(self, alt_path, ref_path, min_alt=4, min_ref=4):
        self._ck(self._lib.cellector_load_mtx(self.h, str(alt_path).encode(), str(ref_path).encode(), min_alt, min_ref))

    def load_coo(self, total_loci, total_cells, locus0, cell0, alt, ref, min_alt=4, min_ref=4):
        self.ingest_coo(total_loci, total_cells, locus0, cell0, alt, ref)
        self.ingest_finish(min_alt, min_ref)

    def load_synthetic(self, total_loci, total_cells, density, seed=4, minority_fraction=0.05, doublet_fraction=0.0,
                       min_alt=4, min_ref=4):
        self.ingest_synthetic(total_loci, total_cells, density, seed, minority_fraction, doublet_fraction)
        self.ingest_finish(min_alt, min_ref)

    # ---- accessors
    def dims(self):
        d = Dims()
        self._ck(self._lib.cellector_dims(self.h, C.byref(d)))
        return d

    @property
    def n_local(self):
        d = self.dims()
        return d.cell_end - d.cell_begin

    def locus_ids(self):
        out = np.empty(self.dims().loci_used, np.uint64)
        self._ck(self._lib.cellector_locus_ids(self.h, _p(out)))
        return out

    def locus_counts(self):
        out = np.empty((self.dims().loci_used, 2), np.float64)
        self._ck(self._lib.cellector_locus_counts(self.h, _p(out)))
        return out

    def entries_per_cell(self):
        out = np.empty(self.n_local, np.uint32)
        self._ck(self._lib.cellector_entries_per_cell(self.h, _p(out)))
        return out

    def csr_rows(self, row_begin, row_end):
        rp = np.empty(row_end - row_begin + 1, np.uint64)
        self._ck(self._lib.cellector_csr_rows(self.h, row_begin, row_end, _p(rp), None, 0))
        ent = np.empty(int(rp[-1]), np.uint64)
        self._ck(self._lib.cellector_csr_rows(self.h, row_begin, row_end, _p(rp), _p(ent), len(ent)))
        return rp, ent

    def exchange_buffer(self, which):
        ptr, n = C.c_void_p(), C.c_uint64()
        self._ck(self._lib.cellector_exchange_buffer(self.h, which, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def bind_exchange_buffer(self, which, dev_ptr, n_f64):
        self._ck(self._lib.cellector_bind_exchange_buffer(self.h, which, C.c_void_p(dev_ptr), int(n_f64)))

    # ---- EM loop
    def em_begin(self):
        self._ck(self._lib.cellector_em_begin(self.h))

    def em_threshold(self, iqr_multiple=5.0):
        self._ck(self._lib.cellector_em_threshold(self.h, float(iqr_multiple)))

    def em_finish(self):
        s = IterSummary()
        self._ck(self._lib.cellector_em_finish(self.h, C.byref(s)))
        return s

    def em_iteration(self, iqr_multiple=5.0):
        s = IterSummary()
        self._ck(self._lib.cellector_em_iteration(self.h, float(iqr_multiple), C.byref(s)))
        return s

    def run(self, iqr_multiple=5.0, max_iter=1000):
        """cellector() outer loop (main.rs:42-46)."""
        out = []
        for _ in range(max_iter):
            s = self.em_iteration(iqr_multiple)
            out.append(s)
            if not s.any_change:
                break
        return out

    def cell_outputs(self):
        n = self.n_local
        ll, ell, nl, norm = (np.empty(n, np.float64) for _ in range(4))
        self._ck(self._lib.cellector_iter_cell_outputs(self.h, _p(ll), _p(ell), _p(nl), _p(norm)))
        return dict(ll=ll, expected_ll=ell, loci_used=nl, normalized=norm)

    def locus_outputs(self):
        L = self.dims().loci_used
        cm, cj = np.empty(L, np.float64), np.empty(L, np.float64)
        ints = [np.empty(L, np.uint64) for _ in range(6)]
        self._ck(self._lib.cellector_iter_locus_outputs(self.h, _p(cm), _p(cj), *[_p(a) for a in ints]))
        keys = ["cells_min", "cells_maj", "alt_min", "ref_min", "alt_maj", "ref_maj"]
        return dict(contrib_min=cm, contrib_maj=cj, **dict(zip(keys, ints)))

    def loci_mask(self):
        out = np.empty(self.dims().loci_used, np.uint8)
        self._ck(self._lib.cellector_loci_mask(self.h, _p(out)))
        return out

    def excluded(self):
        out = np.empty(self.n_local, np.uint8)
        self._ck(self._lib.cellector_excluded(self.h, _p(out)))
        return out

    def alpha_betas(self):
        L = self.dims().loci_used
        a, b = np.empty(L, np.float64), np.empty(L, np.float64)
        self._ck(self._lib.cellector_alpha_betas(self.h, _p(a), _p(b)))
        return a, b

    def cell_log_likelihoods(self, alpha, beta, mask=None):
        n = self.n_local
        alpha = np.ascontiguousarray(alpha, np.float64)
        beta = np.ascontiguousarray(beta, np.float64)
        mask = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        ll, ell, nl = (np.empty(n, np.float64) for _ in range(3))
        self._ck(self._lib.cellector_cell_log_likelihoods(self.h, _p(alpha), _p(beta), _p(mask), _p(ll), _p(ell), _p(nl)))
        return ll, ell, nl

    def posteriors(self, fetch=True):
        if not fetch:  # (the phase on the device only: benchmarks)
            self._ck(self._lib.cellector_posteriors(self.h, None, None, None, None))
            return None
        n = self.n_local
        p, dp, lmaj, lmin = (np.empty(n, np.float64) for _ in range(4))
        self._ck(self._lib.cellector_posteriors(self.h, _p(p), _p(dp), _p(lmaj), _p(lmin)))
        return dict(posterior=p, doublet_posterior=dp, ll_majority=lmaj, ll_minority=lmin)

    def final_allele_tallies(self):
        tl = self.dims().total_loci
        outs = [np.empty(tl, np.uint64) for _ in range(4)]
        self._ck(self._lib.cellector_final_allele_tallies(self.h, *[_p(o) for o in outs]))
        return dict(zip(["alt_min", "ref_min", "alt_maj", "ref_maj"], outs))

    def engine_info(self):
        e = EngineInfo()
        self._ck(self._lib.cellector_engine_info(self.h, C.byref(e)))
        return e

    def order_statistics(self, keys, iqr_multiple=5.0):
        """(median, iqr, threshold) of the keys by the device select (a multi-device ctx: the sharded select)."""
        keys = np.ascontiguousarray(keys, dtype=np.float64)
        out = np.zeros(3, dtype=np.float64)
        self._ck(self._lib.cellector_order_statistics(self.h, keys.ctypes.data, len(keys), float(iqr_multiple), out.ctypes.data))
        return tuple(out.tolist())

    # ---- timing
    def kernel_time(self, which):
        ms, n = C.c_double(), C.c_uint64()
        self._ck(self._lib.cellector_kernel_time(self.h, which, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def reset_timing(self):
        self._ck(self._lib.cellector_reset_timing(self.h))


def device_count():
    n = C.c_int(0)
    load_library().cellector_device_count(C.byref(n))
    return n.value


def comm_unique_id():
    """128 opaque bytes (ncclUniqueId) for Cellector.comm_init_rank; made on rank 0, broadcast by the host"""
    buf = (C.c_char * 128)()
    st = load_library().cellector_comm_unique_id(buf)
    if st != 0:
        raise CellectorError(st, "cellector_comm_unique_id failed (RCCL not loadable?)")
    return bytes(buf)


def assignments(posterior, doublet_posterior, entries_per_cell, excluded, posterior_threshold=0.999,
                min_loci_used=30):
    """The labelling rule of output_final_assignments (main.rs:141-171) on host arrays.

    Returns (posterior_assignment, anomaly_assignment, qual): codes 0 -> "0" (minority), 1 -> "1" (majority),
    2 -> "doublet", 3 -> "unassigned"; anomaly 0 if the cell is in the final exclusion set else 1.
    """
    p = np.asarray(posterior, np.float64)
    pa = np.full(p.shape, 3, np.uint8)
    pa[(1.0 - p) > posterior_threshold] = 1
    pa[p > posterior_threshold] = 0
    pa[np.asarray(doublet_posterior) > 0.5] = 2
    pa[np.asarray(entries_per_cell) < min_loci_used] = 3
    aa = np.where(np.asarray(excluded) != 0, 0, 1).astype(np.uint8)
    post = np.fmax(p, 1.0 - p)
    with np.errstate(divide="ignore", invalid="ignore"):
        q = np.fmin(-10.0 * np.log10(1.0 - post), 255.0)
    q = np.where(np.isnan(q) | (q < 0), 0.0, q)
    return pa, aa, q.astype(np.uint64)
