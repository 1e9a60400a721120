"""Multi-GPU driver: one process per GPU, cells sharded contiguously, torch.distributed for the exchanges.

The scoring loop needs three exchanges (include/cellector_ffi.h), all f64 sum all-reduces:
  PASS1 (once, at load)  — per-locus pass-1 counts and allele totals, so every rank applies the same locus filter;
  NORM  (per iteration)  — every rank contributes its cells' normalised log-likelihoods (other slices zero), so that
                           every rank runs the same exact order-statistic selection and gets the same threshold;
  LOCUS (per iteration)  — per-locus minority tallies / contribution sums and the change counters.
With backend "nccl" these run as RCCL all-reduces over xGMI on the device buffers the engine exposes; with "gloo"
(CPU tests) on host tensors.  The driver is written against a small *engine* interface so that the N > 1 host logic
can be exercised on CPU with a test-only engine; the product engine is HipShardEngine (libcellector_hip.so).
"""
import numpy as np

from . import ffi

XCHG_PASS1, XCHG_NORM, XCHG_LOCUS = ffi.XCHG_PASS1, ffi.XCHG_NORM, ffi.XCHG_LOCUS


def shard_range(total_cells, rank, world):
    """Contiguous, equal-count cell ranges (the last ranks may be one block shorter)."""
    per = (total_cells + world - 1) // world
    return min(total_cells, rank * per), min(total_cells, (rank + 1) * per)


class HipShardEngine:
    """Engine over libcellector_hip.so; exchange buffers are torch CUDA tensors bound into the ctx."""

    def __init__(self, device, stream=None, engine=2, compute_expected=True, keep_coo=False):
        import torch
        self.torch = torch
        self.device = torch.device("cuda", device)
        self.g = ffi.Cellector(device, stream=stream if stream is not None else torch.cuda.current_stream(self.device).cuda_stream)
        self.g.set_option("engine", engine)
        self.g.set_option("compute_expected", 1 if compute_expected else 0)
        self.g.set_option("keep_coo", 1 if keep_coo else 0)
        self.buf = {}

    def set_shard(self, cb, ce):
        self.g.set_shard(cb, ce)

    def _bind(self, which, n):
        t = self.torch.zeros(int(n), dtype=self.torch.float64, device=self.device)
        self.g.bind_exchange_buffer(which, t.data_ptr(), t.numel())
        self.buf[which] = t
        return t

    # ingest: every variant binds PASS1 first and returns the tensor to all-reduce
    def ingest_synthetic(self, total_loci, total_cells, density, **kw):
        self._bind(XCHG_PASS1, 5 * total_loci)
        self.g.ingest_synthetic(total_loci, total_cells, density, **kw)
        return self.buf[XCHG_PASS1]

    def ingest_coo(self, total_loci, total_cells, locus0, cell0, alt, ref):
        self._bind(XCHG_PASS1, 5 * total_loci)
        self.g.ingest_coo(total_loci, total_cells, locus0, cell0, alt, ref)
        return self.buf[XCHG_PASS1]

    def ingest_mtx(self, alt_path, ref_path, total_loci):
        self._bind(XCHG_PASS1, 5 * total_loci)
        self.g.ingest_mtx(alt_path, ref_path)
        return self.buf[XCHG_PASS1]

    def ingest_finish(self, min_alt, min_ref, norm_len=0):
        self.g.ingest_finish(min_alt, min_ref)
        d = self.g.dims()
        self._bind(XCHG_NORM, max(d.total_cells, norm_len))  # padded to world * cells-per-rank for the all-gather
        self._bind(XCHG_LOCUS, 5 * d.loci_used + 8)
        return d

    def em_begin(self):
        self.g.em_begin()
        return self.buf[XCHG_NORM]

    def em_threshold(self, iqr_multiple):
        self.g.em_threshold(iqr_multiple)
        return self.buf[XCHG_LOCUS]

    def em_finish(self):
        return self.g.em_finish()

    def __getattr__(self, name):  # outputs: cell_outputs, locus_outputs, excluded, posteriors, dims, ...
        return getattr(self.g, name)


class ShardedCellector:
    """cellector() (main.rs:36-50) over `world` shards; `dist` is torch.distributed (initialised) or None."""

    def __init__(self, engine, dist=None):
        self.e = engine
        self.dist = dist
        self.world = dist.get_world_size() if dist is not None else 1
        self.rank = dist.get_rank() if dist is not None else 0
        self.summaries = []

    def _allreduce(self, t):
        if self.dist is not None and self.world > 1:
            self.dist.all_reduce(t)  # SUM

    def _exchange_norm(self, t):
        """Every rank contributes its own cells' slice.  Shards are equal-sized contiguous ranges, so this is an
        in-place all-gather (recv = whole buffer, send = this rank's slice of it); half the bytes of the sum
        all-reduce it replaces.  Falls back to the all-reduce (other slices are zero) if the buffer is unpadded."""
        if self.dist is None or self.world == 1:
            return
        per = (self.total_cells + self.world - 1) // self.world
        if t.numel() >= self.world * per:
            self.dist.all_gather_into_tensor(t[: self.world * per], t[self.rank * per:(self.rank + 1) * per])
        else:
            self.dist.all_reduce(t)

    def load(self, ingest, total_cells, min_alt=4, min_ref=4):
        """ingest(engine) -> PASS1 tensor; e.g. lambda e: e.ingest_synthetic(L, N, d, seed=4)."""
        cb, ce = shard_range(total_cells, self.rank, self.world)
        self.total_cells = total_cells
        self.e.set_shard(cb, ce)
        self._allreduce(ingest(self.e))
        per = (total_cells + self.world - 1) // self.world
        self.dims = self.e.ingest_finish(min_alt, min_ref, self.world * per)
        return self.dims

    def em_iteration(self, iqr_multiple=5.0):
        self._exchange_norm(self.e.em_begin())
        self._allreduce(self.e.em_threshold(iqr_multiple))
        s = self.e.em_finish()
        self.summaries.append(s)
        return s

    def run(self, iqr_multiple=5.0, max_iter=1000):
        for _ in range(max_iter):
            if not self.em_iteration(iqr_multiple).any_change:
                break
        return self.summaries

    def gather_cells(self, local, dtype=np.float64):
        """Concatenate a per-local-cell array over ranks in global cell order (host side, all ranks get it)."""
        local = np.ascontiguousarray(local, dtype=dtype)
        if self.dist is None or self.world == 1:
            return local
        out = [None] * self.world
        self.dist.all_gather_object(out, local)
        return np.concatenate(out)
