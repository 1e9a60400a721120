"""Host twin of the device synthetic-matrix generator (csrc/kernels_synth.hip) and .mtx writers.

All-integer definition (see the kernel file's header), so the numpy result is bit-identical to the device
result.  Used to make small parity inputs and CLI demo files; big benchmark matrices are generated on the
GPU by cellector_ingest_synthetic.
"""
import gzip
import os

import numpy as np

GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
GEOM = np.array([0, 1288490188, 386547056, 115964116, 34789235, 10436770, 3131031, 939309, 281792, 84537, 25361,
                 7608, 2282, 684, 205, 61, 18, 5, 1], dtype=np.uint64)
AF16 = np.array([655, 32768, 64881], dtype=np.uint32)


def mix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z ^ (z >> np.uint64(30))
        z = z * _M1
        z = z ^ (z >> np.uint64(27))
        z = z * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def _seeds(seed):
    with np.errstate(over="ignore"):
        base = mix64(np.uint64(seed) * GOLD + np.uint64(0x5EED))
        return [mix64(base + np.uint64(k)) for k in range(1, 6)]


def geom_table(continue_pct=30):
    """thresholds T_k of n = 1 + #{k >= 1 : low32(h2) < T_k} (see csrc/kernels_synth.hip)"""
    if continue_pct == 30:
        return GEOM
    t, out = 1 << 32, [0]
    for _ in range(1, 64):
        t = t * int(continue_pct) // 100
        out.append(t)
    return np.array(out, dtype=np.uint64)


def _thr24(x):
    return np.uint64(int(x * 16777216.0 + 0.5))


def genotype_af16(s, loci):
    v = (mix64(s ^ loci.astype(np.uint64)) >> np.uint64(48)).astype(np.uint32)
    return np.where(v < 32768, AF16[0], np.where(v < 52428, AF16[1], AF16[2])).astype(np.uint32)


def cell_classes(total_cells, seed=4, minority_fraction=0.05, doublet_fraction=0.0):
    """0 = majority, 1 = minority, 2 = doublet (ground truth of the synthetic mixture)."""
    _, s_min, s_dbl, _, _ = _seeds(seed)
    c = np.arange(total_cells, dtype=np.uint64)
    cls = np.zeros(total_cells, np.uint8)
    cls[(mix64(s_min ^ c) >> np.uint64(40)) < _thr24(minority_fraction)] = 1
    cls[(mix64(s_dbl ^ c) >> np.uint64(40)) < _thr24(doublet_fraction)] = 2
    return cls


def generate_coo(total_loci, total_cells, density, seed=4, minority_fraction=0.05, doublet_fraction=0.0,
                 cell_begin=0, cell_end=None, continue_pct=30):
    """COO triplets sorted by (locus, cell), 0-based GLOBAL indices: locus, cell, alt, ref (uint32 each)."""
    cell_end = total_cells if cell_end is None else cell_end
    s_lc, s_min, s_dbl, s_gmaj, s_gmin = _seeds(seed)
    cells = np.arange(cell_begin, cell_end, dtype=np.uint64)
    is_min = (mix64(s_min ^ cells) >> np.uint64(40)) < _thr24(minority_fraction)
    is_dbl = (mix64(s_dbl ^ cells) >> np.uint64(40)) < _thr24(doublet_fraction)
    loci = np.arange(total_loci, dtype=np.uint64)
    af_maj, af_min = genotype_af16(s_gmaj, loci), genotype_af16(s_gmin, loci)
    thr = _thr24(density)
    geom = geom_table(continue_pct)
    out_l, out_c, out_a, out_r = [], [], [], []
    with np.errstate(over="ignore"):
        for l in range(total_loci):
            h1 = mix64(s_lc ^ ((np.uint64(l) << np.uint64(32)) | cells))
            sel = np.nonzero((h1 >> np.uint64(40)) < thr)[0]
            if sel.size == 0:
                continue
            h1 = h1[sel]
            h2 = mix64(h1 + GOLD)
            u = h2 & np.uint64(0xFFFFFFFF)
            n = np.ones(sel.size, np.uint32)
            alive = np.ones(sel.size, bool)
            for k in range(1, len(geom)):
                alive &= u < geom[k]
                if not alive.any():
                    break
                n += alive
            af = np.where(is_min[sel], af_min[l], af_maj[l]).astype(np.uint32)
            af = np.where(is_dbl[sel], (int(af_maj[l]) + int(af_min[l])) >> 1, af).astype(np.uint32)
            h3 = mix64(h2 + GOLD)
            alt = np.zeros(sel.size, np.uint32)
            for i in range(int(n.max())):
                field = ((h3 >> np.uint64(16 * (i & 3))) & np.uint64(0xFFFF)).astype(np.uint32)
                alt += ((field < af) & (i < n)).astype(np.uint32)
                if (i & 3) == 3:
                    h3 = mix64(h3 + GOLD)
            out_l.append(np.full(sel.size, l, np.uint32))
            out_c.append(cells[sel].astype(np.uint32))
            out_a.append(alt)
            out_r.append(n - alt)
    if not out_l:
        z = np.zeros(0, np.uint32)
        return z, z.copy(), z.copy(), z.copy()
    return (np.concatenate(out_l), np.concatenate(out_c), np.concatenate(out_a), np.concatenate(out_r))


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def write_mtx_pair(directory, total_loci, total_cells, locus0, cell0, alt, ref, gz=False, header_nnz=None):
    """alt.mtx / ref.mtx in the layout vartrix writes: 3 header lines, 1-based
    `locus cell count`, identical entry order in both files."""
    os.makedirs(directory, exist_ok=True)
    ext = ".mtx.gz" if gz else ".mtx"
    paths = {}
    nnz = len(locus0) if header_nnz is None else header_nnz
    for name, vals in (("alt", alt), ("ref", ref)):
        path = os.path.join(directory, name + ext)
        body = np.stack([np.asarray(locus0, np.int64) + 1, np.asarray(cell0, np.int64) + 1,
                         np.asarray(vals, np.int64)], axis=1)
        with _open(path, "wt") as f:
            f.write("%%MatrixMarket matrix coordinate real general\n% written by sprs\n")
            f.write(f"{total_loci} {total_cells} {nnz}\n")
            np.savetxt(f, body, fmt="%d")
        paths[name] = path
    return paths["alt"], paths["ref"]


def bgzf_compress(data, block=0xff00, eof_block=True):
    """`data` as a BGZF file (what bgzip writes): gzip members of at most 64 KB of text, each with the 'BC' extra subfield that
    holds the member's size - 1, and the empty end-of-file member."""
    import struct
    import zlib
    out = bytearray()
    pieces = [data[i:i + block] for i in range(0, len(data), block)] + ([b""] if eof_block else [])
    for piece in pieces:
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        body = c.compress(piece) + c.flush()
        bsize = 12 + 6 + len(body) + 8 - 1
        out += b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
        out += body + struct.pack("<II", zlib.crc32(piece) & 0xffffffff, len(piece))
    return bytes(out)


def write_barcodes(path, total_cells):
    with _open(path, "wt") as f:
        for i in range(total_cells):
            f.write(f"CELL{i:07d}-1\n")


def write_ground_truth(path, classes):
    names = {0: "majority", 1: "minority", 2: "doublet"}
    with _open(path, "wt") as f:
        for i, c in enumerate(classes):
            f.write(f"CELL{i:07d}-1\t{names[int(c)]}\n")
