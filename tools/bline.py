#!/usr/bin/env python3
"""One-line summary of a bench.py JSON line.  usage: bline.py <file> [label]"""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
r = d["roofline"]
print("%-28s ms %.3f  evals/s %.3g  kernels %s  tile_alg_frac %.3f" % (
    sys.argv[2] if len(sys.argv) > 2 else "", d["ms_per_step"], d["value"],
    {k: round(v, 3) for k, v in d["kernels_ms"].items()}, r["frac"]))
