#!/usr/bin/env python3
"""Prints the kernels of a rocprofv3 --kernel-trace --stats run, longest total first.  usage: kstats.py <dir> [n]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%-72s calls=%5s avg_us=%10.1f tot_ms=%9.1f" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                        float(r["TotalDurationNs"]) / 1e6))
