#!/usr/bin/env python3
"""Timeline of the last EM iterations out of a rocprofv3 --kernel-trace csv: start offset, duration and the gap to the
previous kernel's end, per stream.  usage: timeline.py <kernel_trace.csv> [n_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = t0
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:60]
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - prev_end) / 1e3:8.1f}  q{r.get('Queue_Id', '?'):>3}  {name}")
    prev_end = max(prev_end, e)
