#!/usr/bin/env python3
"""Turns the output of tools/gpu_profile.sh (gpurun_out/<tag>/) into the committed files under profiles/:
<tag>_cfg4_kernel_stats.csv, <tag>_cfg4_pmc.csv, <tag>_bench_cfg4.json, <tag>_bench_cfg3.json, and the traffic.json entry
that bench.py reports as roofline.traffic.  usage: python tools/make_profiles.py <tag> [engine]"""
import json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
engine = int(sys.argv[2]) if len(sys.argv) > 2 else 2
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, f"{tag}_cfg4_kernel_stats.csv"))
for w in ("cfg4", "cfg3"):
    line = [l for l in open(os.path.join(src, f"bench_{w}.json")) if l.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(dst, f"{tag}_bench_{w}.json"), "w"), indent=1)
fetch = {r["kernel"]: r["mean"] for r in json.load(open(os.path.join(src, "pmc_fetch_summary.json"))) if r["counter"] == "FETCH_SIZE"}
write = {r["kernel"]: r["mean"] for r in json.load(open(os.path.join(src, "pmc_write_summary.json"))) if r["counter"] == "WRITE_SIZE"}
rows = []
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    rows.append((k, f, w, 2 * f * 1024 + w * 1024))
rows.sort(key=lambda r: -r[3])
with open(os.path.join(dst, f"{tag}_cfg4_pmc.csv"), "w") as o:
    o.write("# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), same command (python3 bench.py --no-cpu-baseline --steps 3 --warmup 1); per-launch means in KiB as reported.\n")
    o.write("# gfx950 correction (MI355X_MICROARCH.md, HBM): hbm_bytes = 2*FETCH*1024 + WRITE*1024 (FETCH_SIZE counts 1/2 of streamed read bytes; calibrated on k_fill in r1_v1_cfg4_pmc.csv)\n")
    o.write("kernel,FETCH_SIZE_KiB,WRITE_SIZE_KiB,hbm_bytes_corrected\n")
    for k, f, w, b in rows:
        o.write('"%s",%.1f,%.1f,%d\n' % (k.split("(")[0], f, w, b))
tr_path = os.path.join(dst, "traffic.json")
tr = json.load(open(tr_path)) if os.path.exists(tr_path) else {}
dom = "k_tile_ll" if engine == 2 else "k_cell_ll"
cand = [r for r in rows if dom in r[0] and r[3] > 1e8]
if cand:
    tr[f"cfg4:n1:engine{engine}"] = {"kernel": dom, "hbm_bytes_per_launch": cand[0][3], "source": f"profiles/{tag}_cfg4_pmc.csv"}
    json.dump(tr, open(tr_path, "w"), indent=1)
if cand and engine == 2:  # the bench line was printed before this build's counters existed: carry the new figure over
    bp = os.path.join(dst, f"{tag}_bench_cfg4.json")
    b = json.load(open(bp))
    b["roofline"]["traffic"] = cand[0][3]
    b["roofline"]["traffic_note"] = f"PMC passes of this same build (profiles/{tag}_cfg4_pmc.csv), filled in after the run"
    json.dump(b, open(bp, "w"), indent=1)
print("wrote profiles for", tag, "dominant kernel traffic:", cand[0][3] if cand else None)
