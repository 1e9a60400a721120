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
# SQ counters of the tile kernel alone (tools/gpu_pmc.sh <tag>_sq "SQ_..." --opt side_lds=70000 > gpurun_out/<tag>_sq.log)
sq = None
sq_log = os.path.join(ROOT, "gpurun_out", f"{tag}_sq.log")
if engine == 2 and os.path.exists(sq_log):
    vals, on = {}, False
    for ln in open(sq_log):
        if not ln.startswith(" "):
            on = ln.startswith("void k_tile_ll<true, 4>")
        elif on and len(ln.split()) >= 2:
            vals[ln.split()[0]] = float(ln.split()[1])
    if {"SQ_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"} <= set(vals):
        per_cu_cycles = vals["SQ_BUSY_CYCLES"] / 32.0  # (counted per shader engine x XCD: 32 of them)
        lds_per_cu = vals["SQ_LDS_IDX_ACTIVE"] / 256.0
        sq = {"source": f"profiles/{tag}_cfg4_sq_tile.csv", "kernel_cycles_per_cu": per_cu_cycles,
              "lds_busy_frac": lds_per_cu / per_cu_cycles,
              "bank_conflict_frac_of_lds_cycles": vals["SQ_LDS_BANK_CONFLICT"] / vals["SQ_LDS_IDX_ACTIVE"],
              "wave_wait_frac": vals["SQ_WAIT_INST_ANY"] / vals["SQ_WAVE_CYCLES"]}
        with open(os.path.join(dst, f"{tag}_cfg4_sq_tile.csv"), "w") as o:
            o.write("# rocprofv3 --pmc (one pass), python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 --opt side_lds=70000 (side kernels kept off "
                    "the tile kernel's CUs: the tile kernel alone); per-launch means, summed over the chip (256 CUs)\n")
            o.write("# k_tile_ll<true,4>, cfg4\ncounter,value\n")
            for k in sorted(vals):
                o.write("%s,%.4g\n" % (k, vals[k]))
            o.write("# derived: LDS array active %.0f %% of the kernel's cycles, %.0f %% of those cycles bank conflicts; waves waiting %.0f %% of their cycles\n"
                    % (100 * sq["lds_busy_frac"], 100 * sq["bank_conflict_frac_of_lds_cycles"], 100 * sq["wave_wait_frac"]))
if cand:
    ent = {"kernel": dom, "hbm_bytes_per_launch": cand[0][3], "source": f"profiles/{tag}_cfg4_pmc.csv"}
    old = tr.get(f"cfg4:n1:engine{engine}", {})
    if sq or "sq" in old:
        ent["sq"] = sq or old["sq"]
    tr[f"cfg4:n1:engine{engine}"] = ent
    json.dump(tr, open(tr_path, "w"), indent=1)
if cand and engine == 2:  # the bench line was printed before this build's counters existed: carry the new figure over
    bp = os.path.join(dst, f"{tag}_bench_cfg4.json")
    b = json.load(open(bp))
    b["roofline"]["traffic"] = cand[0][3]
    if sq:
        b["roofline"]["lds"]["sq_counters"] = sq
    b["roofline"]["traffic_note"] = f"PMC passes of this same build (profiles/{tag}_cfg4_pmc.csv), filled in after the run"
    json.dump(b, open(bp, "w"), indent=1)
print("wrote profiles for", tag, "dominant kernel traffic:", cand[0][3] if cand else None)
