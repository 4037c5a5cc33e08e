#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (written by tools/profile.sh) into tracked files under profiles/:
   <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (all kernels of the bench command)
   <tag>_pmc.csv            per-kernel averages of every PMC counter collected
   <tag>_bench.json         the bench line of the same command
   traffic_<workload>.json  HBM bytes per launch of the dominant kernel (read by bench.py)
usage: tools/summarize_profile.py <tag> <workload> <kernel-substring>"""
import collections, csv, glob, json, os, shutil, sys

tag, workload, kname = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))
if stats:
    rows = list(csv.reader(open(stats[0])))
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        for r in rows:
            r[0] = r[0][:120]
            w.writerow(r)
if os.path.exists(os.path.join(src, "bench.json")):
    shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
agg = collections.defaultdict(list)
for f in glob.glob(os.path.join(src, "pmc_*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"][:100], r["Counter_Name"])].append(float(r["Counter_Value"]))
with open(os.path.join(dst, f"{tag}_pmc.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatches", "mean_per_dispatch"])
    for (k, c), v in sorted(agg.items()):
        if kname in k:
            w.writerow([k, c, len(v), sum(v) / len(v)])
fetch = [sum(v) / len(v) for (k, c), v in agg.items() if kname in k and c == "FETCH_SIZE"]
write = [sum(v) / len(v) for (k, c), v in agg.items() if kname in k and c == "WRITE_SIZE"]
if fetch and write:
    # FETCH_SIZE / WRITE_SIZE are in KiB. On gfx950 FETCH_SIZE counts 128-B requests at 64 B for wide (16 B/lane)
    # coalesced streams (MI355X_MICROARCH.md §HBM); the factor for THIS kernel's access pattern comes from the
    # calibration run recorded in DESIGN.md and is passed via MOF_FETCH_FACTOR.
    factor = float(os.environ.get("MOF_FETCH_FACTOR", "1.0"))
    out = {"workload": workload, "kernel": kname, "fetch_size_kib": fetch[0], "write_size_kib": write[0],
           "fetch_factor": factor, "hbm_bytes_per_launch": fetch[0] * 1024 * factor + write[0] * 1024,
           "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, profiles/{tag}_pmc.csv"}
    json.dump(out, open(os.path.join(dst, f"traffic_{workload}.json"), "w"), indent=1)
    print(out)
print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read()[:600] if stats else "no stats")
