#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<round>_<workload>/ (tools/profile_all.sh) into tracked files under profiles/:
   <round>_<wl>_kernel_stats.csv  rocprofv3 --kernel-trace --stats rows of the library's kernels (+ the top others)
   <round>_<wl>_pmc.csv           per-kernel means of FETCH_SIZE / WRITE_SIZE (separate --pmc passes)
   <round>_<wl>_bench.json        the bench line of the same command
   traffic_<wl>.json              fabric bytes per bench step (what bench.py reports as roofline.traffic)
FETCH_SIZE correction (MI355X_MICROARCH.md, HBM section): on gfx950 the counter tallies 128-byte requests at 64 bytes,
i.e. reports HALF the bytes of wide contiguous streaming reads. Which kernels read that way is decided by calibration
against KNOWN byte counts, recorded in FACTORS below with the evidence; everything else is taken at face value.
usage: tools/summarize_round.py r02"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

# kernel-name substring -> (FETCH_SIZE factor, evidence)
FACTORS = {
    # r04: the long-range load path (DS = 4: every lane reads 64 contiguous bytes of each tapped row as four 16-byte loads, lanes 64
    # bytes apart -- the requests leave the L2 as a MIX of 64- and 128-byte ones, so the counter's factor is neither 1 nor 2).
    # Keys are matched in order: the specific instantiations before the generic kernel name.
    "pc_field_kernel<128, 4": (1.469, "callr workload (bench.py): 512x512 frames, pitch 512, one 128x128 quarter-resolution patch = the whole "
                                      "frame; cv::resize(1/4) taps rows 4r+1, 4r+2 only, whole 128-byte lines of them: known 2*256*512*1024 B = "
                                      "268.44 MB per launch, FETCH_SIZE 182.71 MB (profiles/r04_callr_pmc.csv)"),
    "pc_field_kernel_120<4": (1.632, "reflr itself: the tapped row pairs 4r+1, 4r+2 are 960 contiguous bytes every 1920: at least 2*240*480*1024 B "
                                     "= 235.93 MB per launch must be fetched (unaligned 480-byte rows: a few % more in whole sectors), FETCH_SIZE "
                                     "144.53 MB (profiles/r04_reflr_pmc.csv): factor >= 1.632, used as a lower bound"),
    "pc_field_kernel": (1.0, "cal workload: 512x512 frames tiled exactly by 64x64 patches, known 2*512*512*1024 B = 536.87 MB, FETCH_SIZE 536.97 MB"),
    "sr_rows_fwd_kernel": (2.0, "reads 2*480*480 B of u8 per pair = 118.0 MB per 256-pair launch; FETCH_SIZE 59.1 MB = 0.50x"),
    "sr_cols_kernel": (2.0, "reads Zt once: 480*480*8 B per pair = 471.9 MB per 256-pair launch; FETCH_SIZE 236.06 MB = 0.50x"),
    "sr_rows_inv_kernel": (2.0, "reads Dt once: 241*480*8 B per pair = 236.9 MB per 256-pair launch; FETCH_SIZE 118.59 MB = 0.50x"),
    # r03 (sequence pipeline, gpurun_out/prof_r03b_c5seq): known bytes per 512-frame pass
    "sr_rows_real_kernel": (2.0, "reads 480*480 B of u8 per frame = 118.0 MB per 512-frame launch; FETCH_SIZE 58.9 MB = 0.50x"),
    # r04, the planned large-patch pipeline on l480 (512 pairs of ONE 480 x 480 patch per launch; profiles/r04_l480_pmc.csv)
    "pcl_rows_kernel": (2.0, "reads 480*480 B of u8 per image = 235.9 MB per 1024-image launch; FETCH_SIZE 118.8 MB = 0.50x"),
    "pcl_cols_kernel": (2.0, "reads the Zh of both images once: 2 * 925,440 B per pair = 947.6 MB per 512-pair launch; FETCH_SIZE 474.1 MB = 0.50x"),
    "pcl_rows_inv_kernel": (2.0, "reads Dt once: 925,440 B per pair = 473.8 MB per 512-pair launch; FETCH_SIZE 237.1 MB = 0.50x"),
    "sr_cols_fused_kernel": (2.0, "r04, K56: every log-polar image comes from HBM once (TCC_HIT 96 %): 2*480*480 B per pair = 471.9 MB per 1024-pair "
                                  "launch; FETCH_SIZE 234.5 MB = 0.50x (profiles/r04_sr_fused_pmc.csv)"),
    "sr_cols_seq_kernel": (2.0, "reads 17 frames of Zh (925,440 B) per 16-pair run = 503.4 MB per 512-pair launch; FETCH_SIZE 251.5 MB = 0.50x"),
    # r05: the half-tile kernel reads every pixel of every patch exactly once with one unaligned dword per four pixels. At l160 (160-byte
    # patch rows) FETCH_SIZE reads BELOW the bytes that must be fetched -- 189.29 MB against 2*480*480*512 B = 235.93 MB per launch
    # (profiles/r05_l160_pmc.csv): factor 1.247 from its own lower bound, for THAT instantiation only; at ref (120-byte rows: 472.7 MB
    # against 471.9 MB) and p96 the counter matches the known bytes, at p60 (60-byte rows, neighbouring patches share every line) it
    # reads 1.54 x -- real re-fetches. The other instantiations are taken at face value.
    "pc_half_kernel<1, 160": (1.247, "l160: known 235.93 MB per launch (every pixel once), FETCH_SIZE 189.29 MB: factor 1.247 (lower bound from its own algorithmic bytes)"),
    "pc_seq_kernel": (1.0, "8-byte loads; 285.2 MB of patch pixels per launch x the same 1.40 row-gap overhead K1 shows on the c2 layout = 399 MB; FETCH_SIZE 398.7 MB"),
}


def factor(kernel):
    for k, (f, _) in FACTORS.items():
        if k in kernel:
            return f
    return 1.0


# the session record of the f32-limited patches (tests/tolerances.py, written by tests/conftest.py) and the round's fuzz summary
# (tools/fuzz_round.sh), when this round's GPU runs left them
for src, name in (("f32_limited.json", f"{tag}_f32_limited.json"), (f"{tag}_fuzz.txt", f"{tag}_fuzz.txt")):
    if os.path.exists(os.path.join(root, "gpurun_out", src)):
        shutil.copy(os.path.join(root, "gpurun_out", src), os.path.join(dst, name))

for d in sorted(glob.glob(os.path.join(root, "gpurun_out", f"prof_{tag}_*"))):
    wl = os.path.basename(d)[len(f"prof_{tag}_"):]
    if wl == "mfma":
        shutil.copy(os.path.join(d, "result.json"), os.path.join(dst, f"{tag}_mfma_rowdft_result.json"))
        for st in glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")):
            shutil.copy(st, os.path.join(dst, f"{tag}_mfma_rowdft_kernel_stats.csv"))
        continue
    bench = None
    if not os.path.exists(os.path.join(d, "bench.json")):
        continue  # (a counter-only directory of tools/pmc.sh, e.g. prof_<round>_c4_sq: condensed by tools/pmc_table.py)
    if os.path.exists(os.path.join(d, "bench.json")):
        bench = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
        json.dump(bench, open(os.path.join(dst, f"{tag}_{wl}_bench.json"), "w"), indent=1)
    calls = {}
    # gpurun MERGES result files: a re-profiled workload leaves the older run's files beside the new ones -- newest only
    for st in sorted(glob.glob(os.path.join(d, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)[-1:]:
        rows = list(csv.reader(open(st)))
        with open(os.path.join(dst, f"{tag}_{wl}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(rows[0])
            others = 0
            for r in rows[1:]:
                if "mof::" in r[0]:
                    calls[r[0]] = int(r[1])
                    w.writerow([r[0][:120]] + r[1:])
                elif others < 4:  # the largest foreign kernels (torch's, copies) for context
                    w.writerow([r[0][:120]] + r[1:])
                    others += 1
    agg = collections.defaultdict(list)
    newest = [sorted(glob.glob(os.path.join(pd, "*", "*_counter_collection.csv")), key=os.path.getmtime)[-1:]
              for pd in glob.glob(os.path.join(d, "pmc_*")) if os.path.isdir(pd)]
    for f in [x for fs in newest for x in fs]:
        for r in csv.DictReader(open(f)):
            if "mof::" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    steps = (bench["steps"] + bench["warmup"]) if bench else 23
    per_step, per_step_raw, detail = 0.0, 0.0, []
    with open(os.path.join(dst, f"{tag}_{wl}_pmc.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean_KiB_per_dispatch", "fetch_factor", "bytes_per_step"])
        for (k, c), v in sorted(agg.items()):
            fac = factor(k) if c == "FETCH_SIZE" else 1.0
            launches_per_step = len(v) / steps
            b = sum(v) / len(v) * 1024 * fac * launches_per_step
            per_step += b
            per_step_raw += sum(v) / len(v) * 1024 * launches_per_step
            w.writerow([k[:100], c, len(v), round(sum(v) / len(v), 1), fac, round(b)])
            detail.append({"kernel": k[:80], "counter": c, "launches_per_step": launches_per_step,
                           "mean_kib": sum(v) / len(v), "factor": fac})
    if agg:
        out = {"workload": wl, "hbm_bytes_per_launch": per_step, "raw_counter_bytes_per_step": per_step_raw,
               "meaning": "fabric (L2 <-> Infinity Fabric) bytes of all the library's kernels in ONE bench step: "
                          "FETCH_SIZE x factor + WRITE_SIZE, summed over the step's launches",
               "fetch_factors": {k: {"factor": f, "evidence": e} for k, (f, e) in FACTORS.items()},
               "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, profiles/{tag}_{wl}_pmc.csv",
               "kernels": detail}
        alg = bench["roofline"]["algorithmic_bytes_per_launch"] if bench else 0
        if alg and per_step < 0.98 * alg:
            out["note"] = ("below the bytes the kernels must fetch at least once: FETCH_SIZE under-reports this access pattern by an "
                           "uncalibrated factor between 1 and 2 (see fetch_factors for the calibrated kernels) -- a lower bound only")
        json.dump(out, open(os.path.join(dst, f"traffic_{wl}.json"), "w"), indent=1)
        print(f"{wl:6s} value {bench['value']:12.0f}  traffic/step {per_step / 1e6:9.1f} MB (raw {per_step_raw / 1e6:9.1f})  "
              f"= {per_step / alg if alg else 0:5.2f}x algorithmic")
