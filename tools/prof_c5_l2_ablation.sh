#!/bin/bash
# L2 residency of the c5 transforms' intermediates, product against the MOF_SR_L2_ABLATE=2 build: TCC hit / miss / request
# counts and fabric bytes per kernel of a c5seq step, one --pmc pass per group and library.
#   here:  bash tools/build_variant.sh srbase sr_seq_kernel.hip,sr_kernel.hip "" ; bash tools/build_variant.sh srl2b sr_seq_kernel.hip,sr_kernel.hip "-DMOF_SR_L2_ABLATE=2"
#   box:   bash tools/prof_c5_l2_ablation.sh  -> gpurun_out/c5_l2_ablation_tcc.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_c5_l2abl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for L in srbase srl2b; do
  i=0
  # (FETCH_SIZE and WRITE_SIZE each in a pass of their own, as MI355X_MICROARCH.md prescribes: together they exceed the hardware's counters,
  #  rocprofv3 aborts and then waits for a dispatch that never completes; every pass under its own timeout)
  for G in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    echo "pass $L: $G"
    MOF_LIB_PATH=$R/tmp_ab/libmof_$L.so timeout -k 10 240 rocprofv3 --pmc $G --kernel-include-regex "sr_rows_real_kernel|sr_cols_seq_kernel|sr_rows_inv_kernel" --output-format csv -d $OUT/${L}_g$i -- python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --workload c5seq --steps 6 --warmup 2 > $OUT/${L}_g$i.log 2>&1 || { tail -5 $OUT/${L}_g$i.log; exit 1; }
    i=$((i+1))
  done
done
python3 - $OUT > $R/gpurun_out/c5_l2_ablation_tcc.txt <<'PY'
import csv, glob, re, sys
out = sys.argv[1]
print("# c5seq, per launch (1024 frames / pairs): product (srbase) against MOF_SR_L2_ABLATE=2 (srl2b: Zh / Dt slots aliased to slot 0, K6s stores no Dt)")
print(f"{'library':8s} {'kernel':22s} {'TCC_REQ':>12s} {'TCC_HIT':>12s} {'TCC_MISS':>12s} {'hit %':>7s} {'FETCH_SIZE MB':>14s} {'WRITE_SIZE MB':>14s}")
for L in ("srbase", "srl2b"):
    acc = {}
    for f in glob.glob(f"{out}/{L}_g*/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            mk = re.search(r"(sr_\w+_kernel)", row["Kernel_Name"])
            k = mk.group(1) if mk else row["Kernel_Name"][:40]
            a = acc.setdefault(k, {})
            c = row["Counter_Name"]
            s, n = a.get(c, (0.0, 0))
            a[c] = (s + float(row["Counter_Value"]), n + 1)
    for k, a in sorted(acc.items()):
        m = {c: s / max(n, 1) for c, (s, n) in a.items()}
        req, hit, miss = m.get("TCC_REQ_sum", 0), m.get("TCC_HIT_sum", 0), m.get("TCC_MISS_sum", 0)
        # FETCH_SIZE / WRITE_SIZE are reported in KiB (tools/summarize_round.py uses the same conversion); raw, no FETCH_SIZE doubling applied
        print(f"{L:8s} {k:22s} {req:12.4g} {hit:12.4g} {miss:12.4g} {100 * hit / max(hit + miss, 1):7.1f} {m.get('FETCH_SIZE', 0) * 1024 / 1e6:14.1f} {m.get('WRITE_SIZE', 0) * 1024 / 1e6:14.1f}")
PY
cat $R/gpurun_out/c5_l2_ablation_tcc.txt
