#!/bin/bash
# Per-site LDS bank-conflict counters of the half-tile kernel K1h: which access class of the planned stage routines
# (csrc/pc_plan.hpp: stage_rt / stage_rt_ng) carries SQ_LDS_BANK_CONFLICT. Nine builds of csrc/pc_half_kernel.hip for ONE transform size,
# -DMOF_SITE_ABL=k: k = 0 the product, k = 1 .. 8 one access class removed (results wrong by design):
#   k - 1 = (column walk ? 4 : 0) + (later stage ? 2 : 0) + (write ? 1 : 0)
# a class's counters = the product's minus its ablation build's.
#   build (here or on the box; hipcc cross-compiles):  bash tools/half_site_counters.sh build <M> ["extra flags"]   -> tmp_ab/libmof_site_<M>_<k>.so
#   run   (GPU box):                                   bash tools/half_site_counters.sh run <M> <workload> <tag>    -> gpurun_out/<tag>_site_counters.txt
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
MODE=$1; M=$2
SRC=pc_half_kernel.hip
if [ "$MODE" == "build" ]; then
  EXTRA=$3
  mkdir -p $R/tmp_ab
  cd $R/mrs_optic_flow_amd/csrc
  BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
  OTHERS=$(ls *.hip | grep -v "^$SRC$" | grep -v "^pc_kernel_quad.hip$" | sed 's/\.hip$/.o/')
  for k in 0 1 2 3 4 5 6 7 8; do
    ( hipcc --offload-arch=gfx950 $BASE -DMOF_HALF_ONLY=$M -DMOF_SITE_ABL=$k $EXTRA -I../../include -I. -c -o $R/tmp_ab/site_${M}_$k.o $SRC &&
      hipcc --offload-arch=gfx950 -shared -o $R/tmp_ab/libmof_site_${M}_$k.so $OTHERS $R/tmp_ab/site_${M}_$k.o -ldl && rm -f $R/tmp_ab/site_${M}_$k.o ) &
    [ $(( (k + 1) % 4 )) -eq 0 ] && wait
  done
  wait
  ls -la $R/tmp_ab/libmof_site_${M}_*.so
  exit 0
fi
WL=$3; TAG=$4
OUT=$R/gpurun_out/prof_${TAG}_site
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MOF_FFT_HALF=1
NAMES=("product" "rows first-stage reads" "rows first-stage writes" "rows later-stage reads" "rows later-stage writes" "cols first-stage reads" "cols first-stage writes" "cols later-stage reads" "cols later-stage writes")
for k in 0 1 2 3 4 5 6 7 8; do
  MOF_LIB_PATH=$R/tmp_ab/libmof_site_${M}_$k.so rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS --kernel-include-regex "pc_half" --output-format csv -d $OUT/k$k -- python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --workload $WL --steps 6 --warmup 2 > $OUT/k$k.log 2>&1 || { tail -5 $OUT/k$k.log; exit 1; }
  echo "site build $k done"
done
python3 - "$OUT" "$M" "$WL" > $R/gpurun_out/${TAG}_site_counters.txt <<'PY'
import csv, glob, sys
out, M, wl = sys.argv[1:4]
names = ["product", "rows first-stage reads", "rows first-stage writes", "rows later-stage reads", "rows later-stage writes",
         "cols first-stage reads", "cols first-stage writes", "cols later-stage reads", "cols later-stage writes"]
tot = []
for k in range(9):
    acc, n = {}, {}
    for f in glob.glob(f"{out}/k{k}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            c = row["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
            n[c] = n.get(c, 0) + 1
    tot.append({c: acc[c] / max(n[c], 1) for c in acc})  # per launch
b = tot[0]
print(f"# K1h at M = {M}, workload {wl}: per launch, SQ counters summed over the chip; a class = product - (build without the class)")
print(f"# product: LDS_IDX_ACTIVE {b['SQ_LDS_IDX_ACTIVE']:.4g}, BANK_CONFLICT {b['SQ_LDS_BANK_CONFLICT']:.4g} = {100 * b['SQ_LDS_BANK_CONFLICT'] / b['SQ_LDS_IDX_ACTIVE']:.1f} % of the LDS cycles; INSTS_LDS {b['SQ_INSTS_LDS']:.4g}")
print(f"{'class':28s} {'LDS cycles':>12s} {'conflicts':>12s} {'conflict %':>10s} {'share of all conflicts':>24s} {'LDS instr.':>12s}")
sa = sc = 0.0
for k in range(1, 9):
    t = tot[k]
    da, dc, di = b['SQ_LDS_IDX_ACTIVE'] - t['SQ_LDS_IDX_ACTIVE'], b['SQ_LDS_BANK_CONFLICT'] - t['SQ_LDS_BANK_CONFLICT'], b['SQ_INSTS_LDS'] - t['SQ_INSTS_LDS']
    sa += da; sc += dc
    print(f"{names[k]:28s} {da:12.4g} {dc:12.4g} {100 * dc / max(da, 1):10.1f} {100 * dc / b['SQ_LDS_BANK_CONFLICT']:24.1f} {di:12.4g}")
print(f"{'(everything else)':28s} {b['SQ_LDS_IDX_ACTIVE'] - sa:12.4g} {b['SQ_LDS_BANK_CONFLICT'] - sc:12.4g} {100 * (b['SQ_LDS_BANK_CONFLICT'] - sc) / max(b['SQ_LDS_IDX_ACTIVE'] - sa, 1):10.1f} {100 * (b['SQ_LDS_BANK_CONFLICT'] - sc) / b['SQ_LDS_BANK_CONFLICT']:24.1f}")
PY
cat $R/gpurun_out/${TAG}_site_counters.txt
