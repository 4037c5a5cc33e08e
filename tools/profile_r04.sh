#!/bin/bash
# Round-4 evidence run (one box): bench line + rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes for the workloads whose
# records were stale or uncalibrated (c2, c2seq, ref: the r03 kernels had only r02 SQ counters; reflr + its new DS = 4 calibration
# workload callr), then the SQ counter groups for c2 / c2seq / ref (separate --pmc passes, no trace domains).
# usage (on the GPU box): bash tools/profile_r04.sh [tag]
set -e
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-/root/repo}
for wl in ${MOF_PROFILE_WORKLOADS:-c2 c2seq ref callr reflr}; do
  bash $R/tools/profile.sh ${TAG}_$wl --workload $wl --no-others --sustain-s 0 > $R/gpurun_out/profile_${TAG}_$wl.log 2>&1 || { tail -5 $R/gpurun_out/profile_${TAG}_$wl.log; exit 1; }
  echo "$wl done: $(head -c 160 $R/gpurun_out/prof_${TAG}_$wl/bench.json)"
done
G1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
for wl in ${MOF_SQ_WORKLOADS:-c2 c2seq ref}; do
  bash $R/tools/pmc.sh ${TAG}_${wl}_sq "$G1" "$G2" -- --workload $wl --no-others --sustain-s 0 --steps 20 --warmup 5
  python3 $R/tools/pmc_table.py $R/gpurun_out/prof_${TAG}_${wl}_sq > $R/gpurun_out/${TAG}_${wl}_sq_pmc.csv
  echo "$wl sq done"
done
