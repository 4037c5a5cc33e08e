#!/bin/bash
# SQ counters of the planned kernel (compile-time plans) on p60 / p96 and, for comparison on the same box, the tuned K1 on c2
R=${GRAFT_REPO_ROOT:-/root/repo}
G1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU"
for wl in ${PLANNED_WORKLOADS:-p60 p96}; do
  bash $R/tools/pmc.sh r04_${wl}_sq "$G1" "$G2" -- --workload $wl --no-others --sustain-s 0 --steps 20 --warmup 5
  python3 $R/tools/pmc_table.py $R/gpurun_out/prof_r04_${wl}_sq > $R/gpurun_out/r04_${wl}_sq_pmc.csv
  cut -d, -f2- $R/gpurun_out/r04_${wl}_sq_pmc.csv | grep -v "^counter"
done
