#!/bin/bash
# SQ counters (two --pmc passes, no trace domains) of one bench workload -> gpurun_out/<tag>_sq_pmc.csv + a one-line-per-kernel table.
# usage (on the GPU box): [ENV=..] bash tools/prof_sq.sh <tag> <workload> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; WL=$2; shift 2
G1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_SALU"
bash $R/tools/pmc.sh ${TAG}_sq "$G1" "$G2" -- --workload $WL --no-others --sustain-s 0 --steps 10 --warmup 3 "$@" || exit 1
python3 $R/tools/pmc_table.py $R/gpurun_out/prof_${TAG}_sq > $R/gpurun_out/${TAG}_sq_pmc.csv
python3 $R/tools/sq_table.py $R/gpurun_out/${TAG}_sq_pmc.csv
