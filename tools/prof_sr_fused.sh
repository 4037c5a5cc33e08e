#!/bin/bash
# Counters of K56 (sr_fused_kernel.hip) on c5: kernel stats, then SQ / TCP / TCC groups (separate --pmc passes)
R=${GRAFT_REPO_ROOT:-/root/repo}
export MOF_SR_FUSED=1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fz && rocprofv3 --kernel-trace --stats --kernel-include-regex "mof::" --output-format csv -d /tmp/fz -- python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --workload c5 --steps 10 --warmup 3 > /tmp/fz.log 2>&1
python3 - $(find /tmp/fz -name "*kernel_stats.csv" | head -1) <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mof::" in r["Name"]: print(f'{float(r["AverageNs"])/1e3:10.1f} us x {int(r["Calls"]):5d}  {r["Name"][:100]}')
PY
G1="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
G3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum"
G4="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
G5="FETCH_SIZE"
G6="WRITE_SIZE"
bash $R/tools/pmc.sh r04_sr_fused "$G1" "$G2" "$G3" "$G4" "$G5" "$G6" -- --workload c5 --no-others --sustain-s 0 --steps 10 --warmup 3
python3 $R/tools/pmc_table.py $R/gpurun_out/prof_r04_sr_fused > $R/gpurun_out/r04_sr_fused_pmc.csv
grep "fused" $R/gpurun_out/r04_sr_fused_pmc.csv | cut -d, -f2-
