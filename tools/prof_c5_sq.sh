#!/bin/bash
# SQ counters of every kernel of a c5 step (separate --pmc passes): which of the estimator's transforms are bound by their own work
R=${GRAFT_REPO_ROOT:-/root/repo}
G1="SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
G2="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_WR"
bash $R/tools/pmc.sh r04_c5_sq "$G1" "$G2" -- --workload c5 --no-others --sustain-s 0 --steps 10 --warmup 3
python3 $R/tools/pmc_table.py $R/gpurun_out/prof_r04_c5_sq > $R/gpurun_out/r04_c5_sq_pmc.csv
python3 - $R/gpurun_out/r04_c5_sq_pmc.csv <<'PY'
import csv, sys, collections
d=collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])): d[r['kernel'][:60]][r['counter']]=float(r['mean_per_dispatch'])
for k,c in d.items():
    if 'SQ_BUSY_CU_CYCLES' not in c: continue
    b=c['SQ_BUSY_CU_CYCLES']
    print(f"{k:62s} VALU {c['SQ_INSTS_VALU']*2/(4*b)*100:5.1f}%  LDS {c['SQ_LDS_IDX_ACTIVE']/b*100:5.1f}% (conf {c['SQ_LDS_BANK_CONFLICT']/max(c['SQ_LDS_IDX_ACTIVE'],1)*100:4.1f}%)  waves/CU {c['SQ_WAVE_CYCLES']*4/b:5.1f}  VALU/wave {c['SQ_INSTS_VALU']/max(c.get('SQ_WAVES',1),1):7.0f}  LDS/wave {c['SQ_INSTS_LDS']/max(c.get('SQ_WAVES',1),1):6.0f}  wait_any {c.get('SQ_WAIT_ANY',0)/c['SQ_WAVE_CYCLES']*100:5.1f}%  wait_lds {c.get('SQ_WAIT_INST_LDS',0)/c['SQ_WAVE_CYCLES']*100:5.1f}%")
PY
