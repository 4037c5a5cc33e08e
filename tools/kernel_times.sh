#!/bin/bash
# Per-kernel average times of one bench workload under one library (rocprofv3 --kernel-trace --stats): tools/kernel_times.sh <lib.so> <workload> [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
LIB=$1; WL=$2; shift 2
D=$R/gpurun_out/kt_$(basename $LIB .so)_$WL
rm -rf $D; mkdir -p $D
cd /tmp && export TMPDIR=/tmp
export MOF_LIB_PATH=$R/$LIB
timeout -k 10 300 rocprofv3 --kernel-trace --stats --kernel-include-regex "mof::" --output-format csv -d $D -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-others --sustain-s 0 --steps 10 --warmup 3 "$@" > $D/bench.json 2> $D/err.log
python3 - $D <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
for r in csv.DictReader(open(f[0])):
    if "mof::" in r["Name"]: print(f'   {r["Name"][:78]:78s} {float(r["AverageNs"]) / 1e3:9.1f} us x {r["Calls"]}')
PY
