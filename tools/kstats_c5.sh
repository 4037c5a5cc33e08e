#!/bin/bash
# kernel-time table of the c5 workload for one library (MOF_LIB_PATH) -- remap ablations etc. usage: kstats_c5.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --kernel-include-regex "mof::" --output-format csv -d $R/gpurun_out/ks_$1 -- python3 $R/bench.py --workload c5 --no-cpu-baseline --no-others --sustain-s 0 --steps 20 --warmup 5 > /dev/null 2>&1
f=$(ls -t $R/gpurun_out/ks_$1/*/*kernel_stats.csv | head -1)
grep "logpolar" $f | cut -d, -f1-4 | sed "s/^/$1: /"
