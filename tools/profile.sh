#!/bin/bash
# Profiles bench.py on the GPU box: kernel trace + stats, then two PMC passes (HBM read / write bytes).
# usage: tools/profile.sh <tag> [bench args...]     outputs under gpurun_out/prof_<tag>/
set -e
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.err || { cat $OUT/bench.err; exit 1; }
cat $OUT/bench.json
rocprofv3 --kernel-trace --stats --kernel-include-regex "mof::" --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "mof::" --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "mof::" --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -20
