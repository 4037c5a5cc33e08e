#!/bin/bash
# One box, every BASELINE workload: bench line + rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE passes (tools/profile.sh),
# plus the counter calibration workload `cal` and the MFMA row-DFT micro-benchmark. Outputs under gpurun_out/prof_<round>_*.
# usage (on the GPU box): bash tools/profile_all.sh r02
set -e
TAG=$1
R=${GRAFT_REPO_ROOT:-/root/repo}
for wl in ${MOF_PROFILE_WORKLOADS:-c2 cal c2seq c3 c4 c4seq c5 c5seq ref c1 bmref refrt reflr c3bgr}; do
  bash $R/tools/profile.sh ${TAG}_$wl --workload $wl --no-others --sustain-s 0 > $R/gpurun_out/profile_${TAG}_$wl.log 2>&1 || { tail -5 $R/gpurun_out/profile_${TAG}_$wl.log; exit 1; }
  echo "$wl done: $(head -c 160 $R/gpurun_out/prof_${TAG}_$wl/bench.json)"
done
if [ -x $R/tools/ubench/mfma_rowdft ]; then
  mkdir -p $R/gpurun_out/prof_${TAG}_mfma
  cd /tmp && export TMPDIR=/tmp
  $R/tools/ubench/mfma_rowdft 65536 20 > $R/gpurun_out/prof_${TAG}_mfma/result.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${TAG}_mfma/trace -- $R/tools/ubench/mfma_rowdft 65536 20 > $R/gpurun_out/prof_${TAG}_mfma/trace.log 2>&1
  cat $R/gpurun_out/prof_${TAG}_mfma/result.json
fi
