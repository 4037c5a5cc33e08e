#!/bin/bash
# Round-6 evidence run (one box, the final binaries): for EVERY workload the bench line records -- bench line + rocprofv3 kernel stats +
# FETCH_SIZE / WRITE_SIZE passes (tools/profile.sh) -- then the SQ counter groups (two --pmc passes each, no trace domains) for the
# workloads whose binding the bench line quotes. tools/summarize_round.py r06 condenses the result into profiles/.
# usage (on the GPU box): bash tools/profile_r06.sh [tag] ; parts: MOF_PROFILE_WORKLOADS / MOF_SQ_WORKLOADS override the lists
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
for wl in ${MOF_PROFILE_WORKLOADS:-c2 cal c2seq c3 c4 c4seq c5 c5seq ref refseq c1 bmref refrt reflr callr c3bgr p60 p96 l160 l200 l240 l480}; do
  bash $R/tools/profile.sh ${TAG}_$wl --workload $wl --no-others --sustain-s 0 > $R/gpurun_out/profile_${TAG}_$wl.log 2>&1 || { tail -5 $R/gpurun_out/profile_${TAG}_$wl.log; echo "$wl FAILED"; continue; }
  echo "$wl done: $(head -c 140 $R/gpurun_out/prof_${TAG}_$wl/bench.json)"
done
for wl in ${MOF_SQ_WORKLOADS:-c2 ref c4 c5 c5seq l160 l200 p60 c3}; do
  bash $R/tools/prof_sq.sh ${TAG}_${wl} $wl 2>&1 | grep -v amdgpu.ids | tail -8
  echo "$wl sq done"
done
