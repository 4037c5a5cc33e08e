#!/bin/bash
# The round's fuzz record: runs the random-geometry fuzzers of the FINAL build with a fixed seed list and writes
# ONE tracked summary -- seed, patches checked, f32-limited patches, mismatches -- that DESIGN.md quotes and nothing else.
#   usage (on the GPU box): bash tools/fuzz_round.sh <round tag> [seeds...]      -> gpurun_out/<tag>_fuzz.txt (copy to profiles/)
# Every seed's full log stays under gpurun_out/<tag>_fuzz_logs/. Exit code 1 if any seed reports a mismatch.
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}; shift
SEEDS=${@:-"501 502 503 504 505 506 507 508 20261004 101 202 3333"}
TRIALS=${MOF_FUZZ_TRIALS:-40}
OUT=$R/gpurun_out/${TAG}_fuzz.txt
LOGS=$R/gpurun_out/${TAG}_fuzz_logs
mkdir -p $LOGS
{
  echo "# tools/fuzz_round.sh $TAG: library $(md5sum $R/mrs_optic_flow_amd/libmof_hip.so | cut -c1-12), git $(git -C $R rev-parse --short HEAD 2>/dev/null || echo '-'), $(date -u +%FT%TZ)"
  [ -n "$MOF_FUZZ_LARGE" ] && echo "# MOF_FUZZ_LARGE=$MOF_FUZZ_LARGE: every FftMethod / video trial at a patch size in 193 .. 400 (or the band lo-hi given)"
  echo "# fft_sr_fuzz.py <seed> $TRIALS 12: random FftMethod layouts at random patch sizes + estimator settings + sequence trials; bm_fuzz.py <seed> 60"
  echo "# bars: tests/tolerances.py (1e-4 px against both oracles; a patch that misses it is classified from its input pixels, tests/conditioning.py, and held to 1e-4 + 2 x the scatter of independent f32 transforms on it, <= 1e-3 px; unpinned -- integer peak only -- where those scatter further)"
} > $OUT
rc=0
for s in $SEEDS; do
  MOF_FUZZ_RECORDS=$LOGS/records_$s.json python3 $R/tools/fft_sr_fuzz.py $s $TRIALS 12 > $LOGS/fft_sr_$s.log 2>&1; r1=$?
  echo "seed $s  fft_sr rc=$r1 | $(grep -E '^fft:' $LOGS/fft_sr_$s.log | head -1) | $(grep -E '^sr:|estimator' $LOGS/fft_sr_$s.log | head -1) | $(grep -E '^sequence modes' $LOGS/fft_sr_$s.log | head -1) | $(grep -E '^front ends' $LOGS/fft_sr_$s.log | head -1)" >> $OUT
  grep -E "MISMATCH" $LOGS/fft_sr_$s.log | head -5 | sed 's/^/    /' >> $OUT
  grep -E "^off the plain bar" $LOGS/fft_sr_$s.log | sed 's/^/    /' >> $OUT
  [ $r1 -ne 0 ] && rc=1
  echo "seed $s fft_sr done (rc $r1)"
done
for s in 77 91 5; do
  python3 $R/tools/bm_fuzz.py $s 60 > $LOGS/bm_$s.log 2>&1; r2=$?
  echo "seed $s  bm rc=$r2 | $(tail -1 $LOGS/bm_$s.log)" >> $OUT
  grep -q "mismatches: 0" $LOGS/bm_$s.log || rc=1
  echo "seed $s bm done"
done
echo "# overall rc=$rc" >> $OUT
cat $OUT
exit $rc
