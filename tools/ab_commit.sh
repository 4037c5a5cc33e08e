#!/bin/bash
# Same-box A/B of the current library against the kernels of an older commit.
#   local:  tools/ab_commit.sh prepare <commit>        (exports that commit's csrc/ + include/ to tmp_ab/)
#   on box: tools/ab_commit.sh run [bench args...]     (builds tmp_ab into /tmp/lib_old.so, alternates 3x)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" == "prepare" ]; then
  rm -rf $R/tmp_ab && mkdir -p $R/tmp_ab/csrc $R/tmp_ab/include/mof
  for f in $(git -C $R ls-tree --name-only $2 mrs_optic_flow_amd/csrc/); do git -C $R show $2:$f > $R/tmp_ab/csrc/$(basename $f); done
  git -C $R show $2:include/mof.h > $R/tmp_ab/include/mof.h
  echo "prepared tmp_ab from $2"; exit 0
fi
shift
cd $R/tmp_ab/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
OBJS=""
for f in *.hip; do hipcc --offload-arch=gfx950 $BASE -I../include -I. -c -o /tmp/old_${f%.hip}.o $f; OBJS="$OBJS /tmp/old_${f%.hip}.o"; done
hipcc --offload-arch=gfx950 -shared -o /tmp/lib_old.so $OBJS
# the old library is selected through MOF_LIB_PATH: the product library is never overwritten
for rep in 1 2 3; do
  for v in old new; do
    LIB=$R/mrs_optic_flow_amd/libmof_hip.so; [ $v == old ] && LIB=/tmp/lib_old.so
    echo "$v $(MOF_LIB_PATH=$LIB python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --steps 30 "$@" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), d["roofline"]["kernel_ms"])')"
  done
done
