#!/bin/bash
# Upper bound of what an MFMA row-DFT INSIDE K1 (N = 64) could return, measured same-box, interleaved:
#   v0  the product kernel
#   v1  the row arithmetic removed from the VALU (-DMOF_ABLATE_ROWS: S1's radix-16 butterfly and S2's row radix-4 with its
#       twiddles; loads, conversions, LDS traffic, barriers unchanged; results wrong by design) -- an MFMA form still has to
#       convert pixels to f16 fragments and to assemble Z = A + iB from its accumulators, so this is MORE than it can save
#   v2  v1 with 16 KB more LDS per workgroup (MOF_PC_EXTRA_LDS): where the hi + lo f16 DFT matrix (2 x 64 x 64 x 2 B) has
#       to live -- 64 VGPRs of B fragments do not fit beside K1's 109 -- which costs the fourth workgroup per CU
#   v3  the product kernel with the same 16 KB (what 3 workgroups per CU cost on their own)
#   v4  only S1's radix-16 butterfly removed (-DMOF_ABLATE_S1): the ceiling of the one MFMA form whose matrix fits in registers
#       (W16 hi + lo = 32 x 64 f16 = 16 VGPRs of A fragments; 8 v_mfma_f32_32x32x16_f16 per wave)
# usage (GPU box): bash tools/ab_mfma_bound.sh
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R/mrs_optic_flow_amd/csrc
BASE="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
OTHERS=$(ls *.hip | grep -v "^pc_kernel.hip$" | grep -v "^pc_kernel_quad.hip$" | sed 's/\.hip$/.o/')
hipcc --offload-arch=gfx950 $BASE -I../../include -I. -c -o /tmp/abm_0.o pc_kernel.hip
hipcc --offload-arch=gfx950 $BASE -DMOF_ABLATE_ROWS -I../../include -I. -c -o /tmp/abm_1.o pc_kernel.hip
hipcc --offload-arch=gfx950 $BASE -DMOF_ABLATE_S1 -I../../include -I. -c -o /tmp/abm_4.o pc_kernel.hip
hipcc --offload-arch=gfx950 -shared -o /tmp/libmof_abm_4.so $OTHERS /tmp/abm_4.o -ldl
hipcc --offload-arch=gfx950 -shared -o /tmp/libmof_abm_0.so $OTHERS /tmp/abm_0.o -ldl
hipcc --offload-arch=gfx950 -shared -o /tmp/libmof_abm_1.so $OTHERS /tmp/abm_1.o -ldl
run() {  # <lib> <extra lds>
  MOF_PC_EXTRA_LDS=$2 MOF_LIB_PATH=$1 python3 $R/bench.py --no-cpu-baseline --no-others --sustain-s 0 --workload c2 | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["roofline"]["kernel_ms"],4))'
}
for rep in 1 2 3; do
  echo "rep $rep v0 product            : $(run /tmp/libmof_abm_0.so 0)"
  echo "rep $rep v1 rows off the VALU  : $(run /tmp/libmof_abm_1.so 0)"
  echo "rep $rep v2 v1 + 16 KB LDS     : $(run /tmp/libmof_abm_1.so 16384)"
  echo "rep $rep v3 product + 16 KB LDS: $(run /tmp/libmof_abm_0.so 16384)"
  echo "rep $rep v4 S1 butterfly off   : $(run /tmp/libmof_abm_4.so 0)"
done
