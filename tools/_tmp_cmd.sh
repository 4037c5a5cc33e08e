set -e
cd mrs_optic_flow_amd/csrc
B="-O3 -std=c++17 -fPIC -fno-slp-vectorize -Wno-unused-parameter -Wno-unused-function"
OTHERS=$(ls *.hip | grep -v "^pc_half_kernel.hip$" | grep -v "^pc_kernel_quad.hip$" | sed 's/\.hip$/.o/')
hipcc --offload-arch=gfx950 $B -DMOF_HALF_SHIFT=3 -I../../include -I. -c -o /tmp/ab0.o pc_half_kernel.hip
hipcc --offload-arch=gfx950 -shared -o /tmp/libmof_ab_0.so $OTHERS /tmp/ab0.o -ldl
cd ../..
MOF_FFT_HALF=1 python tools/check_half.py 96 128 93 125 > gpurun_out/r05_shift_check.txt 2>&1 || { tail -20 gpurun_out/r05_shift_check.txt; exit 1; }
tail -6 gpurun_out/r05_shift_check.txt
for rep in 1 2 3; do for wl in p96 c4; do for v in 0 1; do
  if [ $v == 0 ]; then L=/tmp/libmof_ab_0.so; else L=$PWD/mrs_optic_flow_amd/libmof_hip.so; fi
  X=""; E="MOF_X=1"; if [ $wl == c4 ]; then X="--batch 128"; E="MOF_FFT_HALF=1"; fi
  line=$(env $E MOF_LIB_PATH=$L python3 bench.py --no-cpu-baseline --no-others --sustain-s 0 --steps 50 --warmup 10 --workload $wl $X | python3 -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["roofline"]["kernel_ms"],4))')
  echo "rep $rep $wl shift4=$v : $line" >> gpurun_out/r05_half_shift_ab.txt
done; done; done
cat gpurun_out/r05_half_shift_ab.txt
