set -e
# Generic block-scan plan sweep (XB / bpw overrides) for the c1 and bmref workloads; run on the GPU box.
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "bm or block or cpp_host or two_rank" 2>&1 | tail -3
for wl in c1 bmref; do
  echo "== $wl auto"; MOF_BM_VERBOSE=1 python bench.py --workload $wl --no-others --sustain-s 0 --no-cpu-baseline 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('mof: block scan plan'): plan=l.strip()
    if l.startswith('{'): d=json.loads(l); print(plan); print(d['value'])
"
  for xb in 1 2 3; do for bpw in 1 2 4 8; do
    MOF_BM_XB=$xb MOF_BM_BPW=$bpw python bench.py --workload $wl --no-others --sustain-s 0 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl xb $xb bpw $bpw', d['value'])
" || echo "$wl xb $xb bpw $bpw failed"
  done; done
done
python bench.py --workload c3 --no-others --sustain-s 0 --no-cpu-baseline 2>/dev/null | cut -c1-200
