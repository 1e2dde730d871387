#!/bin/bash
# forward warps, same-box A/B: lib/libhgwarp_prev.so (HGWARP_LIB) against lib/libhgwarp.so; then a kernel trace of the 10x10 piecewise case
export TMPDIR=/tmp
o=$PWD/gpurun_out/forward_ab; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward" > $o/pytest.log 2>&1; tail -2 $o/pytest.log
for rep in 1 2; do for lib in prev cur; do
  if [ $lib = prev ]; then export HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_prev.so; else unset HGWARP_LIB; fi
  for F in 8 1; do python tools/bench_forward.py $F $((F==1?40:20)) 2>&1 | grep "^{" | sed "s/^/$lib /" >> $o/fwd.log; done
done; done
python - $o/fwd.log <<'PY'
import sys,json,collections
best=collections.defaultdict(lambda: 1e9)
for ln in open(sys.argv[1]):
    lib,js=ln.split(' ',1); d=json.loads(js)
    k=(d['frames'],d['case'],lib); best[k]=min(best[k],d['tiles_us_per_frame'])
    assert d.get('same_bytes',True), ln
for (F,case,lib) in sorted(best):
    if lib=='cur': print(F, case.ljust(32), 'prev', best[(F,case,'prev')], 'cur', best[(F,case,'cur')])
PY
unset HGWARP_LIB
for F in 8 1; do
  (cd /tmp && FWD_ONLY="piecewise 10x10" rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_F$F -o p -- python $GRAFT_REPO_ROOT/tools/bench_forward.py $F 30 > $o/prof_F$F.log 2>&1)
  echo "== F=$F piecewise 10x10"; python - $o/prof_F$F/p_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]: print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'])
PY
done
