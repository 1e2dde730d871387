#!/bin/bash
# round 6, call 42: every asm block without `volatile` (_nv) on what call 37 left out: k_pw_patch / dense grids on a shared source, the row-list forms (single frames), forward warps
export TMPDIR=/tmp
o=$PWD/gpurun_out/c42; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _nv" G24,G40,G64,T20x60,C5flat shared 2 2>&1 | tee $o/ab.txt
bash tools/ab_libs.sh "cur _nv" C3,C4 shared 2 self_spans=0 2>&1 | tee -a $o/ab.txt
bash tools/ab_libs.sh "cur _nv" C3 shared 2 --frames 1 2>&1 | tee -a $o/ab.txt
for rep in 1 2; do for lib in cur nv; do
  if [ $lib = nv ]; then export HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_nv.so; else unset HGWARP_LIB; fi
  python tools/bench_forward.py 8 20 2>&1 | grep "^{" | sed "s/^/$lib /" >> $o/fwd.log
done; done; unset HGWARP_LIB
python - $o/fwd.log <<'PY' | tee $o/fwd.txt
import sys,json,collections
best=collections.defaultdict(lambda: 1e9)
for ln in open(sys.argv[1]):
    lib,js=ln.split(' ',1); d=json.loads(js)
    k=(d['frames'],d['case'],lib); best[k]=min(best[k],d['tiles_us_per_frame'])
for (F,case,lib) in sorted(best):
    if lib=='cur': print(F, case.ljust(32), 'cur', best[(F,case,'cur')], 'nv', best[(F,case,'nv')])
PY
