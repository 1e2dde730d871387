#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/c2; rm -rf $o; mkdir -p $o
HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_ne2304.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $o/parity_ne2304.log 2>&1; echo "parity ne2304 rc=$?"
tail -3 $o/parity_ne2304.log
bash tools/ab_libs.sh "cur _ne _t2304 _ne2304" C4,C5,C3 distinct 3 phase=-1 > $o/ab.txt 2>&1
cat $o/ab.txt
bash tools/ab_libs.sh "cur _ne2304" C4,G16 shared 2 tile=1 patch=1 > $o/ab_shared_tile.txt 2>&1
cat $o/ab_shared_tile.txt
timeout 900 python -m pytest tests/test_gpu_rccl.py -x -q -m gpu > $o/rccl_tests.log 2>&1; echo "rccl tests rc=$?"; tail -15 $o/rccl_tests.log
( time timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2>&1 | grep real; echo "bench rc=$?"
python - <<PY
import json
d = json.loads(open('$o/bench_default.json').read().strip().splitlines()[-1])
def show(x): 
    r = x['roofline']; rd = x.get('roofline_distinct') or {}
    print(x['metric'], x['value'], x['ms_per_step'], 'verified', x['verified'], 'frac', r['frac'], 'step_frac', r['step_frac'], 'fabric', r.get('fabric_frac'), 'wfloor', r.get('write_floor_ms'), '| distinct', rd.get('frac'), rd.get('fabric_frac'))
show(d)
for a in d.get('also', []): show(a)
PY
# R5.10: per-channel TCC requests of k_pw_tile on C4 with one source per frame, six processes (is a slow process a hot-spotted one?)
for i in 1 2 3 4 5 6; do
  rm -rf $o/tcc$i
  (cd /tmp && timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ -d $o/tcc$i -o p -- python $OLDPWD/tools/sweep.py C4 phase=-1 --sources distinct > $o/tcc$i.log 2>&1)
  python tools/tcc_channels.py $o/tcc$i k_pw_tile | tee -a $o/tcc_channels.txt
  rm -rf $o/tcc$i
done
