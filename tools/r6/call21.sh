#!/bin/bash
# round 6, call 21: k_pw_rows -- the branches form gather indices, the gathers are issued in one place (GI=1), + one wait before the stores (GI=2)
export TMPDIR=/tmp
o=$PWD/gpurun_out/c21; rm -rf $o; mkdir -p $o
HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_gi2.so timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $o/parity_gi2.log 2>&1; echo "parity gi2 rc=$?"; tail -2 $o/parity_gi2.log
bash tools/ab_libs.sh "cur _gi1 _gi2" C3,C4,G16 shared 3 > $o/ab.txt 2>&1; cat $o/ab.txt
python tools/latency_f.py 2>/dev/null | tail -4 | cut -c1-200
