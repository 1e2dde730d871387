#!/bin/bash
# round 6, call 19: gathers by element index in every pixel kernel (HG_PW_IDX + HG_GEO_IDX): full parity, then A/B
export TMPDIR=/tmp
o=$PWD/gpurun_out/c19; rm -rf $o; mkdir -p $o
HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_pi.so timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $o/parity_pi.log 2>&1; echo "parity pi rc=$?"; tail -2 $o/parity_pi.log
bash tools/ab_libs.sh "cur _pi" C3,C4,C5,G24,T20x60,C2 shared 2 > $o/ab_shared.txt 2>&1; cat $o/ab_shared.txt
bash tools/ab_libs.sh "cur _pi" C3,C4,C5 distinct 2 > $o/ab_distinct.txt 2>&1; cat $o/ab_distinct.txt
