#!/bin/bash
# round 6, call 13: k_pw_rows<SELF> with the one-fma span records in registers (v_readlane) instead of LDS -- parity, then A/B on C3 / C4 / G16 shared
export TMPDIR=/tmp
o=$PWD/gpurun_out/c13; rm -rf $o; mkdir -p $o
HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_rr.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $o/parity_rr.log 2>&1; echo "parity rr rc=$?"; tail -3 $o/parity_rr.log
bash tools/ab_libs.sh "cur _rr _rr6" C3,C4,G16 shared 2 > $o/ab.txt 2>&1; cat $o/ab.txt
