#!/bin/bash
# round 6, call 32: k_pw_rows is bound by vector issue since the sub-bands (call 31): three cuts in the window loop -- the overlap ballot as two direct
# compares (no v_cndmask + v_cmp_ne), lo | hi << 16 in one register (one v_readlane less per span and window), no zero fill of pixels nobody stores
export TMPDIR=/tmp
o=$PWD/gpurun_out/c32; rm -rf $o; mkdir -p $o
HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_v3.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $o/parity_v3.log 2>&1; echo "parity(v3) rc=$?"; tail -2 $o/parity_v3.log
bash tools/ab_libs.sh "cur _v1 _v2 _v3" C3,C4,G16 shared 2 2>&1 | tee $o/ab.txt
