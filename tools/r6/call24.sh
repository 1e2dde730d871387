#!/bin/bash
# round 6, call 24: k_pw_tile with 3 blocks in flight (76 VGPRs) and, in the room that leaves, the rows' byte offsets in the stores' scalar offset
export TMPDIR=/tmp
o=$PWD/gpurun_out/c24; rm -rf $o; mkdir -p $o
HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_pb3s.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile or auto" > $o/parity.log 2>&1; echo "parity pb3s rc=$?"; tail -2 $o/parity.log
bash tools/ab_libs.sh "cur _pb3 _pb3s" C5,G64,G40 shared 3 > $o/ab_shared.txt 2>&1; cat $o/ab_shared.txt
bash tools/ab_libs.sh "cur _pb3 _pb3s" C5,C4,C3 distinct 2 > $o/ab_distinct.txt 2>&1; cat $o/ab_distinct.txt
