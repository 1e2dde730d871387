#!/bin/bash
# round 6, call 6: vector-instruction cuts in k_pw_tile -- (a) block slopes once per tile, (b) row offsets in the store's scalar offset, (c) x stepped in fp64
export TMPDIR=/tmp
o=$PWD/gpurun_out/c6; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile" > $o/parity_tile.log 2>&1; echo "parity tile rc=$?"; tail -3 $o/parity_tile.log
for l in _abc _abc6; do HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp$l.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile and not tile_p" > $o/parity$l.log 2>&1; echo "parity $l rc=$?"; tail -2 $o/parity$l.log; done
bash tools/ab_libs.sh "_base cur _a _abc _abc6" C4,C5,C3 distinct 2 tile=1 > $o/ab.txt 2>&1; cat $o/ab.txt
bash tools/ab_libs.sh "_base cur _abc" C5,G40 shared 2 tile=1 patch=1 > $o/ab_shared.txt 2>&1; cat $o/ab_shared.txt
