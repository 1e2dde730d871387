#!/bin/bash
# round 6, call 18: gathers by element index (buffer_load ... idxen, stride-4 descriptor) in k_geo_fast: parity of the geometric tests, A/B on C2
export TMPDIR=/tmp
o=$PWD/gpurun_out/c18; rm -rf $o; mkdir -p $o
HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_gi.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "auto or phase1 or phase4 or geometric or projective or affine or division" > $o/parity_gi.log 2>&1; echo "parity gi rc=$?"; tail -2 $o/parity_gi.log
bash tools/ab_libs.sh "cur _gi" C2 shared,distinct 3 > $o/ab.txt 2>&1; cat $o/ab.txt
