#!/bin/bash
# round 6, call 23: heaviest row bands first (longest-processing-time-first frame order per XCD): parity, then A/B against the plain rotating order
export TMPDIR=/tmp
o=$PWD/gpurun_out/c23; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $o/parity.log 2>&1; echo "parity rc=$?"; tail -2 $o/parity.log
bash tools/ab_libs.sh "_nolpt cur" C4,C3,C5,G16 shared 3 > $o/ab_shared.txt 2>&1; cat $o/ab_shared.txt
bash tools/ab_libs.sh "_nolpt cur" C4,C3,C5 distinct 3 > $o/ab_distinct.txt 2>&1; cat $o/ab_distinct.txt
