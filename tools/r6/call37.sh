#!/bin/bash
# round 6, call 37: the rounding / span-update asm blocks without `volatile` (the compiler may then schedule the second pixel pair's LDS reads across them): A/B on every config
export TMPDIR=/tmp
o=$PWD/gpurun_out/c37; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _nv" C3,C4,C5,C2,G16 shared 2 2>&1 | tee $o/ab.txt
bash tools/ab_libs.sh "cur _nv" C3,C4,C5,C2 distinct 1 2>&1 | tee -a $o/ab.txt
