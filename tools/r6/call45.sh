#!/bin/bash
# round 6, call 45: the span search of k_pw_rows per 64-pixel PIECE: a span updates only the pieces of the window it reaches (scalar tests and branches; C4's
# spans reach 1.9 of 4 pieces, C3's 2.3): A/B against the library
export TMPDIR=/tmp
o=$PWD/gpurun_out/c45; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _pc" C4,C3,G16 shared 3 2>&1 | tee $o/ab.txt
