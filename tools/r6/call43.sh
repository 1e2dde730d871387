#!/bin/bash
# round 6, call 43: s_setprio 3 / 1 for the span prologue of k_pw_rows<SELF> and k_pw_tile (a latency chain that competes for issue slots with seven waves of pixel loops), 0 for the pixel phase
export TMPDIR=/tmp
o=$PWD/gpurun_out/c43; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _p3 _p1" C3,C4,C5 shared 2 2>&1 | tee $o/ab.txt
bash tools/ab_libs.sh "cur _p3 _p1" C3,C4,C5 distinct 2 2>&1 | tee -a $o/ab.txt
