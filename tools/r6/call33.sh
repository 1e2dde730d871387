#!/bin/bash
# round 6, call 33: what the rounding-mode switches cost (timing variant: the s_setreg pairs around the round-toward-minus-infinity adds replaced by s_nop;
# wrong pixels where a sum rounds up to a half -- same_bytes may read false)
export TMPDIR=/tmp
o=$PWD/gpurun_out/c33; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _nsr" C3,C4,C5,C2 shared 2 2>&1 | tee $o/ab.txt
bash tools/ab_libs.sh "cur _nsr" C3,C2 distinct 1 2>&1 | tee -a $o/ab.txt
