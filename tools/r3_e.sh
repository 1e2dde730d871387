#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3e; rm -rf $o; mkdir -p $o
python tools/sweep.py C3,C4 patch=0,1 lds_pad=0,6,12,16 phase=4 --sources distinct 2>&1 | grep "config\|rror" | cut -c1-220 >> $o/pad.log
python tools/sweep.py C5 patch=-1 lds_pad=0,6,12 phase=1 --sources distinct,shared 2>&1 | grep "config\|rror" | cut -c1-220 >> $o/pad.log
cat $o/pad.log | awk '{print $2, $6, $8, $10, $12, $16, $18}'
