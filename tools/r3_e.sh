#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3e; rm -rf $o; mkdir -p $o
python tools/sweep.py C3,C4 xcc_rotate=-1,0,1 --sources shared 2>&1 | grep "config\|rror" | cut -c1-220 >> $o/pad.log
python tools/sweep.py C5 xcc_rotate=-1,0 --sources shared 2>&1 | grep "config\|rror" | cut -c1-220 >> $o/pad.log
cat $o/pad.log | awk '{print $2, $6, $8, $10, $12, $14,$16}'
