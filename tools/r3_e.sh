#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3e; rm -rf $o; mkdir -p $o
python tools/sweep.py C5 phase=8,6,4,8,6 --sources shared,distinct 2>&1 | grep "config\|rror" | cut -c1-220 >> $o/pad.log
cat $o/pad.log | awk '{print $2, $6, $8, $10, $12, $14,$16}'
