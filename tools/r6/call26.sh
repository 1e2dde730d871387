#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/c26; rm -rf $o; mkdir -p $o
python tools/sweep.py C3,C4,G16,C3flat sub_bands=0,3,102,103,104,106,108,116,0,104 --sources shared 2>&1 | grep "config\|rror" | cut -c1-200 | tee $o/sweep.txt
