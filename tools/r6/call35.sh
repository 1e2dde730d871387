#!/bin/bash
# round 6, call 35: k_pw_rows<SELF = 2> -- the pixel phase span by span (first light: same bytes as the window walk? kernel time?)
export TMPDIR=/tmp
o=$PWD/gpurun_out/c35; rm -rf $o; mkdir -p $o
timeout 600 python tools/sweep.py C3,C4,G16,C3flat span_major=0,1,0,1 --sources shared 2>&1 | grep "config\|rror\|Trace" | cut -c1-220 | tee $o/sweep.txt
timeout 300 python tools/sweep.py C3 span_major=0,1 --sources distinct 2>&1 | grep "config\|rror\|Trace" | cut -c1-220 | tee -a $o/sweep.txt
