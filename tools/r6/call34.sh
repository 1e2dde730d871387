#!/bin/bash
# round 6, call 34: lo | hi << 16 in one register + two direct ballots in k_pw_rows' window loop (_v2) against the library, 3 alternating passes;
# safe_spans / phase re-checked on C3 / C4 now that the fabric is not the limit any more
export TMPDIR=/tmp
o=$PWD/gpurun_out/c34; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _v2" C3,C4,G16 shared 3 2>&1 | tee $o/ab.txt
python tools/sweep.py C4,C3 safe_spans=0,1,0,1 --sources shared 2>&1 | grep "config\|rror" | cut -c1-200 | tee $o/sweep.txt
python tools/sweep.py C4,C3 phase=4,2,4,2 --sources shared 2>&1 | grep "config\|rror" | cut -c1-200 | tee -a $o/sweep.txt
