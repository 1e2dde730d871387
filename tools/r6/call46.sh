#!/bin/bash
# round 6, call 46: k_pw_rows' window loop in up to four copies (spans in registers or in LDS x one-fma or long records, decided once per row instead of by
# uniform branches inside the loop: _hb); _hbf also drops the `flag_spans &&` in front of the safe-window ballot (66 VGPRs)
export TMPDIR=/tmp
o=$PWD/gpurun_out/c46; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _hb _hbf" C3,C4,G16 shared 3 2>&1 | tee $o/ab.txt
