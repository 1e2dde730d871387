#!/bin/bash
# round 6, call 36: span-by-span pixel phase: 8 pieces in flight, default-policy stores (write-combining of the partial lines at span borders)
export TMPDIR=/tmp
o=$PWD/gpurun_out/c36; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _np8 _nt0 _np8nt0" C3,C3flat shared 1 span_major=0,1 2>&1 | tee $o/ab.txt
