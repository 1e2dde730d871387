#!/bin/bash
# same-box A/B over the phase option: prev vs cur
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3ab; mkdir -p $o; : > $o/ab2.log
for rep in 1 2; do
for lib in prev cur; do
  if [ $lib = prev ]; then export HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_prev.so; else unset HGWARP_LIB; fi
  python tools/sweep.py ${1:-C3,C4} phase=1,2,4 --sources ${2:-shared,distinct} 2>&1 | grep "config\|rror" | sed "s/^/$lib /" | cut -c1-175 >> $o/ab2.log
done; done
sort -s -k3,3 -k7,7 -k9,9 $o/ab2.log | awk '{print $1, $3, $7, $9, $13, $15}'
