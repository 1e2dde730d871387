#!/bin/bash
# same-box A/B: lib/libhgwarp_prev.so (HGWARP_LIB) against lib/libhgwarp.so, alternating; args: configs sources [sweep args]
export TMPDIR=/tmp
o=$PWD/gpurun_out/ab_prev; mkdir -p $o; : > $o/ab.log
for rep in 1 2; do
for lib in prev cur; do
  if [ $lib = prev ]; then export HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_prev.so; else unset HGWARP_LIB; fi
  python tools/sweep.py ${1:-C3,C4,C2} phase=-1 --sources ${2:-shared,distinct} 2>&1 | grep "config\|rror" | sed "s/^/$lib /" | cut -c1-175 >> $o/ab.log
  python tools/sweep.py C5 phase=-1 --sources ${2:-shared,distinct} 2>&1 | grep "config\|rror" | sed "s/^/$lib /" | cut -c1-175 >> $o/ab.log
done; done
sort -s -k3,3 -k5,5 -k7,7 $o/ab.log | awk '{print $1, $3, $7, $13, $15}' 
