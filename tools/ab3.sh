#!/bin/bash
# A/B of two library builds on the same box over the phase option: lib/libhgwarp_prev.so (HGWARP_LIB) vs lib/libhgwarp.so
for rep in 1 2; do
for lib in prev cur; do
  if [ $lib = prev ]; then export HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_prev.so; else unset HGWARP_LIB; fi
  python tools/sweep.py ${1:-C3,C4} phase=2,1,2,4 --sources ${2:-shared} 2>&1 | grep config | sed "s/^/$lib /" | cut -c1-175
done; done
