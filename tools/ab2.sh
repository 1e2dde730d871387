#!/bin/bash
# prev library vs current with an option sweep
export HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_prev.so
python tools/sweep.py $1 phase=-1 --sources $2 2>&1 | grep config | sed "s/^/prev /" | cut -c1-160
unset HGWARP_LIB
python tools/sweep.py $1 $3 --sources $2 2>&1 | grep config | sed "s/^/cur  /" | cut -c1-175
