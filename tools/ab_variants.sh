#!/bin/bash
# usage: tools/r4_var.sh MACRO "v1 v2 .." "sweep args": same-box A/B of the variant libraries built by tools/variants.sh (two alternating passes)
export TMPDIR=/tmp
for rep in 1 2; do for v in $2; do
  HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_$1_$v.so python tools/sweep.py $3 2>&1 | grep "config\|rror" | sed "s/^/$1=$v /" | cut -c1-230
done; done
