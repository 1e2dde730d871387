#!/bin/bash
# usage: tools/variants.sh MACRO "v1 v2 .." [file.hip ...]: builds lib/libhgwarp_<MACRO>_<v>.so, the named translation units (default: the
# kernel files) compiled with -D<MACRO>=<v>, everything else from build/.  For same-box A/B of code variants (HGWARP_LIB=...).
set -e
cd "$(dirname "$0")/../homography.js_amd"
macro=$1; vals=$2; shift 2
files=${@:-csrc/hg_k_piecewise.hip csrc/hg_k_patch.hip}
make -j8 lib/libhgwarp.so > /dev/null
for v in $vals; do
  objs=""
  for o in build/*.o; do
    src=csrc/$(basename ${o%.o})
    if echo " $files " | grep -q " $src "; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result -D$macro=$v -c -o /tmp/var_${v}_$(basename $o) $src &
      objs="$objs /tmp/var_${v}_$(basename $o)"
    else objs="$objs $o"; fi
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libhgwarp_${macro}_$v.so $objs
  echo built lib/libhgwarp_${macro}_$v.so
done
