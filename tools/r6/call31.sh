#!/bin/bash
# round 6, call 31: what bounds k_pw_rows<SELF> on C3 / C4 with a shared source AFTER sub-bands (R6.12): timing variants of the final library
# (temporary compile-time hooks, not in the tree: + 8 dependent fp64 fma per pixel, + 3 ds_read_b128 per pixel, gathers forced out of range,
#  stores dropped by a 0-byte row descriptor, both) against the library itself, one process per library, alternating
export TMPDIR=/tmp
o=$PWD/gpurun_out/c31; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _fma _lds _nog _nos _nogs" C3,C4 shared 2 2>&1 | tee $o/ab.txt
