#!/bin/bash
# round 6, call 39: the kernels under two other scheduling strategies of the AMDGPU back end (-mllvm -amdgpu-sched-strategy=max-ilp | max-memory-clause)
export TMPDIR=/tmp
o=$PWD/gpurun_out/c39; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _ilp _mmc" C3,C4,C5,C2 shared 2 2>&1 | tee $o/ab.txt
bash tools/ab_libs.sh "cur _ilp _mmc" C3,C4,C5,C2 distinct 2 2>&1 | tee -a $o/ab.txt
