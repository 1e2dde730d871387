#!/bin/bash
# round 6, call 40: evidence pass of the final library (round_x8 without volatile): tools/evidence.sh r06 + the node host path
export TMPDIR=/tmp
o=$PWD/gpurun_out/c40; rm -rf $o; mkdir -p $o
rocm-smi --showclocks --showpower > $o/rocm_smi.txt 2>&1
bash tools/evidence.sh r06 2>&1 | tail -14
bash tools/node_host_path.sh 2>&1 | tail -3 | cut -c1-300
