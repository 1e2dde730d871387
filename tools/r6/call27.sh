#!/bin/bash
# round 6, call 27: dealt sub-bands in k_pw_rows / k_pw_patch / k_pw_tile (frame_group): full parity, then the sweep
export TMPDIR=/tmp
o=$PWD/gpurun_out/c27; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $o/parity.log 2>&1; echo "parity rc=$?"; tail -2 $o/parity.log
python tools/sweep.py C3,G16,G24,G40,G64,T20x60,C3flat,C5,C5flat sub_bands=0,-1,0,-1 --sources shared 2>&1 | grep "config\|rror" | cut -c1-200 | tee $o/sweep.txt
python tools/sweep.py C3,G24,G64 sub_bands=0,2,3,4,2,0 --sources shared 2>&1 | grep "config\|rror" | cut -c1-200 | tee -a $o/sweep.txt
