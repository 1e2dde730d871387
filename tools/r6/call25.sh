#!/bin/bash
# round 6, call 25: k_pw_rows sub-bands (loop interchange: every frame of a sub-band before the next sub-band) -- parity under a layout that forces them, sweep
export TMPDIR=/tmp
o=$PWD/gpurun_out/c25; rm -rf $o; mkdir -p $o
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "phase1 or rows1 or auto" > $o/parity.log 2>&1; echo "parity rc=$?"; tail -2 $o/parity.log
python tools/sweep.py C3,G16,C3flat sub_bands=0,2,3,4,6,8,0,3 --sources shared 2>&1 | grep "config\|rror" | cut -c1-200 | tee $o/sweep.txt
python tools/sweep.py C4 sub_bands=0,3,0,3 xcc_rotate=0 --sources shared 2>&1 | grep "config\|rror" | cut -c1-200 | tee -a $o/sweep.txt
