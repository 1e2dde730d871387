#!/bin/bash
# round 6, call 7: with the slopes hoisted, is k_pw_tile now the better kernel for dense meshes on a SHARED source too (policy: k_pw_patch)?
export TMPDIR=/tmp
o=$PWD/gpurun_out/c7; rm -rf $o; mkdir -p $o
python tools/sweep.py C5,G40,G64,G24,G16,T12x60,T20x60 tile=0,1,0,1 --sources shared 2>&1 | grep "config\|rror" | cut -c1-220 | tee $o/shared_tile.txt
HG_A=30 python tools/sweep.py C5 tile=0,1,0,1 --sources shared 2>&1 | grep "config\|rror" | cut -c1-220 | tee -a $o/shared_tile.txt
HG_A=10 python tools/sweep.py C5 tile=0,1,0,1 --sources shared 2>&1 | grep "config\|rror" | cut -c1-220 | tee -a $o/shared_tile.txt
