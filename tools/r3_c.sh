#!/bin/bash
o=$PWD/gpurun_out/r3c; rm -rf $o; mkdir -p $o
python tools/sweep.py C4,C3 patch=-1,1 --sources distinct,shared > $o/sweep_patch.log 2>&1
grep config $o/sweep_patch.log | cut -c1-200
