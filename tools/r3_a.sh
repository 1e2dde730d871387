#!/bin/bash
# round 3, pass A: phase / windows sweep in the one-source-per-frame (true HBM) layout, plus the shared baseline on this box
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3a; rm -rf $o; mkdir -p $o
python tools/sweep.py C3,C4 phase=1,2,4 --sources distinct,shared > $o/sweep_phase.log 2>&1
python tools/sweep.py C2 geo_windows=1,2,4 --sources distinct,shared > $o/sweep_geo.log 2>&1
python tools/sweep.py C5 patch=-1,0 phase=1,2 --sources distinct,shared > $o/sweep_c5.log 2>&1
cat $o/sweep_phase.log $o/sweep_geo.log $o/sweep_c5.log | grep config | cut -c1-200
