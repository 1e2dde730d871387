#!/bin/bash
# round 6, call 4: where the persistent tile kernel's time goes (prologue wave alone / pixel waves alone), other shapes of it; thread tests
export TMPDIR=/tmp
o=$PWD/gpurun_out/c4; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_threads.py -x -q -m gpu > $o/threads.log 2>&1; echo "threads rc=$?"; tail -5 $o/threads.log
bash tools/ab_libs.sh "cur _tp1 _tp2 _tpb4 _tp7" C4,C5,C3 distinct 1 tile=2 > $o/ab.txt 2>&1; cat $o/ab.txt
python tools/sweep.py C4,C5,C3 tile=1 --sources distinct 2>&1 | grep "config" | cut -c1-200
