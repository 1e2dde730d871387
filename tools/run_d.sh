#!/bin/bash
# rocprofv3 summary of EXACTLY the default bench command (shared + distinct blocks in one run)
export TMPDIR=/tmp
o=$PWD/gpurun_out/profile_r02_default; rm -rf $o; mkdir -p $o
timeout 900 rocprofv3 --kernel-trace --stats -d $o/trace -o t -- python bench.py > $o/bench_under_trace.log 2>&1
python tools/profile_summary.py $o > $o/summary.txt 2>&1
find $o -name "*.db" -delete
head -12 $o/summary.txt
