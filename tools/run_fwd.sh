#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward" 2>&1 | tail -25 > gpurun_out/fwd_tests.txt
o=$PWD/gpurun_out/fwd_prof; rm -rf $o; mkdir -p $o
timeout 600 rocprofv3 --kernel-trace --stats -d $o/trace -o t -- python tools/bench_forward.py 8 10 > $o/bench.log 2>&1
python tools/profile_summary.py $o > $o/summary.txt 2>&1
find $o -name "*.db" -delete
