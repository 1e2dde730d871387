#!/bin/bash
# forward warps: full GPU suite + fuzz, then tools/bench_forward.py plain (timings) and under rocprofv3 (kernel durations)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/fwd2_tests.txt
timeout 1500 python tools/fuzz_gpu.py 6000 91 2>&1 | tail -5 > gpurun_out/fwd2_fuzz.txt
o=$PWD/gpurun_out/fwd2_prof; rm -rf $o; mkdir -p $o
for F in 8 1; do python tools/bench_forward.py $F 30; done > $o/bench_plain.txt 2>&1
python tools/bench_forward.py 8 30 1920 1080 >> $o/bench_plain.txt 2>&1
python tools/bench_forward.py 8 30 640 480 >> $o/bench_plain.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $o/trace -o t -- python tools/bench_forward.py 8 10 > $o/bench.log 2>&1
python tools/profile_summary.py $o > $o/summary.txt 2>&1
find $o -name "*.db" -delete
