#!/bin/bash
# round 6, call 30: long fuzz of the final library on fresh seeds (wide fuzz, tile / rows forced, call sequences)
export TMPDIR=/tmp
o=$PWD/gpurun_out/c30; rm -rf $o; mkdir -p $o
timeout 900 python tools/fuzz_gpu.py 8000 9301 2>&1 | tail -1 | tee $o/fuzz.log
FUZZ_TILE=1 timeout 600 python tools/fuzz_gpu.py 1500 9302 2>&1 | tail -1 | tee -a $o/fuzz.log
FUZZ_ROWS=1 timeout 600 python tools/fuzz_gpu.py 1500 9303 2>&1 | tail -1 | tee -a $o/fuzz.log
timeout 900 python tools/fuzz_seq.py 160 9304 2>&1 | tail -1 | tee -a $o/fuzz.log
