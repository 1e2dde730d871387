#!/bin/bash
# round 6, call 44: a long fuzz of the final library (fresh seeds): wide, tile kernel forced, self-span row kernel forced, call sequences
export TMPDIR=/tmp
o=$PWD/gpurun_out/c44; rm -rf $o; mkdir -p $o
timeout 1500 python tools/fuzz_gpu.py 20000 9501 2>&1 | tail -1 | tee $o/fuzz.log
FUZZ_TILE=1 timeout 900 python tools/fuzz_gpu.py 4000 9502 2>&1 | tail -1 | tee -a $o/fuzz.log
FUZZ_ROWS=1 timeout 900 python tools/fuzz_gpu.py 4000 9503 2>&1 | tail -1 | tee -a $o/fuzz.log
timeout 1500 python tools/fuzz_seq.py 400 9504 2>&1 | tail -1 | tee -a $o/fuzz.log
