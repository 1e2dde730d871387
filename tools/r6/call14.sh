#!/bin/bash
# round 6, call 14: long fuzz of the final library (wide fuzz, tile / rows modes, sequence fuzz), the Node GPU tests
export TMPDIR=/tmp
o=$PWD/gpurun_out/c14; rm -rf $o; mkdir -p $o
timeout 900 python tools/fuzz_gpu.py 4000 6101 2>&1 | tail -2 | tee $o/fuzz.log
FUZZ_TILE=1 timeout 900 python tools/fuzz_gpu.py 800 6102 2>&1 | tail -2 | tee -a $o/fuzz.log
FUZZ_ROWS=1 timeout 900 python tools/fuzz_gpu.py 800 6103 2>&1 | tail -2 | tee -a $o/fuzz.log
timeout 900 python tools/fuzz_seq.py 96 6104 2>&1 | tail -2 | tee $o/fuzz_seq.log
timeout 900 python -m pytest tests/test_js_class.py -q -m gpu 2>&1 | tail -3
