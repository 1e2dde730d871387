#!/bin/bash
# usage: tools/gpu_suite.sh [fuzz trials] [seed]: the GPU parity suite, the wide fuzz and the sequence fuzz (run through gpurun)
export TMPDIR=/tmp
o=$PWD/gpurun_out/gpu_suite; rm -rf $o; mkdir -p $o
timeout 1800 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $o/pytest.log
tail -4 $o/pytest.log
timeout 900 python tools/fuzz_gpu.py ${1:-3000} ${2:-778} 2>&1 | tail -2 | tee $o/fuzz.log
timeout 900 python tools/fuzz_seq.py 64 ${2:-778} 2>&1 | tail -2 | tee $o/fuzz_seq.log
