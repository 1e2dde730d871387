#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3t; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -6 $o/pytest.log
timeout 300 python tools/fuzz_gpu.py 1500 781 2>&1 | tail -1
bash tools/r3_t2.sh "C3 C4 C5" | grep -v passed
