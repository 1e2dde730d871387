#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3g; rm -rf $o; mkdir -p $o
timeout 1800 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -6 $o/pytest.log
timeout 300 python tools/fuzz_gpu.py 1500 779 2>&1 | tail -1
