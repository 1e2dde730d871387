#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3g; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl.py -x -q -m gpu -k "multi or rccl or fresh or forward" > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -15 $o/pytest.log
