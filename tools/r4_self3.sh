#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r4self3; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.log
for rep in 1 2; do
HG_DEBUG_STATUS=1 python tools/sweep.py C5 self_spans=0,1 --sources shared,distinct 2>&1 | grep "config\|rror\|debug" | cut -c1-220
done
HG_DEBUG_STATUS=1 python tools/sweep.py G16,G24,G40,G64 self_spans=0,1 --sources shared 2>&1 | grep "config\|rror\|debug" | cut -c1-220
HG_DEBUG_STATUS=1 python tools/sweep.py G40,G64 self_spans=0,1 --sources distinct 2>&1 | grep "config\|rror\|debug" | cut -c1-220
