#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r4self2; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest.log
for rep in 1 2; do
python tools/sweep.py C3,C4 self_spans=0,1 --sources distinct 2>&1 | grep "config\|rror" | cut -c1-220
python tools/sweep.py C5 self_spans=0,1 --sources shared,distinct 2>&1 | grep "config\|rror" | cut -c1-220
done
python tools/sweep.py G16,G40,G64 self_spans=0,1 --sources shared 2>&1 | grep "config\|rror" | cut -c1-220
