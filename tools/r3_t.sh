#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3t; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "auto or phase1 or multi or fresh" > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -4 $o/pytest.log
bash tools/r3_t2.sh C3,C4 | grep table
bash tools/r3_t3.sh C3 | grep abl
python tools/latency_f.py table=0 2>&1 | grep "^{" | head -2
python tools/latency_f.py table=1 2>&1 | grep "^{" | head -2
