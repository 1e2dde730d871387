#!/bin/bash
# round 3: parity suite + fuzz, then same-box A/B against the round-2 library (lib/libhgwarp_prev.so)
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3d; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -3 $o/pytest.log
timeout 600 python tools/fuzz_gpu.py ${1:-3000} 778 2>&1 | tail -2 > $o/fuzz.log; cat $o/fuzz.log
bash tools/r3_ab.sh
