#!/bin/bash
# round 6, call 41: fuzz of the final library (round_x8 without volatile touches k_geo_fast, k_pw_tile, the one-window k_pw_rows forms): fresh seeds
export TMPDIR=/tmp
o=$PWD/gpurun_out/c41; rm -rf $o; mkdir -p $o
timeout 900 python tools/fuzz_gpu.py 5000 9411 2>&1 | tail -1 | tee $o/fuzz.log
FUZZ_TILE=1 timeout 600 python tools/fuzz_gpu.py 1500 9412 2>&1 | tail -1 | tee -a $o/fuzz.log
timeout 900 python tools/fuzz_seq.py 128 9413 2>&1 | tail -1 | tee -a $o/fuzz.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee -a $o/fuzz.log
