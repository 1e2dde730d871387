#!/bin/bash
# round 4: self-span path (k_tri_setup + k_pw_rows<SELF>): parity under the layouts that force it, then same-process A/B against row lists
export TMPDIR=/tmp
o=$PWD/gpurun_out/r4self; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "phase1 or auto" > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $o/pytest.log
for rep in 1 2; do
python tools/sweep.py C3,C4 self_spans=0,1 --sources shared 2>&1 | grep "config\|rror" | cut -c1-220
done
python tools/sweep.py C3,C4 self_spans=0,1 --sources distinct 2>&1 | grep "config\|rror" | cut -c1-220
python tools/sweep.py C3 self_spans=0,1 --sources shared --frames 1 2>&1 | grep "config\|rror" | cut -c1-220
python tools/sweep.py C3 self_spans=0,1 --sources shared --frames 8 2>&1 | grep "config\|rror" | cut -c1-220
