#!/bin/bash
# round 6, call 3: the persistent wave-specialised tile kernel (tile=2) -- parity under its layout, then A/B against k_pw_tile; the new thread tests
export TMPDIR=/tmp
o=$PWD/gpurun_out/c3; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tile_p" > $o/parity_tile_p.log 2>&1; echo "parity tile_p rc=$?"; tail -5 $o/parity_tile_p.log
timeout 600 python -m pytest tests/test_gpu_threads.py -x -q -m gpu > $o/threads.log 2>&1; echo "threads rc=$?"; tail -5 $o/threads.log
for rep in 1 2; do
for c in C4 C5 C3; do python tools/sweep.py $c tile=1,2 --sources distinct 2>&1 | grep "config\|rror" | cut -c1-260; done
done | tee $o/ab_tile_p.txt
python tools/sweep.py C4,C5 tile=1,2 patch=1 --sources shared 2>&1 | grep "config\|rror" | cut -c1-260 | tee $o/ab_tile_p_shared.txt
