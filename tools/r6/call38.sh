#!/bin/bash
# round 6, call 38: round_x8 without `volatile` (k_geo_fast issues the gathers of its 8 windows in one burst): full GPU suite, then C2 in both source layouts
export TMPDIR=/tmp
o=$PWD/gpurun_out/c38; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.log | cut -c1-300
python tools/sweep.py C2,C1 geo_windows=8,4,8,4 --sources shared,distinct 2>&1 | grep "config\|rror" | cut -c1-200 | tee $o/sweep.txt
