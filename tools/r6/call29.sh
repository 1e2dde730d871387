#!/bin/bash
# round 6, call 29: re-created container, library rebuilt from the committed sources: full GPU suite + the plain bench line + C2 A/B baseline
export TMPDIR=/tmp
o=$PWD/gpurun_out/c29; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.log | cut -c1-300
timeout 600 python bench.py > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; cut -c1-600 $o/bench.json
python tools/sweep.py C2 geo_windows=8,8 --sources shared,distinct 2>&1 | grep "config\|rror" | cut -c1-200 | tee $o/sweep.txt
