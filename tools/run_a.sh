#!/bin/bash
# first GPU pass of round 2: tests, bench (shared + distinct), calibration, C2 profile
export TMPDIR=/tmp
o=$PWD/gpurun_out/r2a; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -3 $o/pytest.log
for c in C3 C5 C4 C2; do
  timeout 600 python bench.py --config $c $( [ $c = C5 ] && echo --frames 8 ) > $o/bench_$c.json 2> $o/bench_$c.err; echo "bench $c rc=$?"
done
timeout 600 tools/calibrate_pmc.sh > $o/calib.log 2>&1; echo "calib rc=$?"
cp gpurun_out/calib/calib.json $o/ 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats -d $o/trace_C2 -o t -- python bench.py --config C2 --no-cpu-baseline --sources shared > $o/trace_C2.log 2>&1; echo "trace C2 rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats -d $o/trace_C3d -o t -- python bench.py --no-cpu-baseline --sources distinct > $o/trace_C3d.log 2>&1; echo "trace C3 distinct rc=$?"
find $o -name "*stats*" | head
