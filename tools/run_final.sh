#!/bin/bash
# final evidence pass: GPU tests, smoke, the bench lines of every config (default command first)
export TMPDIR=/tmp
o=$PWD/gpurun_out/final; rm -rf $o; mkdir -p $o
timeout 2400 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1; echo "smoke rc=$?" >> $o/smoke.log
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench default rc=$?"
for c in C5 C4 C2; do
  timeout 600 python bench.py --config $c $( [ $c = C5 ] && echo --frames 8 ) > $o/bench_$c.json 2> $o/bench_$c.err; echo "bench $c rc=$?"
done
tail -2 $o/pytest.log; tail -2 $o/smoke.log
