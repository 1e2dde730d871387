#!/bin/bash
# evidence refresh after the PH / round_x4 change: GPU tests, fuzz, bench lines, C3 + C4 rocprofv3 summaries
export TMPDIR=/tmp
o=$PWD/gpurun_out/final2; rm -rf $o; mkdir -p $o
timeout 2400 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
timeout 900 python tools/fuzz_gpu.py 6000 555 2>&1 | tail -2 > $o/fuzz.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $o/smoke.log 2>&1
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench default rc=$?"
for c in C5 C4 C2; do
  timeout 600 python bench.py --config $c $( [ $c = C5 ] && echo --frames 8 ) > $o/bench_$c.json 2> $o/bench_$c.err; echo "bench $c rc=$?"
done
bash tools/profile_round.sh r02_C3 --sources shared > $o/prof_C3.log 2>&1
bash tools/profile_round.sh r02_C4 --config C4 --sources shared > $o/prof_C4.log 2>&1
bash tools/run_d.sh > $o/prof_default.log 2>&1
tail -2 $o/pytest.log; cat $o/fuzz.log; tail -1 $o/smoke.log
