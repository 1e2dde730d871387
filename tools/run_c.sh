#!/bin/bash
# round-2 evidence pass: default bench line, rocprofv3 summaries per config, node host path
o=$PWD/gpurun_out/r2c; mkdir -p $o
timeout 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench default rc=$?"
for c in C5 C4 C2; do timeout 400 python bench.py --config $c $( [ $c = C5 ] && echo --frames 8 ) --no-cpu-baseline > $o/bench_$c.json 2> $o/bench_$c.err; echo "bench $c rc=$?"; done
bash tools/profile_round.sh r02_C3 --sources shared > $o/prof_C3.log 2>&1
bash tools/profile_round.sh r02_C3_distinct --sources distinct > $o/prof_C3d.log 2>&1
bash tools/profile_round.sh r02_C5 --config C5 --frames 8 --sources shared > $o/prof_C5.log 2>&1
bash tools/profile_round.sh r02_C4 --config C4 --sources shared > $o/prof_C4.log 2>&1
bash tools/profile_round.sh r02_C2 --config C2 --sources shared > $o/prof_C2.log 2>&1
timeout 300 tools/calibrate_pmc.sh > $o/calib.log 2>&1
timeout 300 node tools/bench_node.mjs 2 > $o/bench_node.json 2> $o/bench_node.err
ls $o
