#!/bin/bash
# round 3 evidence pass: bench lines of every config, rocprofv3 summaries (kernel trace + PMC passes, each its own run), F = 1..8 latency
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3ev; rm -rf $o; mkdir -p $o
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench default rc=$?"
for c in C5 C4 C2; do
  timeout 600 python bench.py --config $c $( [ $c = C5 ] && echo --frames 8 ) > $o/bench_$c.json 2> $o/bench_$c.err; echo "bench $c rc=$?"
done
bash tools/profile_round.sh r03_C3 --sources shared --points resident > $o/prof_C3.log 2>&1
bash tools/profile_round.sh r03_C3_distinct --sources distinct > $o/prof_C3d.log 2>&1
bash tools/profile_round.sh r03_C5 --config C5 --frames 8 --sources shared --points resident > $o/prof_C5.log 2>&1
bash tools/profile_round.sh r03_C5_distinct --config C5 --frames 8 --sources distinct > $o/prof_C5d.log 2>&1
bash tools/profile_round.sh r03_C4 --config C4 --sources shared --points resident > $o/prof_C4.log 2>&1
bash tools/profile_round.sh r03_C4_distinct --config C4 --sources distinct > $o/prof_C4d.log 2>&1
bash tools/profile_round.sh r03_C2 --config C2 --sources shared > $o/prof_C2.log 2>&1
bash tools/profile_round.sh r03_C2_distinct --config C2 --sources distinct > $o/prof_C2d.log 2>&1
# the default command, traced as it is (both layouts + the fresh-points region)
mkdir -p $o/default_trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $o/default_trace/trace -o t -- python $OLDPWD/bench.py --no-cpu-baseline > $o/default_trace/bench_under_trace.log 2>&1)
python tools/profile_summary.py $o/default_trace > $o/default_trace/summary.txt 2>&1; find $o/default_trace -name "*.db" -delete
bash tools/r3_lat.sh > $o/latency.log 2>&1
for s in 0.0 0.026 0.5; do timeout 120 tools/bin/calib_shape --slope $s --pad 12 >> $o/calib_shape.log 2>&1; done
ls $o gpurun_out | head -40
# forward warps: scatter + gather vs the tile kernels (4K batch of 8, single frames, 1080p batch of 8)
{ python tools/bench_forward.py 8 30; python tools/bench_forward.py 1 40; python tools/bench_forward.py 8 30 1920 1080; } 2>/dev/null | grep "^{" > $o/forward.jsonl
python tools/fmt_forward.py $o/forward.jsonl > $o/forward_tiles.txt
