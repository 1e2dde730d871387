#!/bin/bash
# usage: tools/evidence.sh rNN   (through gpurun): the round's evidence pass on the library as built -- bench lines of every config (default
# command, C5 / C4 / C2), rocprofv3 summaries (kernel trace + PMC passes, each its own run) in both source layouts, the default command
# traced as it is, F = 1..8 latency, forward warps.  Everything lands under gpurun_out/<tag>ev and gpurun_out/profile_<tag>_*;
# `python tools/update_traffic.py N` then copies the summaries to profiles/ and refreshes profiles/hbm_traffic.json.
# profile_summary.py refuses to write a summary whose bench line says "verified": false.
export TMPDIR=/tmp
tag=${1:?usage: tools/evidence.sh rNN}
o=$PWD/gpurun_out/${tag}ev; rm -rf $o; mkdir -p $o
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench default rc=$?"
for c in C5 C4 C2; do
  timeout 600 python bench.py --config $c $( [ $c = C5 ] && echo --frames 8 ) > $o/bench_$c.json 2> $o/bench_$c.err; echo "bench $c rc=$?"
done
for c in C3 C5 C4 C2; do
  extra=""; [ $c != C3 ] && extra="--config $c"; [ $c = C5 ] && extra="$extra --frames 8"
  bash tools/profile_round.sh ${tag}_$c $extra --sources shared --points resident > $o/prof_$c.log 2>&1; echo "profile $c rc=$?"
  bash tools/profile_round.sh ${tag}_${c}_distinct $extra --sources distinct > $o/prof_${c}d.log 2>&1; echo "profile $c distinct rc=$?"
done
mkdir -p $o/default_trace
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $o/default_trace/trace -o t -- python $OLDPWD/bench.py --no-cpu-baseline > $o/default_trace/bench_under_trace.log 2>&1)
python tools/profile_summary.py $o/default_trace > $o/default_trace/summary.txt 2>&1; echo "default trace summary rc=$?"; find $o/default_trace -name "*.db" -delete
bash tools/latency.sh "" > $o/latency.log 2>&1; cp gpurun_out/latency/device.log $o/latency_device.log 2>/dev/null
{ python tools/bench_forward.py 8 30; python tools/bench_forward.py 1 40; python tools/bench_forward.py 8 30 1920 1080; } 2>/dev/null | grep "^{" > $o/forward.jsonl
python tools/fmt_forward.py $o/forward.jsonl > $o/forward_tiles.txt
ls $o
