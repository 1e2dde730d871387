export TMPDIR=/tmp
o=$PWD/gpurun_out/fwd2_prof; rm -rf $o; mkdir -p $o
for F in 8 1; do python tools/bench_forward.py $F 30; done > $o/bench_plain.txt 2>&1
python tools/bench_forward.py 8 30 1920 1080 >> $o/bench_plain.txt 2>&1
python tools/bench_forward.py 8 30 640 480 >> $o/bench_plain.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $o/trace -o t -- python tools/bench_forward.py 8 10 > $o/bench.log 2>&1
python tools/profile_summary.py $o > $o/summary.txt 2>&1
find $o -name "*.db" -delete
