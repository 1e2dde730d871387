#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3lat; rm -rf $o; mkdir -p $o
python tools/latency_f.py "$@" > $o/host.log 2>&1; cat $o/host.log | grep "^{"
cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $o/trace -o t -- python $OLDPWD/tools/latency_f.py --trace "$@" > $o/trace.log 2>&1; cd $OLDPWD
python tools/latency_f.py --parse $o/trace | tee $o/device.log
find $o -name "*.db" -delete
