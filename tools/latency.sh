#!/bin/bash
# F = 1, 2, 4, 8 single-launch latency: host view + rocprofv3 device view, for each option string given ("self_spans=0" "self_spans=1" ...)
export TMPDIR=/tmp
o=$PWD/gpurun_out/latency; rm -rf $o; mkdir -p $o
for opt in "$@"; do
  tag=$(echo $opt | tr ' =' '__'); tag=${tag:-default}
  tag=${tag:-default}
  python tools/latency_f.py $opt 2>&1 | grep "^{" | sed "s/^/$tag host /"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $o/trace_$tag -o t -- python $OLDPWD/tools/latency_f.py --trace $opt > $o/trace_$tag.log 2>&1)
  python tools/latency_f.py --parse $o/trace_$tag | sed "s/^/$tag device /" | tee -a $o/device.log
done
find $o -name "*.db" -delete
