#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3fwd; rm -rf $o; mkdir -p $o
python tools/bench_forward.py 8 20 > $o/fwd_F8.log 2>&1; python tools/bench_forward.py 1 40 > $o/fwd_F1.log 2>&1
cat $o/fwd_F8.log $o/fwd_F1.log | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(d['frames'], d['case'], d.get('scatter_gather_us_per_frame'), d.get('tiles_us_per_frame'), d.get('same_bytes'))
"
