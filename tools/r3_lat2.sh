#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3lat; rm -rf $o; mkdir -p $o
for t in 256 128; do for mg in 1536 6000; do
  echo "rows1_threads=$t min_row_groups=$mg"; python tools/latency_f.py rows1_threads=$t min_row_groups=$mg 2>&1 | grep "^{"
done; done
