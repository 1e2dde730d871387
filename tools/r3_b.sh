#!/bin/bash
o=$PWD/gpurun_out/r3b; rm -rf $o; mkdir -p $o
for pad in 12 24; do for lds in 1 24 28 36 44 60; do
  timeout 120 tools/bin/calib_shape --slope 0.026 --pad $pad --lds $lds >> $o/shape.log 2>&1
done; done
cat $o/shape.log
