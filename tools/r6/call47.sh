#!/bin/bash
# round 6, call 47: two contexts in flight (tools/two_ctx.py, R5.13) re-measured on the final library: does C3 still lose now that its kernel no longer waits for the fabric?
export TMPDIR=/tmp
o=$PWD/gpurun_out/c47; rm -rf $o; mkdir -p $o
for rep in 1 2; do python tools/two_ctx.py C3,C4,C5 --sources shared 2>&1 | grep "^{" | cut -c1-300 | tee -a $o/two_ctx.txt; done
python tools/two_ctx.py C3,C4,C5 --sources distinct 2>&1 | grep "^{" | cut -c1-300 | tee -a $o/two_ctx.txt
