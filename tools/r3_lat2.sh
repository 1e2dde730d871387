#!/bin/bash
export TMPDIR=/tmp
for t in 64 128 256; do echo "tri_threads=$t"; python tools/latency_f.py tri_threads=$t 2>&1 | grep "^{"; done
python tools/sweep.py C3,C4 tri_threads=64,128,256 --sources shared 2>&1 | grep "config" | awk '{print $2, $6, $8, $10, $12, $14,$16}'
python tools/sweep.py C5 tri_threads=64,128,256 --sources shared 2>&1 | grep "config" | awk '{print $2, $6, $8, $10, $12, $14,$16}'
