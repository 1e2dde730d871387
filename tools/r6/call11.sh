#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/c11; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "^FAILED\|^ERROR" $o/pytest.log | head; tail -4 $o/pytest.log | cut -c1-300
grep -n "Error\|assert" $o/pytest.log | head -20 | cut -c1-250
timeout 900 python tools/census.py > $o/census.txt 2> $o/census.err; echo "census rc=$?"; cut -c1-160 $o/census.txt
