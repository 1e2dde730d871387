#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/c9; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu -k "general_path_between or dense_sheared or batch_and_buffer or full_replay or threads" > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $o/pytest.log | cut -c1-300
timeout 900 python tools/census.py > $o/census.txt 2> $o/census.err; echo "census rc=$?"; cat $o/census.txt | cut -c1-330; tail -5 $o/census.err
