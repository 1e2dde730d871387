#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/c15; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "blank_window or 32767 or reference_state" 2>&1 | tail -15 | cut -c1-250
