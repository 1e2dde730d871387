#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward" 2>&1 | tail -12 > gpurun_out/fwd_tests.txt
timeout 600 python tools/bench_forward.py 8 20 > gpurun_out/fwd_bench.txt 2>&1
