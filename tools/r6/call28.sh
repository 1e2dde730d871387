#!/bin/bash
# round 6, call 28: final library (sub-bands): full GPU suite, fuzz, then the evidence pass again (tools/evidence.sh r06)
export TMPDIR=/tmp
o=$PWD/gpurun_out/c28; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $o/pytest.log | cut -c1-300
timeout 600 python tools/fuzz_gpu.py 2500 7101 2>&1 | tail -1 | tee $o/fuzz.log
FUZZ_TILE=1 timeout 600 python tools/fuzz_gpu.py 500 7102 2>&1 | tail -1 | tee -a $o/fuzz.log
timeout 600 python tools/fuzz_seq.py 64 7103 2>&1 | tail -1 | tee -a $o/fuzz.log
rocm-smi --showclocks --showpower > $o/rocm_smi.txt 2>&1
bash tools/evidence.sh r06 2>&1 | tail -12
bash tools/node_host_path.sh 2>&1 | tail -3 | cut -c1-300
