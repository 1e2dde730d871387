#!/bin/bash
# round 6, call 12: full GPU suite on the final library, then the round's evidence pass (tools/evidence.sh r06) and the Node host path
export TMPDIR=/tmp
o=$PWD/gpurun_out/c12; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "^FAILED\|^ERROR" $o/pytest.log | head; tail -2 $o/pytest.log | cut -c1-300
rocm-smi --showclocks --showpower > $o/rocm_smi.txt 2>&1
bash tools/evidence.sh r06 2>&1 | tail -30
bash tools/node_host_path.sh 2>&1 | tail -8
