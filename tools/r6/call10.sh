#!/bin/bash
# round 6, call 10: the pruned library (no k_pw_rows8, one depth for the self-span and patch kernels): full GPU suite, fuzz, census, A/B against the build before, TSan
export TMPDIR=/tmp
o=$PWD/gpurun_out/c10; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $o/pytest.log | cut -c1-300
timeout 600 python tools/fuzz_gpu.py 1500 991 2>&1 | tail -2 | tee $o/fuzz.log
FUZZ_TILE=1 timeout 600 python tools/fuzz_gpu.py 300 992 2>&1 | tail -2 | tee -a $o/fuzz.log
timeout 600 python tools/fuzz_seq.py 48 993 2>&1 | tail -2 | tee $o/fuzz_seq.log
timeout 900 python tools/census.py > $o/census.txt 2> $o/census.err; echo "census rc=$?"; cut -c1-200 $o/census.txt
TSAN_TIMEOUT=600 bash tools/run_tsan.sh > $o/tsan.log 2>&1; echo "tsan rc=$?"; grep -c "WARNING: ThreadSanitizer" $o/tsan.log; tail -5 $o/tsan.log | cut -c1-300
