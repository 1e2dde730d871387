#!/bin/bash
# The C ABI's threading contract under ThreadSanitizer (GPU box): tests/test_gpu_threads.py -- two hg_ctx on device 0 driven from two
# host threads -- through the host-instrumented library (make tsan).  python and the ROCm runtime are not instrumented: the runtime is
# preloaded, their internals suppressed (tools/tsan.supp); a report therefore names a race inside libhgwarp's own host code.
cd "$(dirname "$0")/.."
make -C homography.js_amd tsan || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.tsan-x86_64.so)
export TSAN_OPTIONS="suppressions=$PWD/tools/tsan.supp:halt_on_error=0:report_signal_unsafe=0:second_deadlock_stack=1:exitcode=66:history_size=4:ignore_noninstrumented_modules=1"
LD_PRELOAD=$RT HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_tsan.so HG_THREAD_ITERS=${HG_THREAD_ITERS:-60} timeout ${TSAN_TIMEOUT:-900} \
  python -m pytest tests/test_gpu_threads.py -x -q -p no:cacheprovider
rc=$?
[ $rc = 0 ] && echo "TSAN_CLEAN"
exit $rc
