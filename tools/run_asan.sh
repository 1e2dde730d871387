#!/bin/bash
# Host-side code under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU needed): the C-ABI host functions (solves, limits,
# pack_offsets, partition, Delaunay incl. its fuzz, error paths without a device) through ctypes, and the N-API addon (argument
# checking, typed-array plumbing, the frame pool) through the JavaScript host tests.  python / node themselves are not
# instrumented, so the sanitizer runtime is preloaded; leak checking is off (the interpreters never free everything) and so is
# the registration of globals (report_globals=0: the module-name string of NAPI_MODULE trips a false "misaligned global").
set -e
cd "$(dirname "$0")/.."
make -C homography.js_amd asan
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1:verify_asan_link_order=0:report_globals=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
echo "== ctypes: tests/test_cabi_cpu.py under ASan/UBSan"
LD_PRELOAD=$RT HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_asan.so python -m pytest tests/test_cabi_cpu.py -x -q -p no:cacheprovider
echo "== Delaunay fuzz (degenerate, duplicate, collinear and huge inputs) under ASan/UBSan"
LD_PRELOAD=$RT HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_asan.so python tools/fuzz_delaunay.py 400
if command -v node >/dev/null && [ -f homography.js_amd/lib/hgwarp_asan.node ]; then
  echo "== N-API addon: tests/js/test_host.mjs + replay_golden --dry under ASan/UBSan"
  LD_PRELOAD=$RT HGWARP_ADDON=$PWD/homography.js_amd/lib/hgwarp_asan.node node tests/js/test_host.mjs
  LD_PRELOAD=$RT HGWARP_ADDON=$PWD/homography.js_amd/lib/hgwarp_asan.node node tests/js/replay_golden.mjs --dry
fi
echo "ASAN_UBSAN_CLEAN"
