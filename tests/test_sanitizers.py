"""Host-side code under AddressSanitizer + UndefinedBehaviorSanitizer: builds `make asan` and runs the C-ABI
host tests, a Delaunay fuzz and the JavaScript host tests through the instrumented library / addon (tools/run_asan.sh)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang"


@pytest.mark.slow
@pytest.mark.skipif(not os.path.exists(CLANG) or shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"),
                    reason="ROCm clang / hipcc not available")
def test_host_code_is_clean_under_asan_ubsan():
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "run_asan.sh")], capture_output=True, text=True, timeout=1200, cwd=ROOT)
    tail = (p.stdout + p.stderr)[-3000:]
    assert p.returncode == 0 and "ASAN_UBSAN_CLEAN" in p.stdout, tail
    assert "runtime error" not in p.stdout + p.stderr and "AddressSanitizer" not in p.stdout + p.stderr, tail
