#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3f; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fresh_point or fall_back or patch_kernel_limits or full_size_batches" > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -15 $o/pytest.log
timeout 600 python bench.py --no-cpu-baseline > $o/bench.json 2> $o/bench.err; echo "bench rc=$?"; tail -3 $o/bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3f/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified"], [c for c in d["checks"] if not c["ok"]])
print("roofline", {k:d["roofline"][k] for k in ("frac","kernel_ms","hbm_compulsory_frac")})
print("fresh", d["roofline_fresh"])
print("distinct", {k:d["roofline_distinct"][k] for k in ("kernel","frac","kernel_ms","ms_per_step")})
PY
timeout 600 python bench.py --no-cpu-baseline --config C5 --frames 8 > $o/bench_C5.json 2> $o/bench_C5.err; echo "bench C5 rc=$?"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r3f/bench_C5.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified"], [c for c in d["checks"] if not c["ok"]])
print("roofline", {k:d["roofline"][k] for k in ("frac","kernel_ms","hbm_compulsory_frac")})
print("fresh", d["roofline_fresh"])
print("distinct", {k:d["roofline_distinct"][k] for k in ("kernel","frac","kernel_ms","ms_per_step")})
PY
