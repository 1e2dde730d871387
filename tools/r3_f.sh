#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3f; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl.py -x -q -m gpu -k "geometric or device_side or rccl or torchrun or projective" > $o/pytest.log 2>&1; echo "pytest rc=$?" >> $o/pytest.log
tail -3 $o/pytest.log
for c in C2 C4; do
timeout 600 python bench.py --no-cpu-baseline --config $c > $o/bench_$c.json 2> $o/bench_$c.err; echo "bench $c rc=$?"
python - <<PY
import json
d=json.loads(open("gpurun_out/r3f/bench_$c.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified"], [c for c in d["checks"] if not c["ok"]])
print("roofline", {k:d["roofline"][k] for k in ("frac","kernel_ms","traffic")})
print("fresh", {k:d["roofline_fresh"][k] for k in ("ms_per_step","vs_resident_ms_per_step","layout_walks_in_region","frames_redone_in_region")})
print("distinct", {k:d["roofline_distinct"][k] for k in ("kernel","frac","kernel_ms","ms_per_step","traffic")})
PY
done
