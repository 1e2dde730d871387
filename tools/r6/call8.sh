#!/bin/bash
# round 6, call 8: full GPU suite after the tile-kernel changes and the shared-source policy (no -x: list every kernel-selection expectation that moved), default bench
export TMPDIR=/tmp
o=$PWD/gpurun_out/c8; rm -rf $o; mkdir -p $o
timeout 1500 python -m pytest tests -q -m gpu > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -30 $o/pytest.log | cut -c1-300
( time timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads(open('$o/bench_default.json').read().strip().splitlines()[-1])
def show(x):
    r = x['roofline']; rd = x.get('roofline_distinct') or {}
    print(x['metric'], x['value'], x['ms_per_step'], 'verified', x['verified'], r['kernel'], 'frac', r['frac'], 'step_frac', r['step_frac'], '| distinct', rd.get('kernel'), rd.get('frac'))
show(d)
for a in d.get('also', []): show(a)
PY
