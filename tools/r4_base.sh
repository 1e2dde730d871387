#!/bin/bash
# round 4, pass 0: GPU suite + default bench (step_frac lines) + C4 / C5 / distinct on the library as committed
export TMPDIR=/tmp
o=$PWD/gpurun_out/r4base; rm -rf $o; mkdir -p $o
timeout 900 python -m pytest tests -m gpu -x -q > $o/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $o/pytest.log
timeout 600 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench default rc=$?"
timeout 600 python bench.py --sources distinct --no-cpu-baseline > $o/bench_distinct.json 2> $o/bench_distinct.err; echo "bench distinct rc=$?"
for c in C4 C5; do
  timeout 600 python bench.py --no-cpu-baseline --config $c $( [ $c = C5 ] && echo --frames 8 ) > $o/bench_$c.json 2> $o/bench_$c.err; echo "bench $c rc=$?"
done
timeout 600 python bench.py --no-cpu-baseline --steps 50 --also C4:64,C5:8 --sources shared --points resident > $o/bench_also.json 2> $o/bench_also.err; echo "bench also rc=$?"
python - <<PY
import json,glob
for f in sorted(glob.glob('$o/bench_*.json')):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    def show(tag, j):
        r=j['roofline']; d=j.get('roofline_distinct') or {}; fr=j.get('roofline_fresh') or {}
        print(tag, 'value', j['value'], 'ms/step', j['ms_per_step'], 'verified', j['verified'], 'frac', r['frac'], 'step_frac', r.get('step_frac'), 'kernel_ms', r['kernel_ms'],
              '| distinct', d.get('frac'), d.get('step_frac'), d.get('kernel_ms'), '| fresh', fr.get('step_frac'), fr.get('ms_per_step'))
    show(f.split('/')[-1], j)
    for a in j.get('also', []): show('   also', a)
PY
