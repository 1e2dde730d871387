#!/bin/bash
# round 6, call 16: the GPU suite twice more (flake check), smoke(), the driver's bench command as it issues it
export TMPDIR=/tmp
o=$PWD/gpurun_out/c16; rm -rf $o; mkdir -p $o
for i in 1 2; do timeout 1500 python -m pytest tests -x -q -m gpu > $o/pytest$i.log 2>&1; echo "pytest $i rc=$?"; tail -2 $o/pytest$i.log | cut -c1-200; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
( time python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/bench_driver.json 2> $o/bench_driver.err ) 2>&1 | grep real
python - <<PY
import json
d = json.loads(open('$o/bench_driver.json').read().strip().splitlines()[-1])
print(d['metric'], d['value'], d['ms_per_step'], d['verified'], [ (a['metric'], a['value'], a['verified']) for a in d.get('also', [])])
PY
