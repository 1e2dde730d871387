#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/r3fwd2; rm -rf $o; mkdir -p $o
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "forward" > $o/pytest.log 2>&1; tail -3 $o/pytest.log
for F in 8 1; do
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $o/prof_F$F -o p -- python $GRAFT_REPO_ROOT/tools/bench_forward.py $F 30 > $o/prof_F$F.log 2>&1)
  f=$(find $o/prof_F$F -name "*kernel_stats.csv" | head -1)
  echo "== F=$F"; python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r['Name'][:60].ljust(60), r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
done
