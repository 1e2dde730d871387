#!/bin/bash
export TMPDIR=/tmp
for cfg in ${1:-C3 C4 C5}; do
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tt_$cfg -o t -- python tools/sweep.py $cfg phase=-1 --sources shared > /tmp/tt_$cfg.log 2>&1
python - <<PY
import sqlite3, glob
for db in glob.glob('/tmp/tt_$cfg/*.db') + glob.glob('/tmp/tt_$cfg/*/*.db'):
    con = sqlite3.connect(db)
    for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 8"):
        if 'hg::' in name: print("$cfg", calls, round(avg / 1e3, 2) if avg > 1e4 else round(avg, 2), name[:75])
PY
done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or oracle or fall_back or quirk or fresh or batches" 2>&1 | tail -2
