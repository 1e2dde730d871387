#!/bin/bash
export TMPDIR=/tmp
for tb in 0 1; do
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tt_$tb -o t -- python tools/sweep.py ${1:-C3,C4} table=$tb --sources shared > /tmp/tt_$tb.log 2>&1
python - <<PY
import sqlite3, glob
for db in glob.glob('/tmp/tt_$tb/*.db') + glob.glob('/tmp/tt_$tb/*/*.db'):
    con = sqlite3.connect(db)
    for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 8"):
        if 'hg::' in name: print("table $tb", calls, round(avg / 1e3, 2) if avg > 1e4 else round(avg, 2), name[:75])
PY
done
