#!/bin/bash
# k_tri_spans ablations on C5 (experiments build): per-kernel averages from rocprofv3 --kernel-trace --stats
export TMPDIR=/tmp
for abl in 0 32 64 96; do
  out=/tmp/tri_$abl; rm -rf $out
  HG_ABLATE_TRI=$abl timeout 200 rocprofv3 --kernel-trace --stats -d $out -o t -- python tools/ablate.py ${1:-C5} shared --one ${2:-8} > /tmp/tri_$abl.log 2>&1
  python - <<PY
import sqlite3, glob
for db in glob.glob('$out/*.db') + glob.glob('$out/*/*.db'):
    con = sqlite3.connect(db)
    for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 3"):
        print("abl $abl", calls, round(avg / 1e3, 2) if avg > 1e4 else round(avg, 2), name[:60])
PY
done
