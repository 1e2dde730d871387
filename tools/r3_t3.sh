#!/bin/bash
# k_tri_table ablations (experiments build): 64 = no table stores, 128 = no solve / taps, 192 = neither
export TMPDIR=/tmp
for abl in 0 64 128 192; do
  rm -rf /tmp/ta_$abl
  HG_ABLATE_TRI=$abl HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_exp.so timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/ta_$abl -o t -- python tools/sweep.py ${1:-C3} table=1 --sources shared > /tmp/ta_$abl.log 2>&1
  python - <<PY
import sqlite3, glob
for db in glob.glob('/tmp/ta_$abl/*.db') + glob.glob('/tmp/ta_$abl/*/*.db'):
    con = sqlite3.connect(db)
    for name, calls, total, avg, pct in con.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 8"):
        if 'k_tri' in name: print("abl $abl", calls, round(avg / 1e3, 2) if avg > 1e4 else round(avg, 2), name[:60])
PY
done
