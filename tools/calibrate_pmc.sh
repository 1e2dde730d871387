#!/bin/bash
# usage: tools/calibrate_pmc.sh   (on the GPU box; tools/bin/calib_fetch is built by `make -C tools`)
# FETCH_SIZE and WRITE_SIZE of kernels with known byte counts, each counter in its own rocprofv3 run (never mixed with tracing).
export TMPDIR=/tmp
out=$PWD/gpurun_out/calib; mkdir -p $out
timeout 120 tools/bin/calib_fetch 2048 3 > $out/plain.jsonl 2>&1
timeout 240 rocprofv3 --pmc FETCH_SIZE -d $out/f -o c -- tools/bin/calib_fetch 2048 2 > $out/f.log 2>&1
timeout 240 rocprofv3 --pmc WRITE_SIZE -d $out/w -o c -- tools/bin/calib_fetch 2048 2 > $out/w.log 2>&1
python - <<PY
import sqlite3, glob, json
res = {}
for db in sorted(glob.glob('$out/*/*.db') + glob.glob('$out/*/*/*.db')):
    con = sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%calib%' group by kernel_name, counter_name"):
        res.setdefault(r[0].split('(')[0].replace('void ', ''), {})[r[1]] = {"dispatches": r[2], "avg": r[3]}
json.dump(res, open('$out/calib.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
cat $out/plain.jsonl
rm -rf $out/f $out/w
