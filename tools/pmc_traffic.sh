#!/bin/bash
# usage: tools/pmc_traffic.sh <tag> [bench args...]
# HBM-side traffic of the warp kernels per dispatch: FETCH_SIZE and WRITE_SIZE, each in its own rocprofv3 run (never mixed
# with tracing); KB per dispatch, averaged over the dispatches of a short bench run.  Result: gpurun_out/pmc_<tag>/traffic.json
export TMPDIR=/tmp
tag=$1; shift
out=$PWD/gpurun_out/pmc_$tag; mkdir -p $out
B="python bench.py --no-cpu-baseline --no-verify --steps 20 --warmup 2 $*"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $out/f -o p -- $B > $out/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $out/w -o p -- $B > $out/w.log 2>&1
python - <<PY
import sqlite3, glob, json
res = {}
for db in sorted(glob.glob('$out/*/*.db') + glob.glob('$out/*/*/*.db')):
    con = sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection group by kernel_name, counter_name"):
        if 'hg::' in r[0] or 'calib' in r[0]:
            res.setdefault(r[0].split('(')[0].replace('void ', ''), {})[r[1]] = {"dispatches": r[2], "avg_KB": r[3], "min_KB": r[4], "max_KB": r[5]}
json.dump({"bench_args": "$*", "kernels": res}, open('$out/traffic.json', 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/f $out/w     # the sqlite outputs are large; gpurun only copies back 64 MiB
