#!/bin/bash
# usage: tools/pmc_sweep.sh "COUNTER ..." <tools/sweep.py args>: one rocprofv3 --pmc pass (its own run, no tracing) over a sweep.py run;
# prints the per-dispatch average of every counter per hg kernel
export TMPDIR=/tmp
ctrs=$1; shift
o=$PWD/gpurun_out/pmc_sweep; rm -rf $o; mkdir -p $o
(cd /tmp && timeout 600 rocprofv3 --pmc $ctrs -d $o/p -o p -- python $OLDPWD/tools/sweep.py "$@" > $o/log.txt 2>&1)
grep "config" $o/log.txt | cut -c1-200
python - <<PY
import sqlite3, glob
for db in sorted(glob.glob('$o/p/**/*.db', recursive=True)):
    con = sqlite3.connect(db)
    rows = con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%hg::%' group by kernel_name, counter_name").fetchall()
    for r in rows: print(r[0][:70].ljust(70), r[1].ljust(24), r[2], "%.5g" % r[3])
PY
rm -rf $o/p
