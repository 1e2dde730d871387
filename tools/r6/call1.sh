#!/bin/bash
export TMPDIR=/tmp
o=$PWD/gpurun_out/c1; rm -rf $o; mkdir -p $o
HGWARP_LIB=$PWD/homography.js_amd/lib/libhgwarp_r2.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > $o/parity_r2.log 2>&1; echo "parity r2 rc=$?"
tail -3 $o/parity_r2.log
bash tools/ab_libs.sh "cur _r2" C3,C4,G16 shared 3 phase=4,2 > $o/ab.txt 2>&1
cat $o/ab.txt
(cd /tmp && rocprofv3 -L > $o/counters.txt 2>&1)
grep -c . $o/counters.txt
for i in 1 2 3 4 5 6; do python tools/sweep.py C4 phase=-1 --sources distinct 2>&1 | grep config | cut -c1-200; done | tee $o/c4d_modes.txt
(cd /tmp && timeout 600 rocprofv3 --pmc TCC_EA0_RDREQ TCC_EA0_WRREQ -d $o/tcc -o p -- python $OLDPWD/tools/sweep.py C4 phase=-1 --sources distinct > $o/tcc.log 2>&1)
python - <<PY > $o/tcc_schema.txt 2>&1
import sqlite3, glob
for db in glob.glob('$o/tcc/**/*.db', recursive=True):
    con = sqlite3.connect(db)
    for name, sql in con.execute("select name, sql from sqlite_master"):
        print(name, '::', (sql or '')[:600].replace('\n', ' '))
    for t in [r[0] for r in con.execute("select name from sqlite_master where type='table' and name like '%pmc%'")]:
        print('----', t, con.execute(f"select count(*) from {t}").fetchone())
        for r in con.execute(f"select * from {t} limit 40"): print(r)
PY
find $o/tcc -name "*.db" -size +20M -delete
tail -5 $o/tcc.log
