#!/bin/bash
# LDS bank conflicts of the warp kernel with span lists from either producer: tools/pmc_lds.sh CONFIG
export TMPDIR=/tmp
cfg=${1:-C3}
for band in 0 1; do
  out=$PWD/gpurun_out/pmc_lds_${cfg}_$band; rm -rf $out; mkdir -p $out
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES -d $out/p -o p -- python tools/sweep.py $cfg band=$band --sources shared > $out/log.txt 2>&1
  python - <<PY
import sqlite3, glob
for db in sorted(glob.glob('$out/p/*.db') + glob.glob('$out/p/*/*.db')):
    con = sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like '%hg::%' group by kernel_name, counter_name"):
        print("band=$band", r[0][:60], r[1], r[2], "%.4g" % r[3])
PY
  rm -rf $out/p
done
