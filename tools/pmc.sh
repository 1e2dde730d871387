#!/bin/bash
# usage: tools/pmc.sh <tag> : SQ-level PMC passes of a short bench run (each pass its own rocprofv3 run), results as sqlite under gpurun_out/pmc_<tag>
# NB (measured the hard way on this pool): passes with TCC_* (other than the derived FETCH_SIZE / WRITE_SIZE), TA_* or TCP_*
# counters never return -- each ran into its timeout.  Only SQ_*/GRBM_* and FETCH_SIZE / WRITE_SIZE passes are usable here.
export TMPDIR=/tmp
out=$PWD/gpurun_out/pmc_$1; mkdir -p $out
B="python bench.py --no-cpu-baseline --also none --steps 3 --warmup 1"
timeout 240 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $out/p1 -o p -- $B > $out/p1.log 2>&1
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $out/p2 -o p -- $B > $out/p2.log 2>&1
timeout 240 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM -d $out/p3 -o p -- $B > $out/p3.log 2>&1
python - <<PY
import sqlite3,glob
for db in sorted(glob.glob('$out/p*/*.db')):
    con=sqlite3.connect(db)
    for r in con.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection where kernel_name like 'hg::%' group by kernel_name, counter_name"):
        print(r[0][:24], r[1], r[2], f"{r[3]:.4g}")
PY
