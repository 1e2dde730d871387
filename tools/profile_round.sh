#!/bin/bash
# usage: tools/profile_round.sh <tag> [bench args]   (run on the GPU box through gpurun)
# 1. rocprofv3 --kernel-trace --stats of the default bench command
# 2. PMC passes, each in its own rocprofv3 run (never combined with tracing), each under its own `timeout`
# 3. tools/profile_summary.py turns the sqlite outputs into small text summaries under gpurun_out/profile_<tag>/
set -u
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/profile_$tag
rm -rf $out; mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python bench.py --no-cpu-baseline "$@" > $out/bench_under_trace.log 2>&1
P="--no-cpu-baseline --steps 2 --warmup 1"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $out/pmc_sq1 -o p -- python bench.py $P "$@" > $out/pmc_sq1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $out/pmc_sq2 -o p -- python bench.py $P "$@" > $out/pmc_sq2.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $out/pmc_fetch -o p -- python bench.py $P "$@" > $out/pmc_fetch.log 2>&1; echo "fetch rc=$?" >> $out/pmc_fetch.log
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $out/pmc_write -o p -- python bench.py $P "$@" > $out/pmc_write.log 2>&1; echo "write rc=$?" >> $out/pmc_write.log
python tools/profile_summary.py $out > $out/summary.txt 2>&1
find $out -name "*.db" -delete
tail -n 60 $out/summary.txt
