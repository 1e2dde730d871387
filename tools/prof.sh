#!/bin/bash
# usage: tools/prof.sh <tag> [bench args...]
# rocprofv3 kernel trace + stats of bench.py, then PMC passes (each in its own run, --pmc never mixed with tracing),
# all under gpurun_out/prof_<tag>/ (scratch; copy what should be judged into profiles/).
set -u
tag=$1; shift
export TMPDIR=/tmp
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o t -- python bench.py --no-cpu-baseline --also none "$@" > $out/bench_trace.log 2>&1
P="--steps 3 --warmup 1"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $out/pmc1 -o p -- python bench.py --no-cpu-baseline $P > $out/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $out/pmc2 -o p -- python bench.py --no-cpu-baseline $P > $out/bench_pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum -d $out/pmc3 -o p -- python bench.py --no-cpu-baseline $P > $out/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum -d $out/pmc4 -o p -- python bench.py --no-cpu-baseline $P > $out/bench_pmc4.log 2>&1
find $out -name "*.csv" | head -30
