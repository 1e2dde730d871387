#!/bin/bash
o=$PWD/gpurun_out/r2b; mkdir -p $o
tools/pmc_traffic.sh C3_distinct --sources distinct > $o/pmc_C3d.log 2>&1
tools/pmc_traffic.sh C3_shared --sources shared > $o/pmc_C3s.log 2>&1
tools/pmc_traffic.sh C5_distinct --config C5 --frames 8 --sources distinct > $o/pmc_C5d.log 2>&1
tools/pmc_traffic.sh C5_shared --config C5 --frames 8 --sources shared > $o/pmc_C5s.log 2>&1
tools/pmc_traffic.sh C4_distinct --config C4 --sources distinct > $o/pmc_C4d.log 2>&1
tools/pmc_traffic.sh C2_distinct --config C2 --sources distinct > $o/pmc_C2d.log 2>&1
timeout 300 tools/calibrate_pmc.sh > $o/calib.log 2>&1
timeout 300 python bench.py --config C5 --frames 8 --no-cpu-baseline > $o/bench_C5.json 2>$o/bench_C5.err
tail -c 600 $o/bench_C5.json
