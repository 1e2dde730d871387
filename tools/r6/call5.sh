#!/bin/bash
# round 6, call 5: is k_pw_tile bound by vector issue?  (a) 8 / 16 extra dependent fp64 fma per pixel, (b) SQ counters of the kernel
export TMPDIR=/tmp
o=$PWD/gpurun_out/c5; rm -rf $o; mkdir -p $o
bash tools/ab_libs.sh "cur _pad8 _pad16" C4,C5,C3 distinct 2 tile=1 > $o/ab_pad.txt 2>&1; cat $o/ab_pad.txt
for c in C3 C4 C5; do
  bash tools/pmc_sweep.sh "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" $c tile=1 --sources distinct 2>&1 | grep -v "k_tri\|k_upload" | tee -a $o/pmc1.txt
  bash tools/pmc_sweep.sh "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" $c tile=1 --sources distinct 2>&1 | grep -v "k_tri\|k_upload" | tee -a $o/pmc2.txt
done
