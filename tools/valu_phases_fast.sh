#!/bin/bash
# r06: instructions per phase of a step in the FAST build of k_step_philox1 (build_var/stopfast.so: make EXTRA=-DCC4_STOP_IN_FAST=1).   gpurun -- bash tools/valu_phases_fast.sh
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp CC4_LIB=$PWD/build_var/stopfast.so CC4_DEBUG_STOP_FAST=1
rm -rf $OUT/vpA $OUT/vpB
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/vpA -- python tools/valu_phases.py run $OUT/vp_seq.json > $OUT/vpA.out 2> $OUT/vpA.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH -d $OUT/vpB -- python tools/valu_phases.py run $OUT/vp_seq.json > $OUT/vpB.out 2> $OUT/vpB.err
python tools/valu_phases.py report $OUT/vp_seq.json $OUT/vpA $OUT/vpB | tee $OUT/r06_valu_phases_fast.txt
rm -rf $OUT/vpA $OUT/vpB
