#!/bin/bash
# r06: instructions per phase of a counter-mode step (tools/valu_phases.py), two --pmc passes.   gpurun -- bash tools/valu_phases.sh
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf $OUT/vpA $OUT/vpB
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $OUT/vpA -- python tools/valu_phases.py run $OUT/vp_seq.json > $OUT/vpA.out 2> $OUT/vpA.err
echo "pass A rc $?"; tail -3 $OUT/vpA.err
timeout 900 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH -d $OUT/vpB -- python tools/valu_phases.py run $OUT/vp_seq.json > $OUT/vpB.out 2> $OUT/vpB.err
echo "pass B rc $?"
python tools/valu_phases.py report $OUT/vp_seq.json $OUT/vpA $OUT/vpB | tee $OUT/r06_valu_phases.txt
rm -rf $OUT/vpA $OUT/vpB
