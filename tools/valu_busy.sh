#!/bin/bash
# r06: which unit of the CU does the one-launch step kernel keep busy?  VALU / SALU / LDS issue cycles of the timed kernel against the cycles of its launches
# (rocprofv3's classic VALUBusy = SQ_ACTIVE_INST_VALU x 4 / SIMDs / GRBM_GUI_ACTIVE), separate --pmc passes, kernel trace only.
#   gpurun -- bash tools/valu_busy.sh        -> gpurun_out/prof/r06_pmc_valu_busy.json
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
rm -f $OUT/r06_pmc_valu_busy.json
for cfg in "8192 philox 100 k_run_philox1" "8192 philox 20 k_run_philox1" "8192 pcg64 100 k_run_pcg" "1024 philox 100 k_run_philox" "32768 philox 50 k_run_philox1"; do
  set -- $cfg; N=$1; R=$2; K=$3; KN=$4
  B="python bench.py --no-alt --no-cpu-baseline --min-seconds 0.05 --steps $K --warmup $K --total-envs $N --rng $R"
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU -d $OUT/vbA -- $B > /dev/null 2> $OUT/vbA.err
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_LDS -d $OUT/vbB -- $B > /dev/null 2> $OUT/vbB.err
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY -d $OUT/vbC -- $B > /dev/null 2> $OUT/vbC.err
  python tools/valu_busy.py $KN $N $K $OUT/r06_pmc_valu_busy.json $OUT/vbA $OUT/vbB $OUT/vbC
  rm -rf $OUT/vbA $OUT/vbB $OUT/vbC
done
cat $OUT/r06_pmc_valu_busy.json
