#!/bin/bash
# r06 profiles on the GPU box -> gpurun_out/prof/ (copied to profiles/r06_* afterwards).   gpurun -- bash tools/profile_r06.sh
set -u
cd "$(dirname "$0")/.."
TAG=r06
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --no-alt --no-cpu-baseline --min-seconds 0.3"
# 1. kernel trace + stats of the DRIVER's command shape (K = 20) and of the long-region form (K = 500), 8192 episodes; numpy stream; 1024 episodes
rocprofv3 --kernel-trace --stats -d $OUT/s20 -- $BENCH --steps 20 --warmup 5 > $OUT/${TAG}_bench_under_rocprof_driver_flags.json 2> $OUT/s20.err
python tools/rocpd_summary.py stats $OUT/s20 $OUT/${TAG}_kernel_stats_8192env_driver_flags.txt > /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/s500 -- $BENCH --steps 500 --warmup 50 > $OUT/${TAG}_bench_under_rocprof_k500.json 2> $OUT/s500.err
python tools/rocpd_summary.py stats $OUT/s500 $OUT/${TAG}_kernel_stats_8192env.txt > /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/spcg -- $BENCH --rng pcg64 --steps 20 --warmup 5 > $OUT/${TAG}_bench_under_rocprof_pcg64.json 2> $OUT/spcg.err
python tools/rocpd_summary.py stats $OUT/spcg $OUT/${TAG}_kernel_stats_8192env_pcg64.txt > /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/s1024 -- $BENCH --total-envs 1024 --steps 20 --warmup 5 > $OUT/${TAG}_bench_under_rocprof_1024env.json 2> $OUT/s1024.err
python tools/rocpd_summary.py stats $OUT/s1024 $OUT/${TAG}_kernel_stats_1024env.txt > /dev/null
rm -rf $OUT/s20 $OUT/s500 $OUT/spcg $OUT/s1024
# 2. HBM counters of the timed kernels (separate --pmc passes; tools/pmc_run_kernels.sh)
bash tools/pmc_run_kernels.sh $TAG > /dev/null 2>&1
# 3. instruction mix of the persistent kernel (separate passes)
B2="python bench.py --no-alt --no-cpu-baseline --min-seconds 0.05 --steps 100 --warmup 5"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/mixA -- $B2 > /dev/null 2> $OUT/mixA.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/mixB -- $B2 > /dev/null 2> $OUT/mixB.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM -d $OUT/mixC -- $B2 > /dev/null 2> $OUT/mixC.err
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_LDS -d $OUT/mixD -- $B2 > /dev/null 2> $OUT/mixD.err
python tools/rocpd_summary.py counters k_run_philox1 $OUT/${TAG}_pmc_instruction_mix_8192env_k_run_philox1.json $OUT/mixA $OUT/mixB $OUT/mixC $OUT/mixD > /dev/null 2>&1
rm -rf $OUT/mix?
# 3b. issue-slot use of the timed kernels, instructions per phase of a step (separate --pmc passes)
bash tools/valu_busy.sh > /dev/null 2>&1
bash tools/valu_phases.sh > /dev/null 2>&1
[ -f build_var/stopfast.so ] && bash tools/valu_phases_fast.sh > /dev/null 2>&1
# 4. the launch's timeline (per-wave time stamps) at the driver's K and the full bench lines
python tools/persist_timeline.py 2>&1 | grep "cc4 timeline" > $OUT/${TAG}_persist_timeline.txt
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_flags.json 2> /dev/null
python bench.py --steps 500 --warmup 50 --no-cpu-baseline > $OUT/${TAG}_bench_default.json 2> /dev/null
for n in 16384 32768 4096 2048; do python bench.py --steps 500 --warmup 50 --no-alt --no-cpu-baseline --total-envs $n --min-seconds 1 > $OUT/${TAG}_bench_${n}env.json 2>/dev/null; done
ls -la $OUT
