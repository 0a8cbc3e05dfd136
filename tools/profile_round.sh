#!/bin/bash
# Collects the round's profiles on the GPU box into gpurun_out/prof/ (copied to profiles/rNN_* afterwards).
# usage: bash tools/profile_round.sh [tag]      (run through gpurun)
set -u
TAG=${1:-r05}
OUT=gpurun_out/prof
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python bench.py --no-alt --no-cpu-baseline --min-seconds 0.15"
# 1. kernel trace + stats of the headline command (8192 episodes) and of the 1024-episode batch
rocprofv3 --kernel-trace --stats -d $OUT/stats8192 -- $BENCH --warmup 5 > $OUT/bench_stats8192.json 2> $OUT/stats8192.err
python tools/rocpd_summary.py stats $OUT/stats8192 $OUT/${TAG}_kernel_stats_8192env.txt > /dev/null      # k_run_philox1: one launch = the 500 steps of a timed region (the 5 warm-up steps are per-step launches)
CC4_PERSIST=0 rocprofv3 --kernel-trace --stats -d $OUT/stats8192ps -- $BENCH > $OUT/bench_stats8192_per_step.json 2> $OUT/stats8192ps.err
python tools/rocpd_summary.py stats $OUT/stats8192ps $OUT/${TAG}_kernel_stats_8192env_per_step_launches.txt > /dev/null
rocprofv3 --kernel-trace --stats -d $OUT/stats1024 -- $BENCH --total-envs 1024 --warmup 1 > $OUT/bench_stats1024.json 2> $OUT/stats1024.err
python tools/rocpd_summary.py stats $OUT/stats1024 $OUT/${TAG}_kernel_stats_1024env.txt > /dev/null      # k_run_philox: one launch = the 500 steps of a timed region
CC4_MULTISTEP=0 rocprofv3 --kernel-trace --stats -d $OUT/stats1024ps -- $BENCH --total-envs 1024 > $OUT/bench_stats1024_per_step.json 2> $OUT/stats1024ps.err
python tools/rocpd_summary.py stats $OUT/stats1024ps $OUT/${TAG}_kernel_stats_1024env_per_step_launches.txt > /dev/null
# 2. HBM traffic of the one-launch kernels the timed regions run (r05: VERDICT r04 weak #4)
bash tools/pmc_run_kernels.sh $TAG > /dev/null
export CC4_PERSIST=0     # (the instruction-mix and phase passes below read the per-step kernel)
# 3. instruction mix (per-step kernel: CC4_PERSIST=0 stays) / issue utilisation at 8192 episodes (separate passes)
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/mixA -- $BENCH --min-seconds 0.05 > /dev/null 2> $OUT/mixA.err
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU -d $OUT/mixB -- $BENCH --min-seconds 0.05 > /dev/null 2> $OUT/mixB.err
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_BRANCH SQ_INSTS_SMEM -d $OUT/mixC -- $BENCH --min-seconds 0.05 > /dev/null 2> $OUT/mixC.err
python tools/rocpd_summary.py counters k_step_philox1 $OUT/${TAG}_pmc_instruction_mix_8192env.json $OUT/mixA $OUT/mixB $OUT/mixC > /dev/null
# 4. in-kernel phase cycles and launch tail
python tools/phase_profile.py 1024 300 1 > $OUT/${TAG}_phase_cycles_philox_1024env.txt 2>&1
python tools/phase_profile.py 8192 200 1 > $OUT/${TAG}_phase_cycles_philox_8192env.txt 2>&1
python tools/phase_profile.py 1024 100 0 > $OUT/${TAG}_phase_cycles_pcg64_1024env.txt 2>&1
python tools/tail_whatif.py 1024 200 > $OUT/${TAG}_tail_whatif.txt 2>&1
python tools/tail_profile.py 1024 100 1 > $OUT/${TAG}_tail_philox.txt 2>&1
unset CC4_PERSIST
# 5. numpy-stream kernel: kernel stats at 8192 episodes, what the lane-parallel green actions do per step
rocprofv3 --kernel-trace --stats -d $OUT/statspcg -- $BENCH --rng pcg64 --warmup 5 > $OUT/bench_statspcg.json 2> $OUT/statspcg.err
python tools/rocpd_summary.py stats $OUT/statspcg $OUT/${TAG}_kernel_stats_8192env_pcg64.txt > /dev/null
python tools/green_batches.py 1024 50 > $OUT/${TAG}_green_batches_pcg64_1024env.txt 2>&1
ls -la $OUT | head -50
rm -rf $OUT/stats8192 $OUT/stats8192ps $OUT/stats1024 $OUT/stats1024ps $OUT/statspcg $OUT/fetch* $OUT/write* $OUT/mix?   # the raw databases are large; the summaries stay
