#!/bin/bash
# Round-end refresh on the GPU box: the profile set (tools/profile_round.sh) and the bench lines kept under profiles/.
# usage: bash tools/final_refresh.sh TAG   (through gpurun; copy gpurun_out/prof/TAG_* to profiles/ afterwards)
TAG=${1:-r04}
OUT=gpurun_out/prof
mkdir -p $OUT
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1
for f in stats8192 stats1024 statspcg; do cp $OUT/bench_$f.json $OUT/${TAG}_bench_under_rocprof_$f.json 2>/dev/null; done
python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/bench_default.err
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_flags.json 2> /dev/null
for N in 1024 2048 3072 4096 16384 32768; do python bench.py --no-alt --no-cpu-baseline --total-envs $N > $OUT/${TAG}_bench_${N}env.json 2> /dev/null; done
python bench.py --no-alt --no-cpu-baseline --total-envs 1024 --rng pcg64 > $OUT/${TAG}_bench_1024env_pcg64.json 2> /dev/null
python bench.py --no-alt --no-cpu-baseline --rng pcg64 > $OUT/${TAG}_bench_8192env_pcg64.json 2> /dev/null
tail -n 3 $OUT/${TAG}_bench_default.json | cut -c1-600
