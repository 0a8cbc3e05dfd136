#!/bin/bash
# r06: PC sampling of the persistent step kernel (where do its VALU issue slots go?).  build_var/dbg.so = the product sources with -gline-tables-only
# (make EXTRA=-gline-tables-only: same code, line tables for the symbolizer).   gpurun -- bash tools/ab/r06_pc_sampling.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/pcs
export TMPDIR=/tmp
export CC4_LIB=$PWD/build_var/dbg.so
M=${1:-host_trap}
timeout 600 rocprofv3 --kernel-trace --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method $M --pc-sampling-interval ${2:-1} --output-format csv -d gpurun_out/pcs/run -- \
  python bench.py --steps 100 --warmup 5 --no-alt --no-cpu-baseline --min-seconds 0.5 > gpurun_out/pcs/bench.out 2> gpurun_out/pcs/bench.err
echo "rc $?"
tail -5 gpurun_out/pcs/bench.err
find gpurun_out/pcs -type f | head -20
for f in $(find gpurun_out/pcs -name "*pc_sampling*.csv"); do wc -l $f; head -5 $f; done
