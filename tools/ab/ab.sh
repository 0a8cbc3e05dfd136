#!/bin/bash
# A/B harness for kernel experiments: builds variants of libcc4.so from the working tree with extra -D flags into build_var/
# (travels with gpurun), and benches them back to back on one box (run-to-run spread on one box is ~0.1 %, between boxes ~3 %).
#   bash tools/ab/ab.sh build NAME [-DFLAG ...]      (in the build container: make -j over the translation units, objects under build_var/NAME.obj)
#   bash tools/ab/ab.sh bench NAME [NAME ...]        (through gpurun)
set -u
cd "$(dirname "$0")/../.."
mkdir -p build_var
if [ "$1" = build ]; then
  name=$2; shift 2
  make -C cage_challenge_4_amd/csrc -j8 BUILD=$PWD/build_var/$name.obj OUT=$PWD/build_var/$name.so EXTRA="$*" > build_var/$name.log 2>&1 || { tail -20 build_var/$name.log; exit 1; }
  find build_var/$name.obj -mindepth 2 -type f -delete      # the -save-temps intermediates
else
  shift
  for round in 1 2; do for v in "$@"; do for n in 8192 1024; do
    CC4_LIB=$PWD/build_var/$v.so python bench.py --no-alt --no-cpu-baseline --total-envs $n --min-seconds 0.4 --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', $n, round(d['value']/1e6,1), 'M  step_ms', round(d['roofline']['step_ms'],4), d['roofline']['kernel'], 'err', d['config']['engine_error_flags'])
"
  done; done; done
fi
