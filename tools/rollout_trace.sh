#!/bin/bash
# kernel trace of a few rollouts: when do the sync / policy kernels of the policy stream start and end, relative to each other?
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/rtrace
rm -rf $OUT; mkdir -p $OUT
CC4_ROLLOUT_GROUPS=${1:-2} rocprofv3 --kernel-trace -d $OUT -- python tools/rollout_rate.py 20 native > $OUT/run.txt 2>&1
python - <<PY
import glob, sqlite3, os
db = max(glob.glob('$OUT/**/*.db', recursive=True), key=os.path.getmtime)
con = sqlite3.connect(db)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table' or type='view'")]
print([t for t in tabs if 'kernel' in t.lower()][:10])
rows = list(con.execute("select name, start, end from kernels order by start"))
# find the 100th k_run_philox1 launch and print the kernels that started during it
runs = [r for r in rows if 'k_run_philox1' in r[0]]
r0 = runs[100]
print('rollout kernel', (r0[2] - r0[1]) / 1e3, 'us')
inside = [r for r in rows if r0[1] <= r[1] <= r0[2] and r is not r0]
prev_end = r0[1]
for r in inside[:40]:
    print('%-28s start +%7.1f us  dur %6.1f us  gap after previous end %6.1f us' % (r[0][:28], (r[1] - r0[1]) / 1e3, (r[2] - r[1]) / 1e3, (r[1] - prev_end) / 1e3))
    prev_end = r[2]
PY
