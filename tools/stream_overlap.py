#!/usr/bin/env python
"""Do the step kernels of the episode groups (one HIP stream each) run side by side?  Reads a rocprofv3 --kernel-trace rocpd
database and prints, per kernel name: launches, the queues / streams they ran on, sum of durations / union of their busy
intervals (1.0 = strictly one after the other, G = G launches side by side).
usage: stream_overlap.py <dir-with-db> [name-substr]"""
import glob
import os
import sqlite3
import sys


def main():
    d = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else 'k_step'
    con = sqlite3.connect(max(glob.glob(d + '/**/*.db', recursive=True), key=os.path.getmtime))
    cols = [r[1] for r in con.execute('pragma table_info(kernels)')]
    print('# kernels view columns:', ' '.join(cols))
    qcol = 'queue_id' if 'queue_id' in cols else None
    scol = 'stream_id' if 'stream_id' in cols else None
    sel = 'name, start, end' + (', ' + qcol if qcol else ', 0') + (', ' + scol if scol else ', 0')
    rows = [r for r in con.execute(f'select {sel} from kernels order by start') if sub in r[0] or 'ccl' in r[0].lower()]
    by = {}
    for n, s, e, q, st in rows:
        by.setdefault(n[:60], []).append((s, e, q, st))
    for n, v in by.items():
        tot = sum(e - s for s, e, _, _ in v)
        union, cur_s, cur_e = 0, None, None
        for s, e, _, _ in v:
            if cur_e is None or s > cur_e:
                if cur_e is not None:
                    union += cur_e - cur_s
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        union += (cur_e - cur_s) if cur_e is not None else 0
        span = v[-1][1] - v[0][0]
        print(f'{n:60s} launches {len(v):6d} avg {tot / len(v) / 1e3:8.2f} us  queues {sorted(set(x[2] for x in v))} streams {sorted(set(x[3] for x in v))}'
              f'  overlap {tot / max(union, 1):5.2f}  busy {union / max(span, 1):5.2f} of the span')


if __name__ == '__main__':
    main()
