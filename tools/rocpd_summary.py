#!/usr/bin/env python
"""Summarise rocprofv3 rocpd (.db) outputs into the small text/JSON files committed under profiles/.
usage: rocpd_summary.py stats <dir-with-db> <out.txt>
       rocpd_summary.py pmc <fetch-dir> <write-dir> <kernel-substr> <out.json> [launch_envs]
       rocpd_summary.py pmc_step <fetch-dir> <write-dir> <exact-kernel-name> <out.json> <envs> <launches-per-step>   (merges into out.json)
       rocpd_summary.py pmc_run <fetch-dir> <write-dir> <exact-kernel-name> <out.json> <envs> <steps-per-launch>     (one-launch kernels; merges)
       rocpd_summary.py counters <kernel-substr> <out.json> <dir> [<dir> ...]   (per-launch averages of every counter found)"""
import glob
import json
import sqlite3
import sys


def db(d):
    import os
    return sqlite3.connect(max(glob.glob(d + '/**/*.db', recursive=True), key=os.path.getmtime))


def stats(d, out):
    con = db(d)
    rows = list(con.execute('select name,total_calls,total_duration,average,percentage from top_kernels'))
    lines = ['# rocprofv3 --kernel-trace --stats summary (durations in us)',
             f'{"kernel":70s} {"calls":>7s} {"total_us":>14s} {"avg_us":>12s} {"pct":>7s}']
    for n, c, t, a, p in rows:
        lines.append(f'{n[:70]:70s} {c:7d} {t:14.3f} {a:12.3f} {p:7.2f}')
    k = list(con.execute("select name, vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x, min(duration), max(duration) "
                         "from kernels group by name"))
    lines.append('')
    lines.append(f'{"kernel":50s} vgpr sgpr lds scratch grid wg min_ns max_ns')
    for r in k:
        lines.append(f'{r[0][:50]:50s} ' + ' '.join(str(x) for x in r[1:]))
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


def pmc(fd, wd, kern, out, envs):
    res = {}
    for name, d in (('FETCH_SIZE', fd), ('WRITE_SIZE', wd)):
        con = db(d)
        r = list(con.execute("select count(*), avg(value), min(value), max(value) from counters_collection "
                             "where counter_name=? and kernel_name like ?", (name, f'%{kern}%')))[0]
        res[name] = {'launches': r[0], 'avg_KB': r[1], 'min_KB': r[2], 'max_KB': r[3]}
    # MI355X_MICROARCH.md (HBM): on gfx950 FETCH_SIZE reports 1/2 of the bytes of a wide (16 B/lane) coalesced read ->
    # double it; WRITE_SIZE calibrated here on hipMemset (fillBufferAligned of 34640 KB reports 34640.0 KB).
    hbm = (2.0 * res['FETCH_SIZE']['avg_KB'] + res['WRITE_SIZE']['avg_KB']) * 1024.0
    res['correction'] = 'hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (FETCH_SIZE halved on gfx950 for 16-B coalesced reads; WRITE_SIZE calibrated on a known-size memset in the same run)'
    res[f'hbm_bytes_per_launch_{envs}env'] = hbm
    res['kernel'] = kern
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


def pmc_step(fd, wd, kern, out, envs, lps):
    """r03 form: HBM bytes per STEP = launches per step x bytes per launch (a step of a large batch is several launches of the same
    kernel on separate streams; under --pmc the launches are serialised, the bytes they move are the same).  Merges into `out`."""
    import os
    res = json.load(open(out)) if os.path.exists(out) else {}
    per = {}
    kname = None
    for name, d in (('FETCH_SIZE', fd), ('WRITE_SIZE', wd)):
        con = db(d)
        r = list(con.execute("select count(*), avg(value), min(value), max(value), min(kernel_name) from counters_collection "
                             "where counter_name=? and kernel_name like ?", (name, f'%{kern}%')))[0]
        per[name] = {'launches': r[0], 'avg_KB': r[1], 'min_KB': r[2], 'max_KB': r[3]}
        kname = r[4]
    hbm = (2.0 * per['FETCH_SIZE']['avg_KB'] + per['WRITE_SIZE']['avg_KB']) * 1024.0
    res['correction'] = 'hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE is halved on gfx950 for 16-B coalesced reads; WRITE_SIZE calibrated on a known-size memset)'
    res[f'{envs}env'] = per
    res[f'hbm_bytes_per_launch_{envs}env'] = hbm
    res[f'launches_per_step_{envs}env'] = lps
    res[f'hbm_bytes_per_step_{envs}env'] = hbm * lps
    res[f'kernel_{envs}env'] = kern
    res[f'kernel_symbol_{envs}env'] = kname
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: v for k, v in res.items() if str(envs) in k}, indent=1))


def pmc_run(fd, wd, kern, out, envs, spl):
    """r05 form, the one-launch kernels profiled as what they are: one launch = `spl` steps of the whole batch, so HBM bytes per STEP =
    bytes of the launch / spl.  Only the launches of exactly `kern` count (k_run_philox1 is not k_run_philox1m).  Merges into `out`."""
    import os
    res = json.load(open(out)) if os.path.exists(out) else {}
    per = {}
    kname = None
    for name, d in (('FETCH_SIZE', fd), ('WRITE_SIZE', wd)):
        con = db(d)
        r = list(con.execute("select count(*), avg(value), min(value), max(value), min(kernel_name) from counters_collection "
                             "where counter_name=? and (kernel_name like ? or kernel_name = ?)", (name, f'%{kern}(%', kern)))[0]
        per[name] = {'launches': r[0], 'avg_KB': r[1], 'min_KB': r[2], 'max_KB': r[3]}
        kname = r[4]
    hbm = (2.0 * per['FETCH_SIZE']['avg_KB'] + per['WRITE_SIZE']['avg_KB']) * 1024.0
    res['correction'] = 'hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (MI355X_MICROARCH.md: FETCH_SIZE is halved on gfx950 for 16-B coalesced reads; WRITE_SIZE calibrated on a known-size memset)'
    res[f'{envs}env'] = per
    res[f'hbm_bytes_per_launch_{envs}env'] = hbm
    res[f'steps_per_launch_{envs}env'] = spl
    res[f'hbm_bytes_per_step_{envs}env'] = hbm / spl
    res[f'kernel_{envs}env'] = kern
    res[f'kernel_symbol_{envs}env'] = kname
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: v for k, v in res.items() if str(envs) in k}, indent=1))


def counters(kern, out, dirs):
    res = {}
    for d in dirs:
        con = db(d)
        for name, n, avg in con.execute("select counter_name, count(*), avg(value) from counters_collection "
                                        "where kernel_name like ? group by counter_name", (f'%{kern}%',)):
            res[name] = {'launches': n, 'avg_per_launch': avg}
    res['kernel'] = kern
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'pmc_run':
        pmc_run(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7]))
    elif sys.argv[1] == 'pmc_step':
        pmc_step(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7]))
    elif sys.argv[1] == 'counters':
        counters(sys.argv[2], sys.argv[3], sys.argv[4:])
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5], int(sys.argv[6]) if len(sys.argv) > 6 else 1024)
