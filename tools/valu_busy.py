#!/usr/bin/env python
"""Issue-slot use of one kernel from rocprofv3 --pmc passes (tools/valu_busy.sh): merges one entry per (kernel, episodes, steps per launch) into a JSON file.

  valu_busy = SQ_ACTIVE_INST_VALU x 4 cycles / (SIMDs x cycles of the launch)        (rocprof's VALUBusy; 1024 SIMDs = 256 CUs x 4)
  cycles of the launch: from the kernel trace (duration x the clock that SQ_BUSY_CYCLES / 32 shader engines implies) and from GRBM_GUI_ACTIVE when it is there
"""
import glob
import json
import os
import sqlite3
import sys


def db(d):
    return sqlite3.connect(max(glob.glob(d + '/**/*.db', recursive=True), key=os.path.getmtime))


def main():
    kern, envs, k, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    c = {}
    dur = []
    for d in sys.argv[5:]:
        con = db(d)
        for name, n, avg in con.execute("select counter_name, count(*), avg(value) from counters_collection where kernel_name like ? group by counter_name", (f'{kern}(%',)):
            c.setdefault(name, []).append(avg)
        r = list(con.execute("select count(*), avg(duration) from kernels where name like ?", (f'{kern}(%',)))[0]
        if r[0]:
            dur.append(r[1])
    c = {n: sum(v) / len(v) for n, v in c.items()}
    simds, ses = 1024, 32
    ent = {'kernel': kern, 'episodes': envs, 'steps_per_launch': k, 'counters_avg_per_launch': c, 'launch_ns_under_pmc': sum(dur) / len(dur) if dur else None}
    cyc = c.get('SQ_BUSY_CYCLES', 0) / ses if c.get('SQ_BUSY_CYCLES') else None
    ent['launch_cycles_sq_busy_per_se'] = cyc
    if c.get('GRBM_GUI_ACTIVE'):
        ent['grbm_gui_active'] = c['GRBM_GUI_ACTIVE']
    if cyc:
        if dur:
            ent['clock_ghz_implied'] = cyc / ent['launch_ns_under_pmc']
        for key, ctr in (('valu_busy', 'SQ_ACTIVE_INST_VALU'), ('salu_busy', 'SQ_ACTIVE_INST_SCA'), ('lds_busy', 'SQ_ACTIVE_INST_LDS')):
            if c.get(ctr):
                ent[key] = c[ctr] * 4.0 / (simds * cyc)
        if c.get('SQ_THREAD_CYCLES_VALU') and c.get('SQ_ACTIVE_INST_VALU'):
            ent['active_lanes_per_valu_instruction'] = c['SQ_THREAD_CYCLES_VALU'] / (4.0 * c['SQ_ACTIVE_INST_VALU'])   # thread-cycles / (quad-cycles x 4)
        es = envs * k
        for key, ctr in (('valu_per_episode_step', 'SQ_INSTS_VALU'), ('salu_per_episode_step', 'SQ_INSTS_SALU'), ('lds_per_episode_step', 'SQ_INSTS_LDS')):
            if c.get(ctr):
                ent[key] = c[ctr] / es
    res = json.load(open(out)) if os.path.exists(out) else {'note': 'valu_busy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles of the launch); cycles of the launch = SQ_BUSY_CYCLES / 32 shader engines'}
    res[f'{kern}_{envs}env_k{k}'] = ent
    json.dump(res, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main()
