#!/usr/bin/env python
"""Vector / scalar / LDS instructions of each PHASE of a counter-mode step (k_step_philox1), from the hardware's instruction counters.

The step kernels are bound by VALU issue slots (profiles/r06_pmc_valu_busy.json), so what a phase costs is the number of instructions it issues.  The full
build of k_step_philox1 can end a step behind any phase (cc4_debug_stop_phase) without writing the row back; this script snapshots a batch, runs the SAME
step from the SAME states once per stop, restores in between, and writes the order of the stops; under rocprofv3 --pmc every such launch gets its own
counter values, and the differences between consecutive stops are the phases:

  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d DIR -- python tools/valu_phases.py run SEQ.json
  python tools/valu_phases.py report SEQ.json DIR [DIR ...] > profiles/r06_valu_phases.txt
"""
import glob
import json
import os
import sqlite3
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STOPS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14]
NAMES = {1: 'stage in, work area, step_phase', 2: 'block bank (one Philox pass, 54 lanes)', 3: 'red policies + queue ticks (lanes 0-5)',
         4: 'blue submissions + ticks + messages (lanes 8-12)', 5: 'green policies (a lane per agent)', 6: 'blue actions', 7: 'green actions (compacted list)',
         8: 'phishing, slot reservation, conflict mask (lane 0)', 9: 'red actions', 10: 'merge + reassignment (lane 0)', 11: 'Monitor roll-over (all lanes)',
         12: 'RedSessionCheck + Monitor hand-over + end of step', 13: 'observation encode', 14: 'stage out, error word'}


def run(seq_path):
    os.environ.setdefault('CC4_PHILOX_LEAN', '1')
    for k in ('CC4_PERSIST', 'CC4_MULTISTEP', 'CC4_RUN1'):
        os.environ[k] = '0'
    from cage_challenge_4_amd import CC4VecEnv
    n = int(os.environ.get('VP_ENVS', '1024'))
    env = CC4VecEnv(n, steps=500, rng_mode=1, autoreset=True)
    env.reset(seeds=1000)
    assert env.step_kernel == 'k_step_philox1', env.step_kernel
    t = 0
    seq = []
    for sample in range(int(os.environ.get('VP_SAMPLES', '4'))):
        env.run_random_steps(1000, t, 70, timed=False)
        t += 70
        snaps = [env.snapshot(i) for i in range(n)]
        for stop in STOPS:
            for i in range(n):
                env.restore(i, snaps[i])
            env._chk(env.lib.cc4_debug_stop_phase(env._h, stop), 'cc4_debug_stop_phase')
            env.run_random_steps(1000, t, 1, timed=False)
            env.synchronize()
            seq.append(stop)
        env._chk(env.lib.cc4_debug_stop_phase(env._h, 0), 'cc4_debug_stop_phase')
        t += 1    # (the last stop was 14 = a whole step: the batch is one step further, consistently)
    json.dump({'episodes': n, 'sequence': seq, 'launches_per_step': env.launches_per_step, 'warm_steps_per_sample': 70,
               'fast_build': bool(os.environ.get('CC4_DEBUG_STOP_FAST'))}, open(seq_path, 'w'))
    env.close()


def report(seq_path, dirs):
    seq = json.load(open(seq_path))
    n, order, lps = seq['episodes'], seq['sequence'], int(seq['launches_per_step'])
    ctr = {}
    for d in dirs:
        con = sqlite3.connect(max(glob.glob(d + '/**/*.db', recursive=True), key=os.path.getmtime))
        rows = list(con.execute("select counter_name, dispatch_id, value from counters_collection where kernel_name like ? order by dispatch_id", ('%k_step_philox1<' + ('false' if os.environ.get('CC4_DEBUG_STOP_FAST') else 'true') + '>%',)))
        for name in sorted({r[0] for r in rows}):
            vals = [r[2] for r in rows if r[0] == name]
            if seq.get('fast_build'):       # the warm-up steps between the samples are launches of the same (fast) kernel: drop them
                per = (seq['warm_steps_per_sample'] + len(STOPS)) * lps
                assert len(vals) == per * (len(order) // len(STOPS)), (name, len(vals), per)
                vals = [v for i, v in enumerate(vals) if i % per >= seq['warm_steps_per_sample'] * lps]
            assert len(vals) == len(order) * lps, (name, len(vals), len(order), lps)
            ctr[name] = [sum(vals[i * lps:(i + 1) * lps]) for i in range(len(order))]
    samples = len(order) // len(STOPS)
    print(f'# r06: instructions of each phase of a counter-mode step (k_step_philox1, {"FAST build (a library built with -DCC4_STOP_IN_FAST=1)" if seq.get("fast_build") else "full build"}), {n} episodes, {samples} sample steps 70 steps apart;')
    print('# per episode-step.  Method: tools/valu_phases.py (the same step from the same states, ended behind each phase in turn; hardware counters per launch).')
    print('# Episodes that regenerate at the sampled step run the generation instead of the phases: it is in every stop alike and cancels in the differences.')
    names = [c for c in ('SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU', 'SQ_INSTS_VMEM_RD', 'SQ_INSTS_VMEM_WR', 'SQ_INSTS_BRANCH') if c in ctr]
    print(f'{"phase":58s} ' + ' '.join(f'{c[3:]:>16s}' for c in names) + '   VALU %')
    tot_valu = sum(ctr['SQ_INSTS_VALU'][s * len(STOPS) + len(STOPS) - 1] for s in range(samples)) / samples / n
    prev = {c: 0.0 for c in names}
    for j, stop in enumerate(STOPS):
        cur = {c: sum(ctr[c][s * len(STOPS) + j] for s in range(samples)) / samples / n for c in names}
        print(f'{stop:2d} {NAMES[stop]:55s} ' + ' '.join(f'{cur[c] - prev[c]:16.1f}' for c in names) + f'   {100.0 * (cur["SQ_INSTS_VALU"] - prev["SQ_INSTS_VALU"]) / tot_valu:5.1f}')
        prev = cur
    print(f'{"   whole step":58s} ' + ' '.join(f'{prev[c]:16.1f}' for c in names))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(sys.argv[2])
    else:
        report(sys.argv[2], sys.argv[3:])
