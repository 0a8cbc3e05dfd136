"""Loader for tests/golden/traj_*.npz (written by oracle/refgen/make_golden.py from the real reference)."""
import glob
import os
import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def list_fixtures():
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, 'traj_*.npz')))


def load(path):
    z = np.load(path)
    obs = np.unpackbits(z['obs_bits'], axis=1)[:, :578].astype(np.int32)
    cols = z['phase_cols']
    obs[:, cols] = z['obs_phase_vals'].astype(np.int32)
    msgs = z['messages']
    return {
        'name': os.path.basename(path), 'seed': int(z['seed']), 'reset_seed': int(z['reset_seed']), 'steps': int(z['steps']),
        'actions': z['actions'].astype(np.int32), 'obs': obs, 'reward': z['reward'], 'done': z['done'].astype(bool),
        'rng': z['rng'], 'mask': z['mask'].astype(bool), 'messages': msgs if msgs.size else None,
        'red_policy': int(z['red_policy']) if 'red_policy' in z else 0,
        'green_policy': int(z['green_policy']) if 'green_policy' in z else 0,
        'blue_policy': int(z['blue_policy']) if 'blue_policy' in z else 0,
    }


def rng_words_match(fix_rng_row, state_row):
    """fixture row: [s_hi, s_lo, has32, uinteger]; state row: [s_hi s_lo inc_hi inc_lo has32 u32 ndraw]."""
    ok = int(fix_rng_row[0]) == int(state_row[0]) and int(fix_rng_row[1]) == int(state_row[1]) and int(fix_rng_row[2]) == int(state_row[4])
    if ok and int(fix_rng_row[2]):
        ok = int(fix_rng_row[3]) == int(state_row[5])
    return ok


def list_ctr_fixtures():
    """tests/golden/ctrstep_*.npz: counter (Philox) mode trajectories recorded from the real reference running under
    oracle/refgen/philox_proxy.PhiloxProxy (oracle/refgen/make_ctr_golden.py)."""
    return sorted(glob.glob(os.path.join(GOLDEN_DIR, 'ctrstep_*.npz')))


def load_ctr(path):
    z = np.load(path)
    obs = np.unpackbits(z['obs_bits'], axis=1)[:, :578].astype(np.int32)
    obs[:, z['phase_cols']] = z['obs_phase_vals'].astype(np.int32)
    msgs = z['messages']
    return {'name': os.path.basename(path), 'seed': int(z['seed']), 'key': int(z['key']), 'steps': int(z['steps']),
            'actions': z['actions'].astype(np.int32), 'obs': obs, 'reward': z['reward'], 'done': z['done'].astype(bool),
            'mask': z['mask'].astype(bool), 'messages': msgs if msgs.size else None,
            'red_policy': int(z['red_policy']), 'green_policy': int(z['green_policy']), 'blue_policy': int(z['blue_policy'])}


def ctr_start(env_cls, fixes, **kw):
    """The episodes of counter-mode fixtures at step 0, on backend `env_cls` (CC4VecEnv or OracleVecEnv): the scenario comes from
    the numpy stream of the fixture's seed (reset(seed) then reset(None), as CybORG(seed) + wrapper.reset() do) on a numpy-stream
    handle, is handed over to a counter-mode handle as a snapshot (cc4_get/set_state + cc4_get/set_cold), and cc4_set_seed puts
    the dynamics on the counter streams of the fixture's key.  Fixtures may differ in policies (one numpy-stream handle each: the
    policy bits live in the row) and share the counter-mode handle; they must agree in episode length (a handle's cold
    containers are sized from it).  Returns (env, first observations, masks)."""
    n = len(fixes)
    assert len({f['steps'] for f in fixes}) == 1
    ctr = env_cls(n, steps=fixes[0]['steps'], rng_mode=1, **kw)
    ctr.reset(seeds=1)
    obs0, masks = [], []
    for i, f in enumerate(fixes):
        g = env_cls(1, steps=f['steps'], rng_mode=0, red_policy=f['red_policy'], green_policy=f['green_policy'], blue_policy=f['blue_policy'])
        g.reset(seeds=np.array([f['seed']], np.uint64))
        obs0.append(g.reset(seeds=None)[0].copy())
        masks.append(g.action_mask[0].copy())
        ctr.restore(i, g.snapshot(0))
        g.close()
    ctr.set_seed(np.array([f['key'] for f in fixes], np.uint64))
    return ctr, np.stack(obs0), np.stack(masks)
