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
