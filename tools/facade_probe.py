#!/usr/bin/env python
"""Single-episode wrapper surface: microseconds per BlueFlatWrapper.step (bench.py host_api_rates), and where they go:
the wrapper's Python, the C call (launch + kernel + host wait), with and without the zero-copy IO blocks / the event log."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def per_step(small_io, evlog, n=400):
    os.environ['CC4_SMALL_IO'] = '1' if small_io else '0'
    from cage_challenge_4_amd import CybORG, EnterpriseScenarioGenerator, SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent, BlueFlatWrapper
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent, red_agent_class=FiniteStateRedAgent, steps=500)
    env = BlueFlatWrapper(CybORG(sg, seed=123))
    obs, info = env.reset()
    if not evlog:
        env.env.vec.enable_event_log(False)
    rng = np.random.default_rng(123)
    valid = {a: np.nonzero(info[a]['action_mask'])[0] for a in env.agents}
    acts = [{a: int(valid[a][rng.integers(len(valid[a]))]) for a in env.possible_agents} for _ in range(n + 20)]
    for i in range(20):
        env.step(acts[i])
    t0 = time.perf_counter()
    for i in range(20, 20 + n // 2):
        env.step(acts[i])
    wrapper = (time.perf_counter() - t0) / (n // 2) * 1e6
    v = env.env.vec
    a = np.full((1, 5), -1, np.int32)
    t0 = time.perf_counter()
    for i in range(n // 2):
        v.step(a, None)
    vec = (time.perf_counter() - t0) / (n // 2) * 1e6
    import ctypes
    vp = ctypes.c_void_p
    ap = a.ctypes.data_as(vp)
    t0 = time.perf_counter()
    for i in range(40):
        v.lib.cc4_step_fetch(v._h, ap, None, *v._p_out)
    c_call = (time.perf_counter() - t0) / 40 * 1e6
    env.close()
    return {'wrapper_step_us': round(wrapper, 1), 'vec_step_us': round(vec, 1), 'cc4_step_fetch_us': round(c_call, 1)}


if __name__ == '__main__':
    out = {}
    for small_io in (1, 0):
        for evlog in (1, 0):
            out[f'small_io={small_io} event_log={evlog}'] = per_step(small_io, evlog)
    print(json.dumps(out, indent=1))
    import bench
    print(json.dumps(bench.host_api_rates(eval_eps=1), indent=1))
