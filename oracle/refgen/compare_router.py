"""Differential check of host actions on a zone's ROUTER given as Action objects (the wrappers' fixed list has no slot for
routers; CybORG.parallel_step forwards the objects, the engine takes them as BLUE_RAW_ACTION | type << 8 | host id): reference
CybORG.parallel_step vs the oracle, full state dump every step.  usage: python compare_router.py <seed> [steps]"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import ref_shim  # noqa
from compare import lib, canon_ref
from ref_dump import dump, host_index
from CybORG import CybORG
from CybORG.Simulator.Scenarios import EnterpriseScenarioGenerator
from CybORG.Agents import SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent
from CybORG.Simulator.Actions import Analyse, Remove, Restore, DeployDecoy, Sleep

seed = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
record = len(sys.argv) > 3 and sys.argv[3] == 'record'     # also write tests/golden/router_actions_seed<seed>.json (codes, rewards, dump digests)
import hashlib, json
rec = {'seed': seed, 'steps': steps, 'codes': [], 'reward': [], 'dump_sha1': []}
sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent, red_agent_class=FiniteStateRedAgent, steps=steps)
env = CybORG(sg, seed=seed)
env.reset()
H = ctypes.c_void_p(lib.cc4o_create2(1, steps))
lib.cc4o_reset(H, 0, ctypes.c_uint64(seed), 0, steps, 0, 0)
lib.cc4o_reset(H, 0, ctypes.c_uint64(seed), 0, steps, 1, 0)
rng = np.random.default_rng(seed)
ZONES = [['restricted_zone_a_subnet'], ['operational_zone_a_subnet'], ['restricted_zone_b_subnet'], ['operational_zone_b_subnet'],
         ['public_access_zone_subnet', 'admin_network_subnet', 'office_network_subnet']]
CLS = {2: Analyse, 3: Remove, 4: Restore, 5: DeployDecoy}
buf = ctypes.create_string_buffer(1 << 20)
bad = 0
for t in range(steps):
    acts, codes = {}, np.full(5, -1, np.int32)
    for b in range(5):
        if rng.random() < 0.5:
            ty = int(rng.integers(2, 6)); sn = ZONES[b][rng.integers(len(ZONES[b]))]
            host = sn + '_router'
            acts[f'blue_agent_{b}'] = CLS[ty](session=0, agent=f'blue_agent_{b}', hostname=host)
            codes[b] = 0x10000 | (ty << 8) | host_index(host)
        else:
            acts[f'blue_agent_{b}'] = Sleep()
            codes[b] = 3 * 16 * len(ZONES[b]) + 1                  # the list's Sleep slot
    obs, rew, done, info = env.parallel_step(actions=acts)
    lib.cc4o_step(H, 0, codes.ctypes.data_as(ctypes.c_void_p), None)
    n = lib.cc4o_dump(H, 0, buf, len(buf))
    mine, ref = buf.raw[:n].decode(), canon_ref(dump(env))
    r = lib.cc4o_reward(H, 0)
    rr = sum(rew['blue_agent_0'].values())
    rec['codes'].append(codes.tolist()); rec['reward'].append(float(rr)); rec['dump_sha1'].append(hashlib.sha1(ref.encode()).hexdigest())
    if mine != ref or abs(r - rr) > 1e-6 or lib.cc4o_err(H, 0):
        bad += 1
        print('step', t, 'MISMATCH reward', r, rr, 'err', lib.cc4o_err(H, 0))
        for a, b_ in zip(mine.split('\n'), ref.split('\n')):
            if a != b_:
                print('  mine:', a[:300], '\n  ref :', b_[:300])
        break
print('OK' if not bad else 'FAILED', 'seed', seed, 'steps', steps)
if record and not bad:
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden', f'router_actions_seed{seed}.json')
    json.dump(rec, open(out, 'w'))
    print('wrote', out)
