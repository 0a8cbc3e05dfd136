"""Marginals of the REFERENCE's scenario generation (EnterpriseScenarioGenerator.create_scenario + State.__init__,
CybORG v4 at /root/reference) over 10 000 seeded scenarios -> tests/golden/gen_marginals_ref.json.  Data only (counts).  The
engine's counter-mode generation (env_reset_counter_mode / k_reset: per-host streams, pid collisions resolved against all
hosts at once -- a deliberate deviation from the reference's serial draw order) is gated on these counts by a two-sample
chi-square (tests/gen_marginals.py, tests/test_semantics.py).

usage: python make_gen_marginals.py [n_scenarios] [first_seed]"""
import os, sys, json
import multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests'))

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden', 'gen_marginals_ref.json')


def describe(seed):
    import ref_shim  # noqa
    from ref_dump import host_index, KIND, SUBNETS
    from CybORG import CybORG
    from CybORG.Simulator.Scenarios import EnterpriseScenarioGenerator
    from CybORG.Agents import SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent, red_agent_class=FiniteStateRedAgent, steps=500)
    env = CybORG(sg, seed=seed)
    ec = env.environment_controller
    st = ec.state
    hosts = {}
    for name, host in st.hosts.items():
        h = host_index(name)
        ip = int(str(host.interfaces[0].ip_address).split('.')[-1]) if name != 'root_internet_host_0' else int(str([i for i in host.interfaces if i.name != 'lo'][0].ip_address).split('.')[-1])
        hosts[h] = {'os': 0 if str(getattr(host.distribution, 'name', host.distribution)).upper().endswith('UBUNTU') else 1, 'ip': ip,
                    'procs': [(int(p.pid), KIND[p.name]) for p in host.processes], 'svcs': [KIND[k] for k in host.services]}
    users = [sum(1 for h in hosts if h // 17 == sn and 1 <= h % 17 <= 10 and h != 136) for sn in range(8)]
    servers = [sum(1 for h in hosts if h // 17 == sn and h % 17 >= 11 and h != 136) for sn in range(8)]
    cidr = [int(str(st.subnet_name_to_cidr[k]).split('.')[2]) for k in sorted(st.subnet_name_to_cidr, key=lambda k: SUBNETS.index(str(getattr(k, 'value', k))))]
    blue_parent = [host_index(st.sessions[f'blue_agent_{b}'][0].hostname) for b in range(5)]
    red_start = []
    for r in range(6):
        known = [k for k, v in ec.agent_interfaces[f'red_agent_{r}'].action_space.hostname.items() if v]
        assert len(known) == 1, known
        red_start.append(host_index(known[0]))
    return {'users': users, 'servers': servers, 'cidr': cidr, 'hosts': hosts, 'blue_parent': blue_parent, 'red_start': red_start}


if __name__ == '__main__':
    import gen_marginals as GM
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
    acc = GM.empty()
    with mp.Pool(min(8, os.cpu_count() or 1)) as pool:
        for i, d in enumerate(pool.imap(describe, range(first, first + n), chunksize=16)):
            GM.accumulate(acc, d)
            if i % 1000 == 999:
                print(i + 1, 'scenarios', flush=True)
    out = GM.to_json(acc)
    out['first_seed'] = first
    out['source'] = 'EnterpriseScenarioGenerator.create_scenario + State.__init__ of the reference (CybORG v4), CybORG(sg, seed=first_seed + i)'
    with open(OUT, 'w') as f:
        json.dump(out, f)
    print('wrote', OUT, {k: out[k] for k in ('scenarios', 'duplicate_pid_scenarios', 'hosts_total')})
