"""Marginal statistics of a generated CC4 scenario -- the same extraction for the reference's EnterpriseScenarioGenerator
(oracle/refgen/make_gen_marginals.py builds the description from the live reference objects) and for the engine (from the
true-state document, cc4_get_true_state / cc4o_true_state) -- and the two-sample chi-square gate between them.

A scenario description is a dict:
  users[8], servers[8]           host counts of the eight populated subnets (SUBNET enum order)
  cidr[9]                        third octet of every subnet's /24
  hosts: {host id: {'os': 0 UBUNTU / 1 KALI, 'ip': last octet, 'procs': [(pid, kind)...] (Host.processes order),
                    'svcs': [kind...] (Host.services order)}}
  blue_parent[5], red_start[6]   host ids
Host id = subnet * 17 + slot (0 router, 1..10 user hosts, 11..16 server hosts), 136 = internet root; kinds: 0 sshd, 1 OT service,
2 apache, 3 mysql, 4 smtp, 9 / 10 / 11 blue / green / red session process (csrc/cc4_state.h K_*)."""
import itertools
import json
import numpy as np

ADDON_TUPLES = [()] + [p for k in (1, 2, 3) for p in itertools.permutations((2, 3, 4), k)]     # ordered add-on service kinds: 16 outcomes
ADDON_INDEX = {t: i for i, t in enumerate(ADDON_TUPLES)}
RED_SUBNETS = [[4], [0], [1], [2], [3], [5, 6, 7]]          # allowed_subnets order of red_agent_r (EnterpriseScenarioGenerator.py:769-776)
BLOCKS = ('users', 'servers', 'addons', 'addons_ot', 'os', 'pid_bin', 'sess_offset', 'blue_parent_kind', 'red_subnet', 'red_kind', 'cidr_bin', 'ip_bin')


def empty():
    return {'users': np.zeros((8, 8), np.int64), 'servers': np.zeros((8, 6), np.int64), 'addons': np.zeros(16, np.int64),
            'addons_ot': np.zeros(16, np.int64), 'os': np.zeros(2, np.int64), 'pid_bin': np.zeros(9, np.int64),
            'sess_offset': np.zeros(9, np.int64), 'blue_parent_kind': np.zeros((5, 3), np.int64), 'red_subnet': np.zeros(3, np.int64),
            'red_kind': np.zeros((6, 2), np.int64), 'cidr_bin': np.zeros(8, np.int64), 'ip_bin': np.zeros(8, np.int64),
            'scenarios': 0, 'duplicate_pid_scenarios': 0, 'hosts_total': 0}


def kind_of_slot(h):
    slot = h % 17 if h != 136 else 0
    return 0 if slot == 0 else (1 if slot <= 10 else 2)


def accumulate(acc, d):
    acc['scenarios'] += 1
    for sn in range(8):
        acc['users'][sn, d['users'][sn] - 3] += 1
        acc['servers'][sn, d['servers'][sn] - 1] += 1
    for c in d['cidr']:
        acc['cidr_bin'][c // 32] += 1
    svc_pids = []
    for h, ho in d['hosts'].items():
        h = int(h)
        acc['hosts_total'] += 1
        acc['os'][ho['os']] += 1
        if kind_of_slot(h) != 2:                                    # routers and user hosts draw their address; servers take the top ones
            acc['ip_bin'][ho['ip'] // 32] += 1
        if kind_of_slot(h) != 0:
            sv = list(ho['svcs'])
            assert sv[0] == 0, sv
            ot = len(sv) > 1 and sv[1] == 1
            assert ot == ((h // 17) in (1, 3)), (h, sv)              # the OT service runs in the two operational zones
            acc['addons_ot' if ot else 'addons'][ADDON_INDEX[tuple(sv[2 if ot else 1:])]] += 1
        mx = 0
        for pid, kind in ho['procs']:
            if kind <= 4:
                svc_pids.append(pid)
                acc['pid_bin'][pid // 1000 - 1] += 1
            elif kind in (9, 10, 11):
                acc['sess_offset'][pid - mx - 1] += 1                # Host.create_pid: max(pids) + integers(1, 10)
            mx = max(mx, pid)
    if len(set(svc_pids)) != len(svc_pids):
        acc['duplicate_pid_scenarios'] += 1
    for b in range(5):
        acc['blue_parent_kind'][b, kind_of_slot(d['blue_parent'][b])] += 1
    for r in range(6):
        h = d['red_start'][r]
        if r == 5:
            acc['red_subnet'][RED_SUBNETS[5].index(h // 17)] += 1
        else:
            assert h // 17 == RED_SUBNETS[r][0]
        acc['red_kind'][r, kind_of_slot(h) - 1] += 1


def desc_from_true_state(doc):
    st = json.loads(doc) if isinstance(doc, (str, bytes)) else doc
    hosts = {h['h']: {'os': h['os'], 'ip': h['ip'], 'procs': [(p[0], p[1]) for p in h['procs']], 'svcs': [s[0] for s in h['svcs']]} for h in st['hosts']}
    users = [sum(1 for h in hosts if h // 17 == sn and 1 <= h % 17 <= 10 and h != 136) for sn in range(8)]
    servers = [sum(1 for h in hosts if h // 17 == sn and h % 17 >= 11 and h != 136) for sn in range(8)]
    return {'users': users, 'servers': servers, 'cidr': st['cidr'], 'hosts': hosts, 'blue_parent': [b['parent'] for b in st['blue']],
            'red_start': [r['start'] for r in st['red']]}


def to_json(acc):
    return {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in acc.items()}


def from_json(d):
    return {k: (np.array(v, np.int64) if isinstance(v, list) else v) for k, v in d.items()}


def chi2_two_sample(a, b):
    """Pearson chi-square of homogeneity for two count vectors (cells empty in both dropped): (statistic, dof, p-value)."""
    from scipy.stats import chi2
    a = np.asarray(a, np.float64).ravel(); b = np.asarray(b, np.float64).ravel()
    keep = (a + b) > 0
    a, b = a[keep], b[keep]
    na, nb = a.sum(), b.sum()
    ea = (a + b) * na / (na + nb); eb = (a + b) * nb / (na + nb)
    stat = float(((a - ea) ** 2 / ea).sum() + ((b - eb) ** 2 / eb).sum())
    dof = max(len(a) - 1, 1)
    return stat, dof, float(chi2.sf(stat, dof))


def compare(ref, eng):
    """{block: (chi-square, dof, p)} for every marginal; 2-D blocks are tested row by row (one distribution per subnet / agent)."""
    out = {}
    for k in BLOCKS:
        r, e = np.asarray(ref[k]), np.asarray(eng[k])
        if r.ndim == 2:
            for i in range(r.shape[0]):
                if (r[i] + e[i]).sum() > 0 and ((r[i] + e[i]) > 0).sum() > 1:
                    out[f'{k}[{i}]'] = chi2_two_sample(r[i], e[i])
        else:
            out[k] = chi2_two_sample(r, e)
    return out
