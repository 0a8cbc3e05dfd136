"""Seeded random externally submitted actions for the parity tests of cc4_step_ex (TEST INFRASTRUCTURE).

The records are built from an episode's true-state document (cc4_get_true_state / cc4o_true_state): session ids the agent
really holds, addresses / hostnames / subnets its ActionSpace knows -- and, now and then, ones it does not (-> InvalidAction),
with duration and rate overrides mixed in.  The same arrays go to the HIP path (CC4VecEnv.step_ex) and to the oracle
(OracleVecEnv.step_ex)."""
import json
import numpy as np

NRED, MAXG = 6, 80


def _bits(words):
    return [32 * w + b for w, v in enumerate(words) for b in range(32) if (v >> b) & 1]


def random_ext(docs, rng, p_red=0.4, p_green=0.1, red_arr=None, green_arr=None):
    """docs: one true-state JSON text per episode.  Fills (and returns) the [N, 6] red and [N, 80] green record arrays."""
    for e, doc in enumerate(docs):
        d = json.loads(doc) if isinstance(doc, str) else doc
        hosts = [h['h'] for h in d['hosts']]
        for r in range(NRED):
            rec = red_arr[e, r]
            rec['type'] = -1
            if rng.random() >= p_red:
                continue
            ag = d['red'][r]
            sess = [s[0] for s in ag['sessions']]
            sess_hosts = [s[1] for s in ag['sessions']]
            ips, hns = _bits(ag['as_ip']), _bits(ag['as_hostname'])
            subs = [b for b in range(9) if (ag['as_subnet'] >> b) & 1]

            def pick(known, everything):
                if known and rng.random() < 0.85:
                    return int(known[rng.integers(len(known))])
                return int(everything[rng.integers(len(everything))])
            t = int(rng.integers(11))
            rec['type'] = t
            rec['session'] = int(sess[rng.integers(len(sess))]) if sess and rng.random() < 0.9 else int(rng.integers(0, 6))
            rec['host'] = rec['arg'] = rec['ticks'] = rec['flags'] = 0
            rec['rate0'] = rec['rate1'] = 0.0
            if t == 0:
                rec['arg'] = pick(subs, list(range(9)))
            elif t in (1, 2, 3, 4):
                rec['host'] = pick(ips, hosts)
            elif t in (5, 6, 7):
                rec['host'] = pick(sess_hosts or hns, hosts)
            elif t == 8:
                rec['arg'] = pick(sess_hosts or hns, hosts)
                rec['host'] = rec['arg'] if rng.random() < 0.8 else pick(ips, hosts)
            if t < 9 and rng.random() < 0.5:
                rec['ticks'] = int(rng.integers(1, 4))
            if t in (1, 2) and rng.random() < 0.4:
                rec['flags'] |= 1; rec['rate0'] = float(rng.choice([0.0, 1.0, 0.5]))
            if t == 3 and rng.random() < 0.5:
                rec['flags'] |= 3; rec['rate0'] = float(rng.choice([0.0, 1.0, 0.3])); rec['rate1'] = float(rng.choice([0.0, 1.0, 0.2]))
            if rng.random() < 0.03:
                rec['flags'] |= 4                       # skip_valid_action_check
        for g in range(MAXG):
            rec = green_arr[e, g]
            rec['type'] = -1
            if g >= d['n_green'] or rng.random() >= p_green:
                continue
            t = int(rng.integers(4))
            rec['type'] = t
            rec['host'] = d['green_hosts'][g] if rng.random() < 0.97 else int(hosts[rng.integers(len(hosts))])
            rec['arg'] = rec['ticks'] = rec['flags'] = 0
            rec['session'] = 0
            rec['rate0'] = rec['rate1'] = 0.0
            if t == 0 and rng.random() < 0.7:
                rec['session'] = int(rng.integers(1, 512))          # allowed_subnets as a mask (any list: the engine validates it)
            if t in (0, 1) and rng.random() < 0.5:
                rec['flags'] |= 1; rec['rate0'] = float(rng.choice([0.0, 1.0, 0.3]))
            if t == 1 and rng.random() < 0.5:
                rec['flags'] |= 2; rec['rate1'] = float(rng.choice([0.0, 1.0, 0.2]))
            if rng.random() < 0.05:
                rec['flags'] |= 4
    return red_arr, green_arr


def with_durations(actions, rng, p=0.15):
    """Blue wrapper indices, some carrying an `action.duration` of 1..3 ticks in bits 20.. (cc4.h cc4_step_ex)."""
    a = actions.copy()
    m = rng.random(a.shape) < p
    a[m] |= (rng.integers(1, 4, size=a.shape)[m] << 20).astype(np.int32)
    return a
