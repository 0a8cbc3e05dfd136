"""True-state view of an episode, decoded from the packed state.

Mirrors what the reference exposes through `CybORG.get_true_state(info)` (CybORG/env.py:297-309 ->
State.get_true_state, Simulator/State.py:150-224) and `TrueStateTableWrapper`
(Agents/Wrappers/TrueStateWrapper.py:25-243): per hostname a dict with 'Interface', 'Processes', 'Sessions',
'System info' (plus 'Services', which the reference reports through the same call).  Only what the simulator tracks is
reported; the reference's static decorations (paths, OS versions, user tables, files) are not modelled.

`decode(json_text)` works on the document produced by `cc4_get_true_state` (schema: csrc/cc4_export.h).
"""
import json
from ipaddress import IPv4Address, IPv4Network

SUBNETS = ('restricted_zone_a_subnet', 'operational_zone_a_subnet', 'restricted_zone_b_subnet',
           'operational_zone_b_subnet', 'contractor_network_subnet', 'public_access_zone_subnet',
           'admin_network_subnet', 'office_network_subnet', 'internet_subnet')
# process / service kinds (csrc/cc4_state.h K_*): name reported as process_name, listening port of the service
KIND_NAME = ('SSHD', 'OTSERVICE', 'APACHE2', 'MYSQLD', 'SMTP', 'apache2', 'tomcat', 'haraka', 'vsftpd',
             'VELOCIRAPTOR_SERVER', 'GREY_SESSION', 'RED_ABSTRACT_SESSION', 'cmd.sh')
KIND_PORT = (22, 1, 80, 3390, 25, 80, 443, 25, 80, None, None, None, None)
DECOY_KINDS = (5, 6, 7, 8)
RS_ABSTRACT, RS_ROOT, RS_ORIG, RS_CHILD = 1, 2, 4, 8


def hostname_of(h):
    """Host id (subnet*17 + slot; 136 = internet root) -> the reference's hostname (EnterpriseScenarioGenerator.py:313-371)."""
    if h == 136:
        return 'root_internet_host_0'
    s, slot = divmod(h, 17)
    if slot == 0:
        return f'{SUBNETS[s]}_router'
    if slot <= 10:
        return f'{SUBNETS[s]}_user_host_{slot - 1}'
    return f'{SUBNETS[s]}_server_host_{slot - 11}'


BLUE_ACTION = ('Sleep', 'Monitor', 'Analyse', 'Remove', 'Restore', 'DeployDecoy', 'BlockTrafficZone', 'AllowTrafficZone')
RED_ACTION = ('DiscoverRemoteSystems', 'AggressiveServiceDiscovery', 'StealthServiceDiscovery', 'DiscoverDeception',
              'ExploitRemoteService', 'PrivilegeEscalate', 'Impact', 'DegradeServices', 'Withdraw', 'Sleep', 'InvalidAction', 'Sleep')


class LastAction:
    """What CybORG.get_last_action(agent) reports for the accelerated path: the action's class name and parameters, with the
    reference's str() form (Action.__str__: class name followed by hostname / ip_address / subnet when it has one)."""

    def __init__(self, name, agent, **params):
        self.name, self.agent, self.session = name, agent, 0
        self.__dict__.update(params)
        self._params = params

    def __str__(self):
        for k in ('hostname', 'ip_address', 'subnet'):
            if k in self._params and self.name not in ('BlockTrafficZone', 'AllowTrafficZone'):
                return f'{self.name} {self._params[k]}'
        return self.name

    def __repr__(self):
        return f'LastAction({self})'


class TrueState:
    """One episode's decoded state.  `.hosts[hostname]` is the reference-shaped dict; `.raw` the parsed document."""

    def __init__(self, doc):
        self.raw = doc if isinstance(doc, dict) else json.loads(doc)
        d = self.raw
        self.step, self.phase = d['step'], d['phase']
        self.cidr = {SUBNETS[s]: IPv4Network(f'10.0.{d["cidr"][s]}.0/24') for s in range(9)}
        self.blocks = {SUBNETS[to]: [SUBNETS[f] for f in range(9) if (d['blocks'][to] >> f) & 1] for to in range(9) if d['blocks'][to]}
        green_of = {h: g for g, h in enumerate(d['green_hosts'])}
        blue_zone = {0: 0, 1: 1, 2: 2, 3: 3, 5: 4, 6: 4, 7: 4}    # subnet -> blue agent (ESG.py:643-649)
        sess_by_host = {}
        for r, ag in enumerate(d['red']):
            for sid, h, pid, fl in ag['sessions']:
                sess_by_host.setdefault(h, []).append({
                    'agent': f'red_agent_{r}', 'session_id': sid, 'PID': pid,
                    'Type': 'RED_ABSTRACT_SESSION' if fl & RS_ABSTRACT else 'UNKNOWN',
                    'username': 'root' if fl & RS_ROOT else 'user'})
        # blue session ids: the VelociraptorServer is 0, the clients count up in creation order (State.py:103-136)
        blue_sid = {}
        for b, ag in enumerate(d['blue']):
            blue_sid[ag['parent']] = 0
        counters = [1] * 5
        for hd in d['hosts']:     # creation order = allowed_subnets order then host order; recomputed below per agent
            pass
        self.hosts = {}
        self.ip_map = {}
        order = {b: [] for b in range(5)}
        for hd in d['hosts']:
            s = hd['h'] // 17
            if hd['blue'] and s in blue_zone:
                order[blue_zone[s]].append(hd['h'])
        alloc = {0: [0], 1: [1], 2: [2], 3: [3], 4: [5, 6, 7]}    # allowed_subnets order of each blue agent
        for b in range(5):
            parent = d['blue'][b]['parent']
            k = 1
            for sn in alloc[b]:
                for h in sorted(x for x in order[b] if x // 17 == sn):
                    if h == parent:
                        continue
                    blue_sid[h] = k
                    k += 1
        for hd in d['hosts']:
            h = hd['h']
            name = hostname_of(h)
            s = h // 17
            ip = IPv4Address(f'10.0.{d["cidr"][s]}.{hd["ip"]}')
            self.ip_map[name] = ip
            procs = []
            svc_kind_of = {pid_: k_ for k_, _a, _rel, pid_ in hd['svcs']}
            for pid, kind, root in hd['procs']:
                if kind == 13:            # K_PLAIN: a port-less process added by cc4_edit_state op 1; it carries its service's name
                    p = {'PID': pid, 'process_name': KIND_NAME[svc_kind_of.get(pid, 1)], 'username': 'root' if root else 'user'}
                    procs.append(p)
                    continue
                p = {'PID': pid, 'process_name': KIND_NAME[kind], 'username': 'root' if root else 'user'}
                if KIND_PORT[kind] is not None:
                    p['Connections'] = [{'local_port': KIND_PORT[kind], 'local_address': IPv4Address('0.0.0.0')}]
                if kind in DECOY_KINDS:
                    p['Properties'] = ['decoy']
                procs.append(p)
            # a service whose process is gone (stopped by Impact, or replaced by a restore) still reports its PID: the
            # reference's true state lists such a PID as a bare process entry (State.get_true_state merges 'Services')
            have = {p['PID'] for p in procs}
            for k, a, rel, pid in hd['svcs']:
                if pid not in have:
                    procs.append({'PID': pid})
                    have.add(pid)
            sessions = []
            if hd['blue'] and s in blue_zone:
                sessions.append({'agent': f'blue_agent_{blue_zone[s]}', 'session_id': blue_sid.get(h, 0), 'PID': hd['blue'],
                                 'Type': 'VELOCIRAPTOR_SERVER' if d['blue'][blue_zone[s]]['parent'] == h else 'UNKNOWN', 'username': 'ubuntu'})
            if hd['green'] and h in green_of:
                sessions.append({'agent': f'green_agent_{green_of[h]}', 'session_id': 0, 'PID': hd['green'], 'Type': 'GREY_SESSION', 'username': 'ubuntu'})
            sessions += sess_by_host.get(h, [])
            entry = {'Interface': [{'interface_name': 'eth0', 'ip_address': ip, 'Subnet': self.cidr[SUBNETS[s]]}],
                     'Processes': procs,
                     'Services': {KIND_NAME[k]: {'active': bool(a), 'reliability': rel, 'PID': pid} for k, a, rel, pid in hd['svcs']},
                     'System info': {'Hostname': name},
                     'Events': {'network_connections': bool(hd['ev'] & 1), 'process_creation': bool(hd['ev'] & 2),
                                'old_network_connections': bool(hd['ev'] & 4), 'old_process_creation': bool(hd['ev'] & 8)}}
            if sessions:
                entry['Sessions'] = sessions
            self.hosts[name] = entry
        self.last_action = {}
        for b, (ty, host, arg) in enumerate(d.get('last_blue', [])):
            name, agent = BLUE_ACTION[ty], f'blue_agent_{b}'
            if ty in (2, 3, 4, 5):
                self.last_action[agent] = LastAction(name, agent, hostname=hostname_of(host))
            elif ty in (6, 7):
                self.last_action[agent] = LastAction(name, agent, to_subnet=SUBNETS[host], from_subnet=SUBNETS[arg])
            else:
                self.last_action[agent] = LastAction(name, agent)
        for r, (ty, host, arg, executed) in enumerate(d.get('last_red', [])):
            name, agent = RED_ACTION[min(ty, 11)], f'red_agent_{r}'
            if ty == 0:
                self.last_action[agent] = LastAction(name, agent, subnet=self.cidr[SUBNETS[arg]])
            elif ty in (1, 2, 4):
                self.last_action[agent] = LastAction(name, agent, ip_address=IPv4Address(f'10.0.{d["cidr"][host // 17]}.{next(x["ip"] for x in d["hosts"] if x["h"] == host)}'))
            elif ty == 3:     # DiscoverDeception carries an ip_address and prints the hostname its execute() resolved (None if it never ran)
                self.last_action[agent] = LastAction(name, agent, hostname=hostname_of(host) if executed else None)
            elif ty in (5, 6, 7):
                self.last_action[agent] = LastAction(name, agent, hostname=hostname_of(host))
            else:
                self.last_action[agent] = LastAction(name, agent)
        self.sus_pids = {f'blue_agent_{b}': {} for b in range(5)}
        for b, ag in enumerate(d['blue']):
            for h, pid in ag['sus']:
                self.sus_pids[f'blue_agent_{b}'].setdefault(hostname_of(h), []).append(pid)

    def red_access(self):
        """The feed of the reference's VisualiseRedExpansion (Agents/Wrappers/VisualiseRedExpansion.py): hostname ->
        'root' | 'user' for every host on which some red agent holds a shell (the strongest one counts)."""
        out = {}
        for name, e in self.hosts.items():
            lv = [s['username'] for s in e.get('Sessions', []) if s['agent'].startswith('red')]
            if lv:
                out[name] = 'root' if 'root' in lv else 'user'
        return out

    def as_dict(self, info=None):
        """`get_true_state(info)` shape: {hostname: {...}, 'success': TRUE}; `info` (hostname -> wanted keys) filters hosts."""
        out = {n: v for n, v in self.hosts.items() if info is None or n in info}
        out['success'] = True
        return out


def decode(doc):
    return TrueState(doc)


# ---------------------------------------------------------------------------------------------------------------------
# Blue dict observations: what CybORG.get_observation('blue_agent_b') returns after a step, rebuilt from
# the per-step event log (cc4_enable_event_log).  The end-of-turn Monitor (Monitor.py:35-74) reports, per host of the
# agent's zone, the network_connections entries and then the process_creation entries through Observation.add_process
# (Shared/Observation.py:56-209); the agent's ObservationSet then combines that observation with itself once
# (Observation.combine_obs, :646-679), which is why connection entries appear twice and pid entries once, last.
_BLUE_ZONE = {0: (0,), 1: (1,), 2: (2,), 3: (3,), 4: (5, 6, 7)}
_OS = ({'OSDistribution': 'UBUNTU', 'OSVersion': 'UNKNOWN'}, {'OSDistribution': 'KALI', 'OSVersion': 'K2019_4'})


class _Obs:
    """The subset of Shared/Observation.py the Monitor path exercises (add_process / add_interface_info / add_system_info)."""

    def __init__(self):
        self.data = {}

    def add_interface(self, hostid, ip):
        lst = self.data.setdefault(hostid, {}).setdefault('Interface', [])
        new = {}
        for itf in list(lst):                      # an entry with the same address is replaced and moves to the end
            if itf.get('ip_address') == ip:
                if len(itf) > len(new):
                    new = itf
                lst.remove(itf)
        new['ip_address'] = ip
        lst.append(new)

    def add_process(self, hostid, pid=None, local_port=None, remote_port=None, local_address=None, remote_address=None):
        procs = self.data.setdefault(hostid, {}).setdefault('Processes', [])
        new = {}
        if pid is not None:
            for old in procs:
                if old.get('PID') == pid:
                    new = old
                    procs.remove(old)
                    break
            new['PID'] = pid
        conn = {}
        new.setdefault('Connections', [])
        if local_port is not None:
            conn['local_port'] = local_port
        if remote_port is not None:
            conn['remote_port'] = remote_port
        if local_address is not None:
            conn['local_address'] = local_address
            self.add_interface(hostid, local_address)
        if remote_address is not None:
            conn['remote_address'] = remote_address
        if conn:
            new['Connections'].append(conn)
        elif new['Connections'] == []:
            new.pop('Connections')
        procs.append(new)

    def add_system_info(self, hostid, info):
        self.data.setdefault(hostid, {}).setdefault('System info', {}).update(info)

    def combine(self, other):
        for key, info in other.items():
            for p in info.get('Processes', []):
                if 'Connections' in p:
                    for c in p['Connections']:
                        self.add_process(key, pid=p.get('PID'), **c)
                else:
                    self.add_process(key, pid=p.get('PID'))
            for itf in info.get('Interface', []):
                self.add_interface(key, itf['ip_address'])
            if 'System info' in info:
                self.add_system_info(key, info['System info'])


def blue_observations(ts, combined=2):
    """{agent: observation dict} of the five blue agents for the step the TrueState was taken after.  Needs the event log
    (CC4VecEnv.enable_event_log()); the keys follow the reference: 'success', 'action', then one entry per host with events.
    `combined`: ObservationSet.get_combined_observation (Shared/ObservationSet.py:22-31) folds the end-of-turn Monitor's
    observation into the action's observation object in place, every time it is called: the dicts returned by the step
    carry it once (combined=1), a later CybORG.get_observation(agent) twice (the default, what the goldens record)."""
    d = ts.raw
    if 'events' not in d:
        raise RuntimeError('blue_observations needs the event log: call enable_event_log() before stepping')
    if d['events_total'] > len(d['events']):
        raise RuntimeError(f"event log truncated: {d['events_total']} entries in step {d['events_step']}")
    ip_of = {hd['h']: IPv4Address(f'10.0.{d["cidr"][hd["h"] // 17]}.{hd["ip"]}') for hd in d['hosts']}
    os_of = {hd['h']: hd['os'] for hd in d['hosts']}
    by_host = {}
    decoys = {}
    for order, seq, host, kind, laddr, lport, raddr, rport, pid, rep in sorted(d['events'], key=lambda e: (e[0], e[1])):
        if kind == 2:
            decoys[host] = (laddr, pid)               # DeployDecoy resolved on this host: (K_* kind, pid)
        else:
            by_host.setdefault(host, ([], []))[kind].extend([(laddr, lport, raddr, rport, pid)] * rep)
    out = {}
    for b in range(5):
        agent = f'blue_agent_{b}'
        parent = d['blue'][b]['parent']
        zone = [hd['h'] for hd in d['hosts'] if hd['h'] // 17 in _BLUE_ZONE[b]]
        hosts = [h for h in zone if h != parent] + [parent]       # the child sessions' hosts, then the server's own
        m = _Obs()
        for h in hosts:
            conns, procs = by_host.get(h, ([], []))
            name = hostname_of(h)
            info = dict(Architecture='x64', Hostname=name, OSType='LINUX', position=[0.0, 0.0], **_OS[os_of[h]])
            for lst in (conns, procs):
                if lst:
                    m.add_system_info(name, info)
                for laddr, lport, raddr, rport, pid in lst:
                    m.add_process(name, pid=pid or None, local_port=lport or None, remote_port=rport or None,
                                  local_address=ip_of[laddr] if laddr != 255 else None,
                                  remote_address=ip_of[raddr] if raddr != 255 else None)
        obs = _Obs()
        la = ts.last_action[agent]
        if la.name == 'DeployDecoy' and not d['blue'][b].get('busy'):
            h = next((hd['h'] for hd in d['hosts'] if hostname_of(hd['h']) == la.hostname), None)
            if h in decoys:                           # the action's own observation comes first (DecoyAction.py:105-113)
                kind, pid = decoys[h]
                p = {'PID': pid, 'PPID': 1, 'service_name': KIND_NAME[kind], 'username': 'ubuntu'}
                if kind != 7:                         # HarakaDecoyFactory has no PROPERTIES
                    p['Properties'] = ['rfi']
                obs.data.setdefault(la.hostname, {}).setdefault('Processes', []).append(p)
        if la.name == 'Analyse' and not d['blue'][b].get('busy'):
            bits = next((hd.get('files', 0) for hd in d['hosts'] if hostname_of(hd['h']) == la.hostname), 0)
            names = [n for n, bit in (('cmd.sh', 1), ('escalate.sh', 2)) if bits & bit]
            if len(names) == 2 and not bits & 4:      # the name appended last is listed last (add_file_info re-appends)
                names.reverse()
            if names:                                 # DensityScout + SigCheck over Host.files (Analyse.py:55-71)
                obs.data.setdefault(la.hostname, {})['Files'] = [
                    {'Density': 0.9, 'File Name': n, 'Known File': 'UNKNOWN', 'Known Path': 'TEMP', 'Path': '/tmp/'} for n in names]
        for _ in range(combined):
            obs.combine(m.data)
        # 'success' / 'action' of the agent's own action: IN_PROGRESS (and no 'action' key) while a multi-tick action runs;
        # Sleep reports UNKNOWN; Monitor, Analyse, Remove, Restore and DeployDecoy report TRUE once they resolve (their
        # parameters are validated on submission); Block/AllowTrafficZone report whether the pair's state changed
        # (BlueAgent.last_ok).
        if d['blue'][b].get('busy'):
            res = {'success': Ternary('IN_PROGRESS')}
        else:
            ok = {'Sleep': 'UNKNOWN', 'Monitor': 'TRUE', 'Analyse': 'TRUE', 'Remove': 'TRUE', 'Restore': 'TRUE',
                  'DeployDecoy': 'TRUE'}.get(la.name)
            if la.name in ('BlockTrafficZone', 'AllowTrafficZone'):
                ok = {1: 'TRUE', 3: 'FALSE'}.get(d['blue'][b].get('traffic_ok'))
            res = {'success': Ternary(ok) if ok is not None else None, 'action': la}
        res.update(obs.data)
        out[agent] = res
    return out


class _Table:
    """Minimal stand-in for prettytable.PrettyTable (not installed in the image): field names, rows, str()."""
    def __init__(self, field_names):
        self.field_names = list(field_names)
        self.rows = []

    def add_row(self, row):
        self.rows.append([str(c) for c in row])

    def __str__(self):
        w = [max(len(self.field_names[i]), *(len(r[i]) for r in self.rows)) if self.rows else len(self.field_names[i])
             for i in range(len(self.field_names))]
        line = '+' + '+'.join('-' * (x + 2) for x in w) + '+'
        fmt = lambda r: '|' + '|'.join(' ' + r[i].ljust(w[i]) + ' ' for i in range(len(w))) + '|'   # noqa: E731
        return '\n'.join([line, fmt(self.field_names), line] + [fmt(r) for r in self.rows] + [line])


class TrueStateTableWrapper:
    """Counterpart of the reference's TrueStateTableWrapper (TrueStateWrapper.py:6-243) over a cage_challenge_4_amd
    `CybORG` (or anything with `.get_true_state()`): host overview and per-subnet process tables."""

    def __init__(self, env):
        self.env = env
        self.hostnames = [n for n in env.get_true_state() if n != 'success']

    def get_raw_full_true_state(self):
        return self.env.get_true_state()

    def get_host_overview_table(self):
        t = _Table(['Hostname', 'IP Address', 'Sessions', 'No. Processes'])
        ts = self.env.get_true_state()
        ts.pop('success')
        for hostname, st in ts.items():
            sess = [s['agent'] for s in st['Sessions']] if 'Sessions' in st else '-'
            t.add_row([hostname, st['Interface'][0]['ip_address'], sess, len(st.get('Processes', []))])
        return t

    def get_host_processes_tables(self):
        ts = self.env.get_true_state()
        ts.pop('success')
        tables = {sn: _Table(['Hostname', 'PID', 'Name', 'Username', 'Session', 'SID']) for sn in SUBNETS}
        for hostname, st in ts.items():
            sn = next(s for s in SUBNETS if hostname.startswith(s) or (s == 'internet_subnet' and hostname == 'root_internet_host_0'))
            by_pid = {s['PID']: s for s in st.get('Sessions', [])}
            for p in st.get('Processes', []):
                s = by_pid.get(p['PID'])
                tables[sn].add_row([hostname, p['PID'], p.get('process_name', '-'), p.get('username', '-'), s['agent'] if s else '-', s['session_id'] if s else '-'])
        return tables


# ---------------------------------------------------------------------------------------------------------------------
# Red dict observations: what CybORG.get_observation('red_agent_r') / Results.observation hold after a step
# (env.py:270-283; what a red agent's get_action consumes, FiniteStateRedAgent.py:190-250).  The engine keeps, per red agent and step,
# the ordered keys of the combined observation with what each holds (RedAgent.obs: ip- or hostname-keyed; 'Sessions' /
# 'Interface' / 'System info'), `success` and the action of observations[0], whether the end-of-turn RedSessionCheck listed the
# agent's sessions (RedSessionCheck.py:57-65: one hostname-keyed entry per session with Sessions / Interface{ip, Subnet} /
# System info), and the session an exploit created.  From that and the true state this rebuilds the dict: 'success', 'action',
# every key with its 'Interface' (ip_address, plus Subnet where the reference reports it: Pingsweep and the session listing),
# 'Sessions' (session_id, agent, username, Type), 'System info' {'Hostname'}, and for a service discovery the target's open ports
# as 'Processes' [{'Connections': [{'local_port', 'local_address'}]}] (Portscan.py:44-64).  Not rebuilt: the connection / process
# details of an exploit's own observation and the process descriptions DegradeServices / DiscoverDeception attach (ephemeral
# ports, paths, versions) -- nothing on the hot path or in the built-in agents reads them.
OE_KEY_IP, OE_SESS, OE_IFACE, OE_SYSHN = 1, 2, 4, 8


class Ternary(str):
    """Shared/Enums.py TernaryEnum as the dict observations of this package carry it: a str ('TRUE' / 'FALSE' / 'UNKNOWN' /
    'IN_PROGRESS') with the enum's .name and .value and its comparison with bools (TRUE == True, FALSE == False; UNKNOWN and
    IN_PROGRESS equal neither), so that code written against the reference -- obs['success'] == True, success.name -- runs unchanged."""
    _VALUE = {'TRUE': 1, 'UNKNOWN': 2, 'FALSE': 3, 'IN_PROGRESS': 4}

    @property
    def name(self):
        return str(self)

    @property
    def value(self):
        return self._VALUE[str(self)]

    def __eq__(self, other):
        if isinstance(other, bool):
            return (str(self) == 'TRUE') if other else (str(self) == 'FALSE')
        n = getattr(other, 'name', other)
        return str(self) == n

    def __ne__(self, other):
        return not self.__eq__(other)

    __hash__ = str.__hash__


# agent_interface.allowed_subnets of a green agent in mission phase p: its own subnet, then its partners in the phase's pair list
# (EnterpriseScenarioGenerator.py:281-306 + SimulationController.py:747-765); same table as green_allowed_mask (csrc/cc4_tables.h)
_GREEN_ALLOWED = ((0xf7, 0x03, 0xfd, 0x0c, 0xf5, 0x35, 0x55, 0x95, 0x100), (0xe1, 0x02, 0xfc, 0x0c, 0xf4, 0x35, 0x55, 0x95, 0x100),
                  (0xf3, 0x03, 0xe4, 0x08, 0xf1, 0x35, 0x55, 0x95, 0x100))


def green_allowed_subnets(phase, subnet):
    m = _GREEN_ALLOWED[phase][subnet]
    return [SUBNETS[subnet]] + [SUBNETS[i] for i in range(9) if (m >> i) & 1 and i != subnet]
TERNARY = {0: 'UNKNOWN', 1: 'TRUE', 2: 'UNKNOWN', 3: 'FALSE', 4: 'IN_PROGRESS'}


def _session_entry(r, sid, flags):
    return {'session_id': sid, 'agent': f'red_agent_{r}', 'username': 'root' if flags & RS_ROOT else ('ubuntu' if flags & RS_ORIG else 'user'),
            'Type': 'RED_ABSTRACT_SESSION' if flags & RS_ABSTRACT else 'SHELL'}


def red_observations(ts):
    """{agent: observation dict} of the six red agents for the step the TrueState was taken after."""
    d = ts.raw
    ip_of = {hd['h']: IPv4Address(f'10.0.{d["cidr"][hd["h"] // 17]}.{hd["ip"]}') for hd in d['hosts']}
    procs_of = {hd['h']: hd['procs'] for hd in d['hosts']}
    out = {}
    for r, ag in enumerate(d['red']):
        agent = f'red_agent_{r}'
        obs = {'success': Ternary(TERNARY[ag['obs_success']])}
        atype = ag['obs_action'][0]
        ex_type, _h, _a, executed = d['last_red'][r]
        # 'action' of observations[0]: the agent's own action when it executed (Sleep and InvalidAction included); the end-of-turn
        # RedSessionCheck when the action was dropped by filter_actions; absent while a multi-tick action runs or nothing was observed
        if ag['obs_success'] not in (0, 4):
            if atype != 11 or (executed and ex_type in (9, 10)):
                obs['action'] = ts.last_action[agent]
            elif not executed:
                obs['action'] = LastAction('RedSessionCheck', agent)
        sess_on = {}
        for sid, h, pid, fl in ag['sessions']:
            sess_on.setdefault(h, []).append(_session_entry(r, sid, fl))
        new_h, new_id = ag.get('new_session', [255, 0])
        for h, fl in ag['obs']:
            if h not in ip_of:
                continue
            key = str(ip_of[h]) if fl & OE_KEY_IP else hostname_of(h)
            e = obs.setdefault(key, {})
            if fl & OE_IFACE:
                itf = {'ip_address': ip_of[h]}
                if atype == 0 and fl & OE_KEY_IP:                     # Pingsweep reports the subnet with every address
                    itf['Subnet'] = ts.cidr[SUBNETS[h // 17]]
                e['Interface'] = [itf]
            if fl & OE_SESS:
                if fl & OE_KEY_IP and h == new_h:                     # the exploit's own report of the session it opened
                    sid_now = new_id
                    if not any(s_[0] == new_id and s_[1] == h for s_ in ag['sessions']):
                        # the session landed in another agent's zone and different_subnet_agent_reassignment handed it over within
                        # the same step (SC:820-903): the report shows the ident it has there (the last session that agent got on the host)
                        for other in d['red']:
                            if other is not ag:
                                here = [s_ for s_ in other['sessions'] if s_[1] == h]
                                if here:
                                    sid_now = here[-1][0]
                    e['Sessions'] = [{'session_id': sid_now, 'agent': agent, 'username': 'user', 'Type': 'SHELL'}]
                else:
                    e['Sessions'] = [dict(s) for s in sess_on.get(h, [])]
            if fl & OE_SYSHN:
                e['System info'] = {'Hostname': hostname_of(h)}
            if atype in (1, 2) and fl & OE_KEY_IP and h == ag['obs_action'][1] and ag['obs_success'] == 1:
                e['Processes'] = [{'Connections': [{'local_port': KIND_PORT[k], 'local_address': ip_of[h]}]}
                                  for _pid, k, _root in procs_of[h] if k < len(KIND_PORT) and KIND_PORT[k] is not None]
        if ag['rsc_listed']:
            for sid, h, pid, fl in ag['sessions']:
                e = obs.setdefault(hostname_of(h), {})
                e['Interface'] = [{'ip_address': ip_of[h], 'Subnet': ts.cidr[SUBNETS[h // 17]]}]
                lst = e.setdefault('Sessions', [])
                if not any(s['session_id'] == sid for s in lst):
                    lst.append(_session_entry(r, sid, fl))
                e['System info'] = {'Hostname': hostname_of(h)}
        out[agent] = obs
    return out


def red_obs_skeleton(obs):
    """The part of a red dict observation red_observations() rebuilds, in canonical form (works on the reference's dicts as well):
    {'success', 'action' (str or None), 'hosts': {key: {'ips': [[ip, has Subnet]...], 'sessions': [[id, username]...], 'hostname', 'ports'}}}."""
    sk = {'success': getattr(obs.get('success'), 'name', obs.get('success')), 'action': None if obs.get('action') is None else str(obs['action']).split(' ')[0],
          'hosts': {}}
    for key, v in obs.items():
        if key in ('success', 'action', 'message') or not isinstance(v, dict):
            continue
        e = {}
        if 'Interface' in v:
            e['ips'] = sorted([str(i['ip_address']), 'Subnet' in i] for i in v['Interface'] if 'ip_address' in i)
        if 'Sessions' in v:
            # (an exploit's own, ip-keyed report names the user only when it was the SSH brute force: which exploit opened a session is
            # not kept in the packed state, so the user name of that one entry is outside the canonical form)
            e['sessions'] = sorted([int(s['session_id']), None if '.' in str(key) else str(s.get('username'))] for s in v['Sessions'])
        if 'System info' in v and 'Hostname' in v['System info']:
            e['hostname'] = str(v['System info']['Hostname'])
        if sk['action'] in ('AggressiveServiceDiscovery', 'StealthServiceDiscovery') and 'Processes' in v:
            e['ports'] = sorted(int(c['local_port']) for p in v['Processes'] for c in p.get('Connections', []) if 'local_port' in c and 'remote_port' not in c)
        sk['hosts'][str(key)] = e
    return sk
