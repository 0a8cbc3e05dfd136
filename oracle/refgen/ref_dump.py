"""Canonical text dump of the *reference* simulator state, in the format of cc4o_dump() (oracle/cc4_oracle.cpp).
Test infrastructure for parity bisecting; runs only where /root/reference exists."""
import ref_shim  # noqa: F401
from CybORG.Shared.Enums import ProcessName, SessionType

SUBNETS = ['restricted_zone_a_subnet', 'operational_zone_a_subnet', 'restricted_zone_b_subnet',
           'operational_zone_b_subnet', 'contractor_network_subnet', 'public_access_zone_subnet',
           'admin_network_subnet', 'office_network_subnet', 'internet_subnet']


def host_index(hostname):
    if hostname == 'root_internet_host_0':
        return 136
    for s, sn in enumerate(SUBNETS):
        if hostname.startswith(sn + '_'):
            rest = hostname[len(sn) + 1:]
            if rest == 'router':
                return s * 17
            if rest.startswith('user_host_'):
                return s * 17 + 1 + int(rest[len('user_host_'):])
            if rest.startswith('server_host_'):
                return s * 17 + 11 + int(rest[len('server_host_'):])
    raise ValueError(hostname)


def subnet_index(name):
    return SUBNETS.index(str(getattr(name, 'value', name)))


KIND = {ProcessName.SSHD: 0, ProcessName.OTSERVICE: 1, ProcessName.APACHE2: 2, ProcessName.MYSQLD: 3, ProcessName.SMTP: 4,
        'apache2': 5, 'Tomcat.exe': 6, 'tomcat': 6, 'haraka': 7, 'vsftpd': 8,
        None: 9, SessionType.UNKNOWN: 9, SessionType.VELOCIRAPTOR_SERVER: 9, SessionType.GREY_SESSION: 10,
        SessionType.RED_ABSTRACT_SESSION: 11, 'cmd.sh': 12}
FSM = {'K': 0, 'KD': 1, 'S': 2, 'SD': 3, 'U': 4, 'UD': 5, 'R': 6, 'RD': 7, 'F': 8}


def dump(env):
    from CybORG.Shared.Session import RedAbstractSession
    ec = env.environment_controller
    st = ec.state
    out = []
    blocks = [0] * 9
    for to, frs in st.blocks.items():
        for fr in frs:
            blocks[subnet_index(to)] |= 1 << subnet_index(fr)
    out.append(f"step {ec.step_count} phase {st.mission_phase} blocks " + ' '.join(str(b) for b in blocks))
    ip2h = {str(ip): host_index(h) for ip, h in st.ip_addresses.items()}
    for hostname, host in st.hosts.items():
        h = host_index(hostname)
        # (a hand-made service process without open_ports -- the reference's tests add such -- is the engine's K_PLAIN = 13)
        procs = ' '.join(f"({p.pid},{13 if (KIND[p.name] < 9 and not p.open_ports) else KIND[p.name]},{1 if p.user in ('root', 'SYSTEM') else 0})" for p in host.processes)
        svcs = ' '.join(f"({KIND[k]},{1 if v.active else 0},{v._percent_reliable},{v.process})" for k, v in host.services.items())
        e = host.events
        ev = f"{int(len(e.network_connections) > 0)}{int(len(e.process_creation) > 0)}{int(len(e.old_network_connections) > 0)}{int(len(e.old_process_creation) > 0)}"
        out.append(f"host {h} procs{' ' if procs else ''}{procs} svcs{' ' if svcs else ''}{svcs} ev {ev}")
    for r in range(6):
        name = f'red_agent_{r}'
        ai = ec.agent_interfaces[name]
        sess = ' '.join(f"({sid},{host_index(s.hostname)},{s.pid},{int(isinstance(s, RedAbstractSession))},{int(s.username in ('root', 'SYSTEM'))})"
                        for sid, s in st.sessions[name].items())
        known = ' '.join(str(k) for k, v in ai.action_space.server_session.items() if v)
        ag = ai.agent
        if hasattr(ag, 'host_states'):   # live (non-'F') hosts in dict order, then the 'F' hosts by host id (cc4o_dump convention)
            ent = [(ip2h[ip], FSM[d['state']], int(d['hostname'] is not None)) for ip, d in ag.host_states.items()]
            ent = [t for t in ent if t[1] != 8] + sorted(t for t in ent if t[1] == 8)
            fsm = ' '.join(f"({a},{b},{c})" for a, b, c in ent)
        else:
            fsm = ''
        subn = 0
        cidr2s = {c: subnet_index(n) for n, c in st.subnet_name_to_cidr.items()}
        for c, v in ai.action_space.subnet.items():
            if v:
                subn |= 1 << cidr2s[c]
        aip = ec.actions_in_progress.get(name)
        busy = int(aip is not None)
        out.append(f"red {r} active {int(ai.active)} sess{' ' if sess else ''}{sess} known{' ' if known else ''}{known} "
                   f"fsmstep {getattr(ag, 'step', 0)} fsm{' ' if fsm else ''}{fsm} subnets {subn} busy {busy} qt {type(aip['action']).__name__ if aip else -1}")
    for b in range(5):
        name = f'blue_agent_{b}'
        srv = st.sessions[name][0]
        ent = []
        for hn, pids in getattr(srv, 'sus_pids', {}).items():
            for p in pids:
                ent.append((host_index(hn), p))
        ent.sort(key=lambda t: t[0])
        s = ' '.join(f"({h},{p})" for h, p in ent)
        out.append(f"blue {b} sus{' ' if s else ''}{s}")
    return '\n'.join(out) + '\n'
