"""Blue action objects with the reference's constructors (CybORG/Simulator/Actions/AbstractActions/{Monitor,Analyse,
Remove,Restore}.py, ConcreteActions/ControlTraffic.py:12-185, AbstractActions/DeployDecoy.py:8-31, Action.py Sleep).

On the accelerated path an action is a wrapper index; these classes exist so that code written against the reference --
`env.step(actions={agent: BlockTrafficZone(session=0, agent=agent, from_subnet=..., to_subnet=...)})`
(CybORG/Tests/test_cc4/test_BlueEnterpriseWrapper.py:232-254) or `CybORG.parallel_step({agent: Restore(...)})` -- drops in:
the wrappers map an object to the index of the equal action of the agent's fixed action list (BlueFixedActionWrapper.py:142-148
takes either).  Any object with the same class name and attributes (e.g. the reference's own Action instances) maps the
same way."""

DURATION = {'Sleep': 1, 'Monitor': 1, 'Analyse': 2, 'Remove': 3, 'Restore': 5, 'DeployDecoy': 2, 'BlockTrafficZone': 1,
            'AllowTrafficZone': 1}


class Action:
    cost = 0
    priority = 99

    def __init__(self):
        self.name = type(self).__name__
        self.duration = DURATION.get(self.name, 1)

    def __str__(self):     # Action.__str__: the class name followed by the hostname / ip_address / subnet parameter, if any
        h = getattr(self, 'hostname', None)
        return f'{self.name} {h}' if h is not None else self.name

    def __repr__(self):
        return str(self)


class Sleep(Action):
    def __init__(self):
        super().__init__()


class Monitor(Action):
    def __init__(self, session: int, agent: str):
        super().__init__()
        self.session, self.agent = session, agent


class _HostAction(Action):
    def __init__(self, session: int, agent: str, hostname: str):
        super().__init__()
        self.session, self.agent, self.hostname = session, agent, hostname


class Analyse(_HostAction):
    pass


class Remove(_HostAction):
    pass


class Restore(_HostAction):
    cost = -1


class DeployDecoy(_HostAction):
    def __init__(self, *, session: int, agent: str, hostname: str):
        super().__init__(session, agent, hostname)


class _ControlTraffic(Action):
    priority = 1

    def __init__(self, session: int, agent: str, from_subnet: str, to_subnet: str):
        super().__init__()
        self.session, self.agent, self.from_subnet, self.to_subnet = session, agent, from_subnet, to_subnet


class BlockTrafficZone(_ControlTraffic):
    pass


class AllowTrafficZone(_ControlTraffic):
    pass


BLUE_ACTIONS = (Sleep, Monitor, Analyse, Remove, Restore, DeployDecoy, BlockTrafficZone, AllowTrafficZone)


SUBNET_ORDER = ('restricted_zone_a_subnet', 'operational_zone_a_subnet', 'restricted_zone_b_subnet', 'operational_zone_b_subnet',
                'contractor_network_subnet', 'public_access_zone_subnet', 'admin_network_subnet', 'office_network_subnet')
BLUE_RAW_ACTION = 0x10000        # csrc/cc4_engine.h: BLUE_RAW_ACTION | BA_* type << 8 | host id
_BA = {'Analyse': 2, 'Remove': 3, 'Restore': 4, 'DeployDecoy': 5}


def raw_host_action(name, host, labels):
    """(type, host id) code of a host action the fixed list cannot express, or None: `host` must be the router of one of the
    subnets the agent's list covers."""
    if name not in _BA or not host.endswith('_router'):
        return None
    sn = host[:-len('_router')]
    if sn not in SUBNET_ORDER or not any(sn + '_' in lab for lab in labels):
        return None
    return BLUE_RAW_ACTION | (_BA[name] << 8) | (SUBNET_ORDER.index(sn) * 17)


def action_index(action, labels):
    """Index of `action` (an int, an object of the classes above, or anything with the same class name and attributes) in
    an agent's fixed action list, given the list's labels (BlueFixedActionWrapper.action_labels).  An action that names a
    host outside the episode's topology maps to its '[Invalid] ...' slot, which the engine resolves as Sleep."""
    if isinstance(action, (int,)) or type(action).__module__ == 'numpy':
        return int(action)
    name = getattr(action, 'name', None) or type(action).__name__
    if name in ('Sleep', 'Monitor'):
        return labels.index(name)
    host = getattr(action, 'hostname', None)
    if host is not None:
        for lab in (f'{name} {host}', f'[Invalid] {name} {host}'):
            if lab in labels:
                return labels.index(lab)
        # a host the agent's fixed list has no slot for -- the routers of its zone: the reference forwards the object and the
        # simulator executes it (cc4BlueRandomAgent picks routers too); the engine takes it as (type, host id) (cc4.h cc4_step)
        raw = raw_host_action(name, host, labels)
        if raw is not None:
            return raw
        raise ValueError(f'{name} {host}: not an action of this agent')
    src, dst = getattr(action, 'from_subnet', None), getattr(action, 'to_subnet', None)
    if src is not None and dst is not None:
        src, dst = str(src).lower(), str(dst).lower()
        for i, lab in enumerate(labels):
            if lab.startswith(f'{name} {dst} (') and f'<- {src} (' in lab:
                return i
        raise ValueError(f'{name} {dst} <- {src}: not an action of this agent')
    s = str(action)
    if s in labels:
        return labels.index(s)
    raise ValueError(f'cannot map {action!r} to an action index')


# ---------------------------------------------------------------------------------------------------------------------
# Red and green action objects (the step takes a submitted action for ANY agent: SimulationController.py:236-240,
# CybORG.step(agent, action), env.py:125-161).  Constructors and attribute names are the reference's
# (AbstractActions/{DiscoverRemoteSystems,DiscoverNetworkServices,DiscoverDeception,ExploitRemoteService,PrivilegeEscalate,
# Impact,DegradeServices}.py, ConcreteActions/Withdraw.py, GreenActions/{GreenLocalWork,GreenAccessService}.py); `duration`,
# `detection_rate`, `fp_rate` may be overwritten after construction, as the reference's tests do.  encode_agent_action turns
# one -- or any object with the same class name and attributes, e.g. a reference Action instance -- into the record
# cc4_step_ex takes (include/cc4.h cc4_agent_action).
RED_TYPE = {'DiscoverRemoteSystems': 0, 'AggressiveServiceDiscovery': 1, 'StealthServiceDiscovery': 2, 'DiscoverDeception': 3,
            'ExploitRemoteService': 4, 'PrivilegeEscalate': 5, 'Impact': 6, 'DegradeServices': 7, 'Withdraw': 8, 'Sleep': 9}
RED_INVALID, GREEN_ACCESS, GREEN_LOCAL, GREEN_SLEEP, GREEN_INVALID = 10, 0, 1, 2, 3
RED_DURATION = {'DiscoverRemoteSystems': 1, 'AggressiveServiceDiscovery': 1, 'StealthServiceDiscovery': 3, 'DiscoverDeception': 2,
                'ExploitRemoteService': 4, 'PrivilegeEscalate': 2, 'Impact': 2, 'DegradeServices': 2, 'Withdraw': 1, 'Sleep': 1}
DURATION.update(RED_DURATION)
DURATION.update({'GreenLocalWork': 1, 'GreenAccessService': 1, 'InvalidAction': 1})
ACT_RATE0, ACT_RATE1, ACT_SKIP_VALID = 1, 2, 4


class _ParamAction(Action):
    """str(): the class name followed by the first of hostname / ip_address / subnet (the reference's per-class __str__)."""
    def __str__(self):
        for k in ('hostname', 'ip_address', 'subnet'):
            if getattr(self, k, None) is not None:
                return f'{self.name} {getattr(self, k)}'
        return self.name


class DiscoverRemoteSystems(_ParamAction):
    def __init__(self, session: int, agent: str, subnet):
        super().__init__()
        self.session, self.agent, self.subnet = session, agent, subnet


class _ServiceDiscovery(_ParamAction):
    def __init__(self, session: int, agent: str, ip_address):
        super().__init__()
        self.session, self.agent, self.ip_address = session, agent, ip_address


class AggressiveServiceDiscovery(_ServiceDiscovery):
    def __init__(self, session: int, agent: str, ip_address):
        super().__init__(session, agent, ip_address)
        self.detection_rate = 0.75


class StealthServiceDiscovery(_ServiceDiscovery):
    def __init__(self, session: int, agent: str, ip_address):
        super().__init__(session, agent, ip_address)
        self.detection_rate = 0.25


class DiscoverDeception(_ServiceDiscovery):
    def __init__(self, session: int, agent: str, ip_address):
        super().__init__(session, agent, ip_address)
        self.detection_rate, self.fp_rate = 0.5, 0.1


class ExploitRemoteService(_ParamAction):
    def __init__(self, ip_address, session: int, agent: str):
        super().__init__()
        self.ip_address, self.session, self.agent = ip_address, session, agent


class _RedHostAction(_ParamAction):
    def __init__(self, hostname: str, session: int, agent: str):
        super().__init__()
        self.hostname, self.session, self.agent = hostname, session, agent


class PrivilegeEscalate(_RedHostAction):
    pass


class Impact(_RedHostAction):
    pass


class DegradeServices(_RedHostAction):
    pass


class Withdraw(_ParamAction):
    def __init__(self, session: int, agent: str, ip_address, hostname: str):
        super().__init__()
        self.session, self.agent, self.ip_address, self.hostname = session, agent, ip_address, hostname


class GreenLocalWork(_ParamAction):
    def __init__(self, agent: str, session_id: int, ip_address, fp_detection_rate=0.01, phishing_error_rate=0.01):
        super().__init__()
        for nm, v in (('fp_detection_rate', fp_detection_rate), ('phishing_error_rate', phishing_error_rate)):
            if not 0.0 <= v <= 1.0:
                raise ValueError(f"GreenLocalWork: {nm} must be a value equal or between 0 and 1")   # GreenLocalWork.py:52-57
        self.agent, self.session, self.ip_address = agent, session_id, ip_address
        self.fp_detection_rate, self.phishing_error_rate = fp_detection_rate, phishing_error_rate


class GreenAccessService(_ParamAction):
    def __init__(self, agent: str, session_id: int, src_ip, allowed_subnets, fp_detection_rate=0.01):
        super().__init__()
        self.agent, self.session, self.ip_address = agent, session_id, src_ip
        self.allowed_subnets, self.fp_detection_rate = allowed_subnets, fp_detection_rate
        self.dest_ip = self.dest_port = ""

    def __str__(self):
        return f'{self.name} {self.dest_ip} {self.dest_port}'


class InvalidAction(Action):
    """What replace_action_if_invalid substitutes (Action.py:52-64); as a submitted action it resolves as Observation(False)."""
    cost = -0.1

    def __init__(self, action=None, error=None):
        super().__init__()
        self.action, self.error = action, error


RED_ACTIONS = (DiscoverRemoteSystems, AggressiveServiceDiscovery, StealthServiceDiscovery, ExploitRemoteService, PrivilegeEscalate,
               DegradeServices, DiscoverDeception, Impact, Withdraw, Sleep)          # EnterpriseScenarioGenerator.py:764-768
GREEN_ACTIONS = (GreenAccessService, GreenLocalWork, Sleep)                          # :714


def _subnet_index(v, cidr_of):
    """A subnet parameter (IPv4Network / 'a.b.c.0/24' string, a SUBNET enum member or a subnet name) -> subnet index, or None."""
    s = str(getattr(v, 'value', v)).lower()
    if s in SUBNET_ORDER:
        return SUBNET_ORDER.index(s)
    if s == 'internet_subnet':
        return 8
    for i, c in enumerate(cidr_of):
        if str(c) == s:
            return i
    return None


def encode_agent_action(action, kind, host_of_ip, host_of_name, cidr_of, own_host=None, skip_valid=False):
    """One submitted red / green action as the fields of cc4_agent_action: (type, host, arg, ticks, session, flags, rate0, rate1).
    kind: 'red' | 'green'.  host_of_ip / host_of_name: {str(ip) | hostname: host id} of the episode; cidr_of: the nine subnet
    CIDRs in SUBNET order.  A parameter the episode does not know (an address outside the network, a class the agent's action
    space does not hold) makes the action what the reference makes of it: an InvalidAction."""
    name = getattr(action, 'name', None) or type(action).__name__
    dur = int(getattr(action, 'duration', DURATION.get(name, 1)))
    ticks = 0 if dur == DURATION.get(name, 1) else dur
    if not 0 <= ticks <= 255:
        raise ValueError(f'{name}: duration {dur} out of range')
    flags = ACT_SKIP_VALID if skip_valid else 0
    inv = (RED_INVALID if kind == 'red' else GREEN_INVALID, 0, 0, 0, 0, 0, 0.0, 0.0)
    if kind == 'red':
        if name not in RED_TYPE:
            return inv
        t = RED_TYPE[name]
        if name == 'Sleep':
            return (t, 0, 0, ticks, 0, 0, 0.0, 0.0)
        sid = int(action.session)
        if not 0 <= sid <= 0xFFFF:
            return inv
        host = arg = 0
        r0 = r1 = 0.0
        if name == 'DiscoverRemoteSystems':
            arg = _subnet_index(action.subnet, cidr_of)
            if arg is None:
                return inv
        elif name in ('PrivilegeEscalate', 'Impact', 'DegradeServices'):
            host = host_of_name.get(str(action.hostname))
        elif name == 'Withdraw':
            host, arg = host_of_ip.get(str(action.ip_address)), host_of_name.get(str(action.hostname))
            if arg is None:
                return inv
        else:
            host = host_of_ip.get(str(action.ip_address))
        if host is None:
            return inv
        if name in ('AggressiveServiceDiscovery', 'StealthServiceDiscovery', 'DiscoverDeception'):
            default = {'AggressiveServiceDiscovery': 0.75, 'StealthServiceDiscovery': 0.25, 'DiscoverDeception': 0.5}[name]
            if float(getattr(action, 'detection_rate', default)) != default:
                flags |= ACT_RATE0; r0 = float(action.detection_rate)
        if name == 'DiscoverDeception' and float(getattr(action, 'fp_rate', 0.1)) != 0.1:
            flags |= ACT_RATE1; r1 = float(action.fp_rate)
        return (t, host, arg, ticks, sid, flags, r0, r1)
    if name == 'Sleep':
        return (GREEN_SLEEP, 0, 0, 0, 0, 0, 0.0, 0.0)
    if name not in ('GreenLocalWork', 'GreenAccessService'):
        return inv
    if ticks > 1:
        raise NotImplementedError(f'{name}: a green action takes one tick on the accelerated path (duration {dur})')
    sess = getattr(action, 'session', 0)
    host = host_of_ip.get(str(action.ip_address))
    if host is None:
        return inv
    if own_host is not None and host != own_host:
        raise NotImplementedError(f"{name}: a green agent acts from its own host ({action.ip_address} is another host's address)")
    if sess != 0:
        raise NotImplementedError(f'{name}: session_id {sess} (a green agent holds session 0 only)')
    r0 = r1 = 0.0
    if float(action.fp_detection_rate) != 0.01:
        flags |= ACT_RATE0; r0 = float(action.fp_detection_rate)
    if name == 'GreenLocalWork':
        if float(action.phishing_error_rate) != 0.01:
            flags |= ACT_RATE1; r1 = float(action.phishing_error_rate)
        return (GREEN_LOCAL, host, 0, 0, 0, flags, r0, r1)
    mask = 0
    for sn in action.allowed_subnets:
        i = _subnet_index(sn, cidr_of)
        if i is None:
            return inv
        mask |= 1 << i
    return (GREEN_ACCESS, host, 0, 0, mask, flags, r0, r1)
