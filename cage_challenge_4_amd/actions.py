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
