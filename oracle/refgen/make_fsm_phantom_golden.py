#!/usr/bin/env python
"""What the reference does when a FiniteStateRedAgent reads the observation of a submitted hostname-keyed action on a host whose hostname it has
never seen (VERDICT r05 missing 4).  Runs the reference's own agent class (imported from /root/reference) and records the outcome in
tests/golden/fsm_phantom_host.json:
  * _process_new_observations files the host under host_states[None] (FiniteStateRedAgent.py:190-236) -- no exception at that step;
  * the next DiscoverRemoteSystems result the agent processes raises ipaddress.AddressValueError (IPv4Address(None), :141-143);
  * the phantom being chosen raises TypeError (_choose_host_and_action returns a bare Sleep(), :297-299, which get_action unpacks, :113).
The engine raises E_UNREACHABLE on the step that would put the agent on that path (csrc/cc4_engine.h fsm_observe)."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: F401
import numpy as np
from ipaddress import IPv4Network
from CybORG.Agents import FiniteStateRedAgent
from CybORG.Simulator.Actions import PrivilegeEscalate, DiscoverRemoteSystems
from CybORG.Shared.Enums import TernaryEnum

out = {'numpy': np.__version__}
ag = FiniteStateRedAgent(name='red_agent_0', np_random=np.random.default_rng(1))
ag.agent_subnets = [IPv4Network('10.0.0.0/24')]
ag.step = 3
ag.host_states = {'10.0.0.5': {'state': 'U', 'hostname': 'h5'}}
hn = 'contractor_network_subnet_user_host_1'
obs = {hn: {'Sessions': [{'session_id': 1, 'agent': 'red_agent_0'}]}}
try:
    ag._host_state_transition(PrivilegeEscalate(hostname=hn, session=0, agent='red_agent_0'), TernaryEnum.TRUE)
    ag._process_new_observations(obs)
    out['at_the_observation'] = 'no exception'
    out['host_states_keys'] = [str(k) for k in ag.host_states]
    out['phantom'] = ag.host_states[None]
except Exception as e:      # noqa: BLE001
    out['at_the_observation'] = type(e).__name__
try:
    ag._host_state_transition(DiscoverRemoteSystems(subnet=IPv4Network('10.0.0.0/24'), session=0, agent='red_agent_0'), TernaryEnum.TRUE)
    out['next_discover_remote_systems_result'] = 'no exception'
except Exception as e:      # noqa: BLE001
    out['next_discover_remote_systems_result'] = type(e).__name__


class PickNone:
    def choice(self, a, p=None):
        return None

    def random(self):
        return 0.9


ag.np_random = PickNone()
try:
    chosen_host, action = ag._choose_host_and_action({'action': {}}, [h for h in ag.host_states if ag.host_states[h]['state'] != 'F'])
    out['phantom_chosen'] = 'no exception'
except Exception as e:      # noqa: BLE001
    out['phantom_chosen'] = type(e).__name__
p = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden', 'fsm_phantom_host.json')
json.dump(out, open(p, 'w'), indent=1)
print(json.dumps(out, indent=1))
