"""Structured blue policies for the differential tools (compare.py / fuzz.py / make_golden.py): instead of uniform action
indices, every blue agent plays one kind of action.  The indices are picked from the REFERENCE wrapper's own action labels
(BlueFixedActionWrapper.action_labels), so nothing here depends on the engine's index arithmetic.

  decoy_one      DeployDecoy on the first valid host of the agent's label list, every step (stacks decoys on one host)
  decoy          DeployDecoy on a random valid host
  decoy_restore  40 steps of DeployDecoy on one host, 10 steps of Restore on it, repeated
  restore / remove / analyse   the action on a random valid host
  block          BlockTrafficZone on a random pair;  block_allow: Block or Allow, random pair
  mix            one of the above kinds, chosen per agent and step
"""
import numpy as np

KINDS = ('decoy_one', 'decoy', 'decoy_restore', 'restore', 'remove', 'analyse', 'block', 'block_allow', 'mix')


class BluePolicy:
    def __init__(self, name, labels, seed):
        """labels: {agent_name: list of action labels}"""
        assert name in KINDS, name
        self.name = name
        self.rng = np.random.default_rng(seed ^ 0xB10E)
        self.by_prefix = {}
        for agent, ls in labels.items():
            d = {}
            for i, lab in enumerate(ls):
                if lab.startswith('[Invalid]'):
                    continue
                d.setdefault(lab.split(' ')[0], []).append(i)
            self.by_prefix[agent] = d

    def _pick(self, agent, prefix):
        c = self.by_prefix[agent][prefix]
        return int(c[self.rng.integers(len(c))])

    def act(self, t):
        out = []
        for b in range(5):
            agent = f'blue_agent_{b}'
            n = self.name
            if n == 'mix':
                n = ('decoy', 'restore', 'remove', 'analyse', 'block_allow', 'decoy_one')[self.rng.integers(6)]
            if n == 'decoy_one':
                out.append(self.by_prefix[agent]['DeployDecoy'][0])
            elif n == 'decoy':
                out.append(self._pick(agent, 'DeployDecoy'))
            elif n == 'decoy_restore':
                out.append(self.by_prefix[agent]['Restore' if t % 50 >= 40 else 'DeployDecoy'][0])
            elif n == 'restore':
                out.append(self._pick(agent, 'Restore'))
            elif n == 'remove':
                out.append(self._pick(agent, 'Remove'))
            elif n == 'analyse':
                out.append(self._pick(agent, 'Analyse'))
            elif n == 'block':
                out.append(self._pick(agent, 'BlockTrafficZone'))
            else:
                out.append(self._pick(agent, 'BlockTrafficZone' if self.rng.integers(2) else 'AllowTrafficZone'))
        return np.array(out, np.int32)
