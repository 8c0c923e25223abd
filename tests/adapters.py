"""Backend adapters for tests/scenarios.py (oracle, host-interpreted kernels, HIP)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle.replay_oracle import ReplayOracle  # noqa: E402


class OracleReplayAdapter:
    def __init__(self, capacity, history, n, discount, omega):
        self.o = ReplayOracle(capacity, history=history, discount=discount, multi_step=n, priority_weight=0.4,
                              priority_exponent=omega)

    def append(self, state, action, reward, terminal):
        self.o.append(state, action, reward, terminal)

    def sample(self, batch, unit_uniforms, beta):
        self.o.priority_weight = beta
        return self.o.sample_with_uniforms(batch, unit_uniforms)

    def update_priorities(self, tree_idxs, losses):
        self.o.update_priorities(tree_idxs, losses)

    def find(self, values):
        return self.o.transitions.find(values)

    def tree(self):
        return self.o.transitions.tree.copy()

    def header(self):
        t = self.o.transitions
        return t.index, t.full, t.max

    def state_at(self, i):
        return self.o.state_at(i)
