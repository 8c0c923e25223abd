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


# --------------------------------------------------------------------------- learn step
from oracle import learner_oracle as O  # noqa: E402
import scenarios  # noqa: E402


class OracleLearnAdapter:
    def __init__(self, name):
        self.cfg = O.Config(**scenarios.LEARN_CONFIGS[name])
        self.h = scenarios.LEARN_HYPER

    def load(self, online, target):
        self.online = {k: v.copy() for k, v in online.items()}
        self.target = {k: v.copy() for k, v in target.items()}
        self.adam = O.AdamOracle(self.online, self.h["lr"], self.h["adam_eps"])
        self.noise_online = None

    def reset_noise_online(self, raw):
        self.noise_online = O.make_noise(self.cfg, raw)

    def learn_step(self, batch, target_raw):
        out = O.learn(self.cfg, self.online, self.target, self.noise_online, O.make_noise(self.cfg, target_raw), batch)
        total, clipped = O.clip_grads(out["grads"], self.h["norm_clip"])
        self.online = self.adam.step(clipped)
        self.last = out
        return dict(loss=out["loss"], grad_norm=total, grads=clipped)

    def params(self):
        return self.online

    def sync_target(self):
        self.target = {k: v.copy() for k, v in self.online.items()}     # agent.py:102-103 (the noise is re-drawn every step)

    def act(self, state, noisy):
        return O.act(self.cfg, self.online, self.noise_online if noisy else None, state)
