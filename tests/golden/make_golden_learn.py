"""Golden vectors for the learn step from the REAL reference Agent (agent.py/model.py).
Run via tests/golden/make_golden.py in the build container."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "tests"), "/root/reference"):
    if p not in sys.path:
        sys.path.insert(0, p)

import agent as ref_agent  # noqa: E402  (reference)

import scenarios  # noqa: E402
from oracle import learner_oracle as O  # noqa: E402  (Config/init_params bookkeeping only)


class _RandnFeeder:
    """Replaces torch.randn so NoisyLinear._scale_noise (model.py:33) consumes injected draws."""

    def __init__(self):
        self.buf = None
        self.pos = 0

    def feed(self, raw):
        self.buf, self.pos = np.asarray(raw, dtype=np.float32), 0

    def __call__(self, size, device=None, **kw):
        n = int(size)
        out = torch.from_numpy(self.buf[self.pos:self.pos + n].copy())
        assert out.numel() == n, "randn feeder exhausted"
        self.pos += n
        return out


class _FakeMem:
    def __init__(self):
        self.batch = None
        self.priorities = None

    def sample(self, batch_size):
        b = self.batch
        f = lambda a: torch.from_numpy(np.asarray(a)).to(torch.float32).div_(255)  # memory.py:137-138
        return (np.arange(batch_size), f(b["states"]), torch.from_numpy(b["actions"]), torch.from_numpy(b["returns"]),
                f(b["next_states"]), torch.from_numpy(b["nonterminals"]).reshape(-1, 1), torch.from_numpy(b["weights"]))

    def update_priorities(self, idxs, priorities):
        self.priorities = np.array(priorities, dtype=np.float32)


class ReferenceLearnAdapter:
    def __init__(self, name):
        c = scenarios.LEARN_CONFIGS[name]
        h = scenarios.LEARN_HYPER
        self.args = types.SimpleNamespace(
            atoms=c["atoms"], V_min=c["v_min"], V_max=c["v_max"], batch_size=c["batch"], multi_step=c["multi_step"],
            discount=c["discount"], norm_clip=h["norm_clip"], model=None, learning_rate=h["lr"], adam_eps=h["adam_eps"],
            device=torch.device("cpu"), architecture=c["architecture"], history_length=c["history"],
            hidden_size=c["hidden"], noisy_std=0.1)
        env = types.SimpleNamespace(action_space=lambda: c["actions"])
        torch.manual_seed(0)
        self.dqn = ref_agent.Agent(self.args, env)
        self.feeder = _RandnFeeder()
        self.mem = _FakeMem()
        self.total_norm = None

    def _load(self, net, params):
        sd = net.state_dict()
        for k, v in params.items():
            sd[k] = torch.from_numpy(v.copy())
        net.load_state_dict(sd)

    def load(self, online, target):
        self._load(self.dqn.online_net, online)
        self._load(self.dqn.target_net, target)

    def _with_randn(self, raw, fn):
        real = torch.randn
        self.feeder.feed(raw)
        torch.randn = self.feeder
        try:
            return fn()
        finally:
            torch.randn = real

    def reset_noise_online(self, raw):
        self._with_randn(raw, self.dqn.reset_noise)               # main.py:151

    def learn_step(self, batch, target_raw):
        self.mem.batch = batch
        real_clip = ref_agent.clip_grad_norm_

        def spy(params, max_norm):
            params = list(params)
            self.total_norm = float(real_clip(params, max_norm))
            return self.total_norm

        ref_agent.clip_grad_norm_ = spy
        try:
            self._with_randn(target_raw, lambda: self.dqn.learn(self.mem))
        finally:
            ref_agent.clip_grad_norm_ = real_clip
        grads = {k: p.grad.detach().numpy().copy() for k, p in self.dqn.online_net.named_parameters()}
        return dict(loss=self.mem.priorities, grad_norm=self.total_norm, grads=grads)

    def params(self):
        return {k: p.detach().numpy().copy() for k, p in self.dqn.online_net.named_parameters()}

    def sync_target(self):
        self.dqn.update_target_net()                              # agent.py:102-103

    def act(self, state, noisy):
        (self.dqn.train if noisy else self.dqn.eval)()
        st = torch.from_numpy(state)
        a = self.dqn.act(st)
        q = self.dqn.evaluate_q(st)
        self.dqn.train()
        return a, q


def main():
    only = [a for a in sys.argv[1:] if not a.startswith("--")]     # optional: names of the fixtures to (re)generate
    for name in scenarios.LEARN_CONFIGS:
        if only and name not in only:
            continue
        trace = scenarios.learn_scenario(ReferenceLearnAdapter(name), name, O)
        path = os.path.join(HERE, "learn_%s.npz" % name)
        np.savez_compressed(path, **trace)
        print("wrote", path, os.path.getsize(path), "bytes,", len(trace), "arrays")


if __name__ == "__main__":
    main()
