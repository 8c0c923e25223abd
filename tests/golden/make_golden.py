"""Generates tests/golden/*.npz by running the REAL reference classes.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py
The GPU box has no /root/reference; tests there read the committed .npz files.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF = "/root/reference"
sys.path.insert(0, REF)

import memory as ref_memory  # noqa: E402  (the reference's memory.py)

import scenarios  # noqa: E402


class ReferenceReplayAdapter:
    """Drives reference ReplayMemory; the sampler's np.random.uniform draws are made to
    consume exactly the injected unit uniforms by seeding a private RandomState whose
    random_sample stream equals them (uniform(0,seg) == 0.0 + seg*random_sample())."""

    def __init__(self, capacity, history, n, discount, omega):
        args = types.SimpleNamespace(device=torch.device("cpu"), history_length=history, discount=discount,
                                     multi_step=n, priority_weight=0.4, priority_exponent=omega)
        self.m = ref_memory.ReplayMemory(args, capacity)

    def append(self, state, action, reward, terminal):
        self.m.append(torch.from_numpy(state), action, reward, terminal)

    def sample(self, batch, unit_uniforms, beta):
        self.m.priority_weight = beta
        feed = iter(np.asarray(unit_uniforms, dtype=np.float64))

        def fake_uniform(low, high, size):
            u = next(feed)
            assert list(size) == [batch] and low == 0.0
            return np.float64(low) + (np.float64(high) - np.float64(low)) * u

        real = np.random.uniform
        np.random.uniform = fake_uniform
        try:
            tree_idxs, states, actions, returns, next_states, nonterminals, weights = self.m.sample(batch)
        finally:
            np.random.uniform = real
        s8 = np.rint(states.numpy() * 255).astype(np.uint8)
        n8 = np.rint(next_states.numpy() * 255).astype(np.uint8)
        assert np.array_equal(s8.astype(np.float32) / np.float32(255), states.numpy())
        assert np.array_equal(n8.astype(np.float32) / np.float32(255), next_states.numpy())
        return dict(tree_idxs=np.asarray(tree_idxs, dtype=np.int64), states=s8, next_states=n8,
                    actions=actions.numpy(), returns=returns.numpy(), nonterminals=nonterminals.numpy(),
                    weights=weights.numpy())

    def update_priorities(self, tree_idxs, losses):
        self.m.update_priorities(tree_idxs, losses)

    def find(self, values):
        return self.m.transitions.find(values)

    def tree(self):
        return self.m.transitions.sum_tree.copy()

    def header(self):
        t = self.m.transitions
        return t.index, t.full, t.max

    def state_at(self, i):
        self.m.current_idx = i
        return self.m.__next__().numpy()


def check_fake_uniform_is_faithful():
    """np.random.uniform(0.0, seg32, [B]) must equal 0.0 + float64(seg32) * random_sample(B)."""
    seg = np.float32(0.7315)
    a = np.random.RandomState(5).uniform(0.0, seg, [64])
    b = 0.0 + np.float64(seg) * np.random.RandomState(5).random_sample(64)
    assert np.array_equal(a, b), "uniform() decomposition assumption broken"


def main():
    check_fake_uniform_is_faithful()
    only = [a for a in sys.argv[1:] if not a.startswith("--")]     # optional: names of the fixtures to (re)generate
    for name, cfg in scenarios.REPLAY_CONFIGS.items():
        if only and name not in only:
            continue
        capacity, history, n, discount, omega, _ = cfg
        trace = scenarios.replay_scenario(ReferenceReplayAdapter(capacity, history, n, discount, omega), name)
        path = os.path.join(HERE, "replay_%s.npz" % name)
        np.savez_compressed(path, **trace)
        print("wrote", path, os.path.getsize(path), "bytes,", len(trace), "arrays")
    if "--replay-only" in sys.argv:
        return
    try:
        import make_golden_learn
    except ImportError:
        return
    make_golden_learn.main()


if __name__ == "__main__":
    main()
