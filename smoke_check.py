"""smoke(): ONE tiny learn step of the hot path on cuda:0 (replay sample -> learn -> clip -> Adam ->
priority update through librainbow_hip.so), checked against the CPU oracle.  The oracle is only
the checker here; the measured/shipped path never touches it.  Lives beside __graft_entry__.py, NOT inside the rainbow_amd
package: nothing under rainbow_amd/ imports the oracle."""
import os
import sys
import types

import numpy as np
import torch


def run(verbose=True):
    root = os.path.dirname(os.path.abspath(__file__))
    if root not in sys.path:
        sys.path.insert(0, root)
    from oracle import learner_oracle as O
    from oracle.replay_oracle import ReplayOracle
    from rainbow_amd.agent import Agent
    from rainbow_amd.memory import ReplayMemory

    assert torch.cuda.is_available(), "smoke() needs a GPU"
    dev = torch.device("cuda:0")
    B, cap = 4, 256
    args = types.SimpleNamespace(device=dev, history_length=4, discount=0.99, multi_step=3, priority_weight=0.4,
                                 priority_exponent=0.5, atoms=51, V_min=-10.0, V_max=10.0, batch_size=B,
                                 norm_clip=10.0, model=None, learning_rate=6.25e-5, adam_eps=1.5e-4,
                                 architecture="data-efficient", hidden_size=32, noisy_std=0.1)
    env = types.SimpleNamespace(action_space=lambda: 4)
    torch.manual_seed(0)
    agent = Agent(args, env)
    mem = ReplayMemory(args, cap, seed=1)
    ref = ReplayOracle(cap, history=4, discount=0.99, multi_step=3, priority_weight=0.4, priority_exponent=0.5)
    rs = np.random.RandomState(0)
    for _ in range(200):
        st = (rs.randint(0, 256, size=(4, 84, 84)).astype(np.float32) / np.float32(255))
        a, r, term = int(rs.randint(0, 4)), float(rs.choice([-1.0, 0.0, 1.0])), bool(rs.random_sample() < 0.05)
        mem.append(torch.from_numpy(st).to(dev), a, r, term)
        ref.append(st, a, r, term)
    cfg = O.Config(batch=B, atoms=51, actions=4, history=4, hidden=32, architecture="data-efficient", multi_step=3)
    online = {k: v.cpu().numpy() for k, v in agent.state_dict().items() if "epsilon" not in k}
    draws = O.noise_draw_count(cfg)
    raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
    uu = rs.random_sample((32, B))
    agent.reset_noise(torch.from_numpy(raw_on))
    agent.learn(mem, _target_raw_normals=torch.from_numpy(raw_tg), _unit_uniforms=torch.from_numpy(uu))
    torch.cuda.synchronize()
    loss = agent._loss.cpu().numpy()
    batch = ref.sample_with_uniforms(B, uu)
    want = O.learn(cfg, online, online, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg),
                   dict(states=batch["states"], next_states=batch["next_states"], actions=batch["actions"],
                        returns=batch["returns"], nonterminals=batch["nonterminals"], weights=batch["weights"]))
    idx = mem._out[B]["tree_idxs"].cpu().numpy()
    assert np.array_equal(idx, batch["tree_idxs"]), (idx, batch["tree_idxs"])
    np.testing.assert_allclose(loss, want["loss"], rtol=2e-5, atol=1e-6)
    if verbose:
        from rainbow_amd import _lib
        print("smoke ok: loss", loss, "grad_norm", float(agent._norm.item()), "library source hash", _lib.source_hash(_lib.load()))
