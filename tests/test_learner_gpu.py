"""GPU parity tests of the learn step through the C ABI of librainbow_hip.so against the REAL
reference's golden vectors, and of the drop-in Python classes against the CPU oracle."""
import types

import numpy as np
import pytest
import torch

import scenarios
from helpers import assert_learn_trace_matches, load_golden
from oracle import learner_oracle as O
from oracle.replay_oracle import ReplayOracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from rainbow_amd import _lib
    return _lib.load()


@pytest.mark.parametrize("name", sorted(scenarios.LEARN_CONFIGS))
def test_learn_step_hip_matches_reference_golden(hip, name):
    from cabi_adapter import CAbiLearnAdapter, TorchMem
    ad = CAbiLearnAdapter(hip, TorchMem(), name)
    trace = scenarios.learn_scenario(ad, name, O)
    assert_learn_trace_matches(trace, load_golden("learn_%s.npz" % name), label="hip/" + name)
    ad.close()


def test_intermediates_match_oracle(hip):
    """m (projection), a* (double-Q argmax), pns_a and log p(s,a) against the oracle on the corner-case batch."""
    from cabi_adapter import CAbiLearnAdapter, TorchMem
    name = "canon"
    c = scenarios.LEARN_CONFIGS[name]
    cfg = O.Config(**c)
    ad = CAbiLearnAdapter(hip, TorchMem(), name)
    online, target = O.init_params(cfg, 77), O.init_params(cfg, 78)
    ad.load(online, target)
    rs = np.random.RandomState(5)
    draws = O.noise_draw_count(cfg)
    raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
    ad.reset_noise_online(raw_on)
    batch = scenarios.make_batch(c, 123)
    ad.learn_step(batch, raw_tg)
    want = O.learn(cfg, online, target, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg), batch)
    B, Z = c["batch"], c["atoms"]
    assert np.array_equal(ad.debug(2, (B,), np.int32), want["a_star"].astype(np.int32))
    np.testing.assert_allclose(ad.debug(1, (B, Z), np.float32), want["m"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(ad.debug(1, (B, Z), np.float32).sum(1), 1.0, rtol=1e-5)
    np.testing.assert_allclose(ad.debug(3, (B, Z), np.float32), want["pns_a"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(ad.debug(0, (B, Z), np.float32), want["log_ps_a"], rtol=1e-5, atol=1e-5)
    ad.close()


def _args(**kw):
    base = dict(device=torch.device("cuda:0"), history_length=4, discount=0.99, multi_step=3, priority_weight=0.4,
                priority_exponent=0.5, atoms=51, V_min=-10.0, V_max=10.0, batch_size=8, norm_clip=10.0, model=None,
                learning_rate=6.25e-5, adam_eps=1.5e-4, architecture="canonical", hidden_size=64, noisy_std=0.1)
    base.update(kw)
    return types.SimpleNamespace(**base)


def test_agent_and_memory_classes_end_to_end_vs_oracle(hip, tmp_path):
    """The drop-in classes (Agent.learn(mem) on the device-resident fast path) against oracle replay + oracle
    learner for several consecutive steps with injected sampler uniforms and noise; then save/load."""
    from rainbow_amd.agent import Agent
    from rainbow_amd.memory import ReplayMemory
    args = _args()
    A, B, cap = 6, args.batch_size, 1024
    env = types.SimpleNamespace(action_space=lambda: A)
    torch.manual_seed(3)
    agent = Agent(args, env)
    mem = ReplayMemory(args, cap, seed=11)
    ora_mem = ReplayOracle(cap)
    rs = np.random.RandomState(21)
    for _ in range(1500):
        st = rs.randint(0, 256, size=(4, 84, 84)).astype(np.float32) / np.float32(255)
        a, r, term = int(rs.randint(0, A)), float(rs.choice([-1.0, 0.0, 1.0])), bool(rs.random_sample() < 0.02)
        mem.append(torch.from_numpy(st).cuda(), a, r, term)
        ora_mem.append(st, a, r, term)
    cfg = O.Config(batch=B, atoms=51, actions=A, history=4, hidden=64, architecture="canonical", multi_step=3)
    online = {k: v.cpu().numpy() for k, v in agent.state_dict().items() if "epsilon" not in k}
    target = {k: v.copy() for k, v in online.items()}
    adam = O.AdamOracle(online, args.learning_rate, args.adam_eps)
    draws = O.noise_draw_count(cfg)
    for step in range(4):
        raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
        uu = rs.random_sample((32, B))
        beta = 0.4 + 0.1 * step
        mem.priority_weight = beta
        ora_mem.priority_weight = beta
        agent.reset_noise(torch.from_numpy(raw_on))
        agent.learn(mem, _target_raw_normals=torch.from_numpy(raw_tg), _unit_uniforms=torch.from_numpy(uu))
        batch = ora_mem.sample_with_uniforms(B, uu)
        want = O.learn(cfg, online, target, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg), batch)
        total, clipped = O.clip_grads(want["grads"], args.norm_clip)
        online = adam.step(clipped)
        ora_mem.update_priorities(batch["tree_idxs"], want["loss"])
        torch.cuda.synchronize()
        assert np.array_equal(mem._out[B]["tree_idxs"].cpu().numpy(), batch["tree_idxs"]), "step %d" % step
        np.testing.assert_allclose(agent._loss.cpu().numpy(), want["loss"], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(float(agent._norm.item()), total, rtol=2e-5)
        got = {k: v.cpu().numpy() for k, v in agent.state_dict().items() if "epsilon" not in k}
        for k in online:
            np.testing.assert_allclose(got[k], online[k], rtol=0, atol=3e-7, err_msg="step %d %s" % (step, k))
        if step == 1:
            agent.update_target_net()
            target = {k: v.copy() for k, v in online.items()}
    # tree after 4 priority updates matches the oracle's to ~1 ulp (loss^0.5 is a float32 pow)
    np.testing.assert_allclose(mem._dump()["tree"], ora_mem.transitions.tree, rtol=2e-5)
    # act / evaluate_q, train vs eval mode
    st = rs.randint(0, 256, size=(4, 84, 84)).astype(np.float32) / np.float32(255)
    noise_on = O.make_noise(cfg, raw_on)
    a_want, q_want = O.act(cfg, online, noise_on, st)
    assert agent.act(torch.from_numpy(st).cuda()) == a_want
    np.testing.assert_allclose(agent.evaluate_q(torch.from_numpy(st).cuda()), q_want, rtol=2e-5)
    agent.eval()
    a_want, q_want = O.act(cfg, online, None, st)
    assert agent.act(torch.from_numpy(st).cuda()) == a_want
    np.testing.assert_allclose(agent.evaluate_q(torch.from_numpy(st).cuda()), q_want, rtol=2e-5)
    agent.train()
    # vectorised actors: act_batch(states)[i] == act(states[i]) (batched forward through the training kernels)
    sts = torch.from_numpy(rs.randint(0, 256, size=(5, 4, 84, 84)).astype(np.float32) / np.float32(255)).cuda()
    assert list(agent.act_batch(sts)) == [agent.act(s) for s in sts]
    many = sts.repeat(4, 1, 1, 1)[:19]                       # more than 2 * batch_size states: processed in chunks
    assert list(agent.act_batch(many)) == [agent.act(s) for s in many]
    # checkpoint interchange: reference key names, round trip (agent.py:26-36,106-107)
    agent.save(str(tmp_path), "model.pth")
    sd = torch.load(str(tmp_path / "model.pth"), map_location="cpu")
    assert list(sd.keys())[:8] == ["convs.0.weight", "convs.0.bias", "convs.2.weight", "convs.2.bias", "convs.4.weight",
                                   "convs.4.bias", "fc_h_v.weight_mu", "fc_h_v.weight_sigma"]
    assert tuple(sd["fc_h_v.weight_epsilon"].shape) == (64, 3136) and tuple(sd["fc_z_a.bias_epsilon"].shape) == (A * 51,)
    args2 = _args(model=str(tmp_path / "model.pth"))
    agent2 = Agent(args2, env)
    for k, v in agent2.state_dict().items():
        if "epsilon" not in k:
            assert torch.equal(v.cpu(), sd[k]), k


def test_compat_path_with_foreign_replay(hip):
    """Agent.learn(mem) with a replay object that only offers the reference's sample()/update_priorities()."""
    from rainbow_amd.agent import Agent
    args = _args(architecture="data-efficient", hidden_size=32, batch_size=6)
    env = types.SimpleNamespace(action_space=lambda: 4)
    agent = Agent(args, env)
    c = dict(scenarios.LEARN_CONFIGS["dataeff"])
    batch = scenarios.make_batch(c, 5)

    class Foreign:
        def sample(self, B):
            f = lambda a: torch.from_numpy(a).to(torch.float32).div_(255).cuda()
            return (np.arange(B), f(batch["states"]), torch.from_numpy(batch["actions"]).cuda(),
                    torch.from_numpy(batch["returns"]).cuda(), f(batch["next_states"]),
                    torch.from_numpy(batch["nonterminals"]).reshape(-1, 1).cuda(), torch.from_numpy(batch["weights"]).cuda())

        def update_priorities(self, idxs, prios):
            self.got = (idxs, prios)

    fm = Foreign()
    agent.learn(fm)
    assert isinstance(fm.got[1], np.ndarray) and fm.got[1].shape == (6,) and np.all(np.isfinite(fm.got[1]))
    assert np.all(fm.got[1] > 0)


def test_graph_replay_matches_eager(hip, monkeypatch):
    """Agent.learn captured into a hipGraph must follow the eager trajectory (same device Philox streams)."""
    from rainbow_amd.agent import Agent
    from rainbow_amd.memory import ReplayMemory

    def run(graph):
        monkeypatch.setenv("RAINBOW_AMD_GRAPH", "1" if graph else "0")   # graph replay is opt-in
        args = _args(architecture="data-efficient", hidden_size=64, batch_size=16)
        env = types.SimpleNamespace(action_space=lambda: 4)
        torch.manual_seed(5)
        np.random.seed(5)
        agent = Agent(args, env)
        mem = ReplayMemory(args, 2048, seed=17)
        g = torch.Generator(device="cuda").manual_seed(1)
        rs = np.random.RandomState(1)
        for _ in range(2):
            mem.append_batch(torch.randint(0, 256, (1500, 84, 84), dtype=torch.uint8, device="cuda", generator=g),
                             rs.randint(0, 4, 1500), rs.choice([-1.0, 0.0, 1.0], size=1500), rs.random_sample(1500) < 0.01)
        losses = []
        for k in range(12):
            mem.priority_weight = min(1.0, 0.4 + 0.05 * k)      # annealed beta must reach the captured sampler
            agent.reset_noise()
            agent.learn(mem)
            losses.append(agent._loss.clone())
            if k == 6:
                agent.update_target_net()
        torch.cuda.synchronize()
        assert (agent._graph is not None) == graph
        return torch.stack(losses).cpu().numpy(), agent.params.detach().cpu().numpy(), mem._grab("tree")

    le, pe, te = run(False)
    lg, pg, tg = run(True)
    np.testing.assert_allclose(lg, le, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pg, pe, rtol=0, atol=2e-6)
    np.testing.assert_allclose(tg, te, rtol=1e-4)


@pytest.mark.parametrize("base", ["dataeff", "canon"])
def test_wide_batch_matches_oracle(hip, monkeypatch, base):
    """Batch 64 (128 online rows): the noisy-linear forward switches to 64-row m-chunks (k_nl_fwd2<0, 4>) and the conv
    weight-gradient workgroups sum two images each; loss and every gradient against the oracle on the same inputs."""
    from cabi_adapter import CAbiLearnAdapter, TorchMem
    cfgd = dict(scenarios.LEARN_CONFIGS[base], batch=64, multi_step=3)
    monkeypatch.setitem(scenarios.LEARN_CONFIGS, "wide", cfgd)
    cfg = O.Config(**cfgd)
    ad = CAbiLearnAdapter(hip, TorchMem(), "wide")
    online, target = O.init_params(cfg, 177), O.init_params(cfg, 178)
    ad.load(online, target)
    rs = np.random.RandomState(15)
    draws = O.noise_draw_count(cfg)
    raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
    ad.reset_noise_online(raw_on)
    batch = scenarios.make_batch(cfgd, 321)
    got = ad.learn_step(batch, raw_tg)
    want = O.learn(cfg, online, target, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg), batch)
    np.testing.assert_allclose(got["loss"], want["loss"], rtol=2e-5, atol=1e-6)
    for k, g in want["grads"].items():
        scale = float(np.max(np.abs(g))) if g.size else 0.0
        np.testing.assert_allclose(got["grads"][k], g, rtol=2e-4, atol=5e-6 * scale + 1e-9, err_msg=k)
    ad.close()
