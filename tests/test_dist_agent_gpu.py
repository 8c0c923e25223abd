"""The class-level replica path on hardware: TWO processes on ONE MI355X (backend gloo — RCCL refuses two ranks per device;
gloo stages device tensors through the host), each with its own `Agent(args, env)` created under the initialised process
group and its own HBM replay with different contents.  This drives exactly what `bench.py --gpus N` / `torchrun main.py`
execute between /root/reference/agent.py:96 and :97: Agent.__init__'s rank-0 broadcast + FactoredExchange, and
Agent._learn_eager's exchange branch (`self._exchange.run()` / `average_gradients` + the deferred clip + Adam pass).
Three learn steps with injected noise and sampler uniforms, both exchange modes: the replicas' parameters must be
BIT-IDENTICAL, and equal to the oracle fed with the mean of the two replicas' gradients (tolerances of the one-device tests)."""
import os
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

HIDDEN, BATCH, CAP, APPENDS, ACTIONS, STEPS = 64, 8, 1024, 1500, 6, 3


def _args():
    return types.SimpleNamespace(device=torch.device("cuda:0"), history_length=4, discount=0.99, multi_step=3,
                                 priority_weight=0.4, priority_exponent=0.5, atoms=51, V_min=-10.0, V_max=10.0,
                                 batch_size=BATCH, norm_clip=10.0, model=None, learning_rate=6.25e-5, adam_eps=1.5e-4,
                                 architecture="canonical", hidden_size=HIDDEN, noisy_std=0.1)


def _transitions(rank):
    """The rank's replay contents (seeded): what the worker appends and what the checker feeds the oracle replay."""
    rs = np.random.RandomState(21 + 7 * rank)
    for _ in range(APPENDS):
        st = rs.randint(0, 256, size=(4, 84, 84)).astype(np.float32) / np.float32(255)
        yield st, int(rs.randint(0, ACTIONS)), float(rs.choice([-1.0, 0.0, 1.0])), bool(rs.random_sample() < 0.02)


def _randomness(rank, step, draws):
    rs = np.random.RandomState(1000 + 10 * step + rank)
    return rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32), rs.random_sample((32, BATCH))


def _worker(rank, world, port, outdir, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RAINBOW_AMD_EXCHANGE"] = mode
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import learner_oracle as O
    from rainbow_amd.agent import Agent
    from rainbow_amd.memory import ReplayMemory
    args = _args()
    env = types.SimpleNamespace(action_space=lambda: ACTIONS)
    torch.manual_seed(3 + 100 * rank)            # replicas START different on purpose: the constructor's broadcast must fix it
    agent = Agent(args, env)
    assert agent._dist and agent._world == world
    assert (agent._exchange is not None) == (mode == "factored")
    mem = ReplayMemory(args, CAP, seed=11 + rank)
    for st, a, r, term in _transitions(rank):
        mem.append(torch.from_numpy(st).cuda(), a, r, term)
    init = {k: v.cpu().numpy() for k, v in agent.state_dict().items() if "epsilon" not in k}
    cfg = O.Config(batch=BATCH, atoms=51, actions=ACTIONS, history=4, hidden=HIDDEN, architecture="canonical", multi_step=3)
    draws = O.noise_draw_count(cfg)
    out = {"init/" + k: v for k, v in init.items()}
    for step in range(STEPS):
        raw_on, raw_tg, uu = _randomness(rank, step, draws)
        agent.reset_noise(torch.from_numpy(raw_on))
        agent.learn(mem, _target_raw_normals=torch.from_numpy(raw_tg), _unit_uniforms=torch.from_numpy(uu))
        torch.cuda.synchronize()
        out["s%d_loss" % step] = agent._loss.cpu().numpy().copy()
        out["s%d_norm" % step] = np.float32(agent._norm.item())
        out["s%d_idx" % step] = mem._out[BATCH]["tree_idxs"].cpu().numpy().copy()
    for k, v in agent.state_dict().items():
        if "epsilon" not in k:
            out["final/" + k] = v.cpu().numpy()
    out["grads"] = agent.grads.detach().cpu().numpy()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["factored", "allreduce"])
def test_two_agents_on_one_device_stay_identical_and_match_the_oracle(tmp_path, mode):
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    world = 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(world)]
    names = [k[len("final/"):] for k in r[0].files if k.startswith("final/")]
    for n in names:      # identical after the broadcast, identical after three exchanged steps
        assert np.array_equal(r[0]["init/" + n], r[1]["init/" + n]), "initial broadcast: " + n
        assert np.array_equal(r[0]["final/" + n], r[1]["final/" + n]), "replicas diverged: " + n
    assert np.array_equal(r[0]["grads"], r[1]["grads"])
    for step in range(STEPS):
        assert r[0]["s%d_norm" % step] == r[1]["s%d_norm" % step]

    # oracle: the same three steps with the mean of the two replicas' gradients (each replica's own replay, noise, uniforms)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import learner_oracle as O
    from oracle.replay_oracle import ReplayOracle
    cfg = O.Config(batch=BATCH, atoms=51, actions=ACTIONS, history=4, hidden=HIDDEN, architecture="canonical", multi_step=3)
    a = _args()
    online = {n: r[0]["init/" + n].copy() for n in names}
    target = {k: v.copy() for k, v in online.items()}
    adam = O.AdamOracle(online, a.learning_rate, a.adam_eps)
    mems = []
    for rank in range(world):
        om = ReplayOracle(CAP)
        for st, act, rew, term in _transitions(rank):
            om.append(st, act, rew, term)
        mems.append(om)
    draws = O.noise_draw_count(cfg)
    for step in range(STEPS):
        gs = []
        for rank in range(world):
            raw_on, raw_tg, uu = _randomness(rank, step, draws)
            batch = mems[rank].sample_with_uniforms(BATCH, uu)
            assert np.array_equal(r[rank]["s%d_idx" % step], batch["tree_idxs"]), "rank %d step %d" % (rank, step)
            want = O.learn(cfg, online, target, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg), batch)
            np.testing.assert_allclose(r[rank]["s%d_loss" % step], want["loss"], rtol=2e-5, atol=1e-6)
            mems[rank].update_priorities(batch["tree_idxs"], want["loss"])
            gs.append(want["grads"])
        mean = {n: (gs[0][n] + gs[1][n]) / np.float32(2) for n in gs[0]}
        total, clipped = O.clip_grads(mean, a.norm_clip)
        online = adam.step(clipped)
        np.testing.assert_allclose(r[0]["s%d_norm" % step], total, rtol=2e-5)
    for n in names:
        np.testing.assert_allclose(r[0]["final/" + n], online[n], rtol=0, atol=3e-7, err_msg=n)


def test_bench_multi_rank_control_flow_rehearsal_on_one_gpu():
    """`bench.py --gpus N` has only ever run with ONE rank (no multi-GPU box in the pool), so its multi-rank
    branch — the torchrun respawn, process-group init, the barriers around the timed region, the MAX-reduced elapsed time,
    `value = world * steps / elapsed`, the Agent's exchange between backward and clip (agent.py:96-97) — is REHEARSED here:
    `--backend gloo` puts both ranks on cuda:0 (RCCL refuses a second rank per device; the replica exchange stages its blocks
    through the host).  Checks the one JSON line rank 0 prints: two ranks claimed, the value arithmetic, finite numbers, the
    rehearsal marked as such.  It measures NOTHING about scaling."""
    import json
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    env.pop("RB_OPTS", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "20", "--warmup", "5",
           "--capacity", "65536", "--no-cpu-baseline"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly ONE JSON line (rank 0): %r" % lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 20 and d["warmup"] == 5 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 64 and "REHEARSAL" in d["config"]["parallelism"] and "replicas x2" in d["config"]["parallelism"]
    assert np.isfinite(d["value"]) and d["value"] > 0
    np.testing.assert_allclose(d["value"], 2 * 20 / (d["ms_per_step"] * 1e-3 * 20), rtol=1e-9)     # whole-job aggregate over both ranks
    assert "roofline" in d and "cpu_baseline" not in d
