"""Soak test of the drop-in classes in the shape of the reference's training loop (main.py:147-179) with a synthetic
environment: act -> append every step (ring wraps several times), beta annealing, reset_noise + learn every 4 steps,
target sync every 2000, an eval-mode pass now and then.  Checks after every block of steps that the loss is finite,
the sampler never exhausted its attempts, the sum-tree root equals the sum of the leaves and the maximum priority is
sane.  Prints one JSON line."""
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    import __graft_entry__
    __graft_entry__.build()
    from rainbow_amd.agent import Agent
    from rainbow_amd.memory import ReplayMemory
    steps = int(os.environ.get("SOAK_STEPS", "60000"))
    cap = int(os.environ.get("SOAK_CAPACITY", "20000"))
    dev = torch.device("cuda", 0)
    cfg = dict(bench.CONFIGS[os.environ.get("SOAK_CONFIG", "pong-canonical-b32")]); cfg["capacity"] = cap
    args = bench.make_args(cfg, dev)
    env = types.SimpleNamespace(action_space=lambda: cfg["actions"])
    torch.manual_seed(3); np.random.seed(3)
    agent = Agent(args, env)
    mem = ReplayMemory(args, cap, seed=11)
    pool = torch.randint(0, 256, (64, 4, 84, 84), device=dev).float().div_(255)
    rs = np.random.RandomState(5)
    learn_start, T = 1600, steps
    beta0 = 0.4
    t0 = time.perf_counter()
    learns = 0
    checks = []
    for t in range(1, T + 1):
        state = pool[t & 63]
        if t % 4 == 0:
            agent.reset_noise()
        a = agent.act(state)
        assert 0 <= a < cfg["actions"]
        reward = float(rs.choice([-1.0, 0.0, 1.0], p=[0.05, 0.9, 0.05]))
        done = bool(rs.random_sample() < 1e-3)
        mem.append(state, a, reward, done)
        if t >= learn_start:
            mem.priority_weight = min(beta0 + (1 - beta0) * (t - learn_start) / (T - learn_start), 1.0)   # main.py:161
            if t % 4 == 0:
                agent.learn(mem)
                learns += 1
            if t % 2000 == 0:
                agent.update_target_net()
        if t % 10000 == 0:
            torch.cuda.synchronize(dev)
            hdr = mem._header()
            d = mem._dump()
            tree = d["tree"]
            ts = (1 << int(cap - 1).bit_length()) - 1
            leaves = tree[ts:ts + cap].astype(np.float64).sum()
            loss = agent._loss.detach().cpu().numpy()
            ok = bool(np.isfinite(loss).all()) and hdr.last_status == 0 and abs(float(tree[0]) - leaves) <= 1e-3 * max(1.0, leaves) \
                and np.isfinite(float(hdr.max)) and float(hdr.max) >= 1.0
            agent.eval(); q = agent.evaluate_q(state); agent.train()
            checks.append(dict(t=t, ok=ok, loss_mean=float(loss.mean()), root=float(tree[0]), max_p=float(hdr.max),
                               attempts=int(hdr.last_attempts), q=float(q)))
            assert ok and np.isfinite(q), checks[-1]
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(env_steps=T, learn_steps=learns, seconds=round(dt, 1), env_steps_per_s=round(T / dt, 1),
                          ring_wraps=round(T / cap, 1), checks=checks[-3:])))


if __name__ == "__main__":
    main()
