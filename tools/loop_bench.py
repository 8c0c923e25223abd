"""the ACTING side of the reference's training loop (main.py:147-179) with a synthetic
environment — per env step: dqn.act(state) (single un-batched state, device-to-host action) and mem.append(...);
every `replay_frequency` (4) steps: dqn.reset_noise(); dqn.learn(mem).  Prints one JSON line with the latency of
each piece and the env-steps/s of the whole loop.  Not the headline metric (bench.py is); it tells which piece
limits wall-clock training once the learn step is sub-millisecond."""
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


def timed(fn, n, dev):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n * 1e6


def main():
    import __graft_entry__
    __graft_entry__.build()
    from rainbow_amd.agent import Agent
    from rainbow_amd.memory import ReplayMemory
    dev = torch.device("cuda", 0)
    cfg = dict(bench.CONFIGS["pong-canonical-b32"])
    cfg["capacity"] = int(os.environ.get("LOOP_CAPACITY", "200000"))
    args = bench.make_args(cfg, dev)
    env = types.SimpleNamespace(action_space=lambda: cfg["actions"])
    agent = Agent(args, env)
    mem = ReplayMemory(args, cfg["capacity"], seed=7)
    bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
    states = [torch.rand(4, 84, 84, device=dev) for _ in range(16)]
    out = {}
    for _ in range(50):
        agent.act(states[0])
    k = [0]

    def act():
        k[0] += 1
        return agent.act(states[k[0] & 15])

    def append():
        k[0] += 1
        mem.append(states[k[0] & 15], 1, 0.0, False)

    def learn():
        agent.reset_noise()
        agent.learn(mem)

    out["act_us"] = timed(act, 2000, dev)
    if os.environ.get("LOOP_ONLY_ACT") == "1":
        print(json.dumps(out))
        return
    out["append_us"] = timed(append, 2000, dev)
    for _ in range(100):
        learn()
    out["learn_us"] = timed(learn, 1000, dev)

    def loop_iter():     # main.py:150-164, one learn period
        agent.reset_noise()
        for _ in range(4):
            k[0] += 1
            a = agent.act(states[k[0] & 15])
            mem.append(states[k[0] & 15], a, 0.0, False)
        agent.learn(mem)

    for _ in range(50):
        loop_iter()
    per = timed(loop_iter, 500, dev)
    out["loop_period_us"] = per
    out["env_steps_per_s"] = 4e6 / per
    if hasattr(agent, "act_batch"):
        for n in (16, 64, 256):
            sb = torch.rand(n, 4, 84, 84, device=dev)
            for _ in range(20):
                agent.act_batch(sb)
            t = timed(lambda: agent.act_batch(sb), 500, dev)
            out["act_batch%d_us" % n] = t
            out["act_batch%d_states_per_s" % n] = n * 1e6 / t
    print(json.dumps(out))


if __name__ == "__main__":
    main()
