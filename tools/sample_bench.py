"""Sampler-only timing on the GPU: back-to-back rb_replay_sample launches (no gather), with and without the learner's
noise job riding along, plus the stand-alone priority update.  Prints one JSON line (us per launch, stream time)."""
import json
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def stream_us(fn, n, dev):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize(dev)
    return e0.elapsed_time(e1) * 1e3 / n


def main():
    import __graft_entry__
    __graft_entry__.build()
    from rainbow_amd import _lib as L
    from rainbow_amd.agent import Agent
    from rainbow_amd.memory import ReplayMemory
    dev = torch.device("cuda", 0)
    cfg = dict(bench.CONFIGS[os.environ.get("SAMPLE_CONFIG", "pong-canonical-b32")])
    args = bench.make_args(cfg, dev)
    env = types.SimpleNamespace(action_space=lambda: cfg["actions"])
    agent = Agent(args, env)
    mem = ReplayMemory(args, cfg["capacity"], seed=7)
    bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
    B = cfg["batch_size"]
    out = {}
    out["sample_us"] = stream_us(lambda: mem.sample_device(B, gather=False), 2000, dev)
    out["sample_gather_us"] = stream_us(lambda: mem.sample_device(B, gather=True), 2000, dev)
    job = L.NoiseJob()
    L.check(agent._lib, agent._lib.rb_learner_noise_job(agent._h, 2, job))
    out["sample_noise2_us"] = stream_us(lambda: mem.sample_device(B, gather=False, noise_job=job), 2000, dev)
    o = mem.sample_device(B, gather=False)
    loss = torch.rand(B, device=dev) + 0.1
    out["update_us"] = stream_us(lambda: mem.update_priorities(o["tree_idxs"], loss), 2000, dev)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
