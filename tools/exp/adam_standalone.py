"""How fast does the paired optimiser pass stream as a launch of its OWN (k_adam_pending: every learn() followed by flush())?
Run under rocprofv3 --kernel-trace --stats; compare with the hosted form (k_sample<1024,4>) of a plain bench run."""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from rainbow_amd.agent import Agent
from rainbow_amd.memory import ReplayMemory
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["pong-canonical-b32"]); cfg["capacity"] = 100000
args = bench.make_args(cfg, dev)
agent = Agent(args, types.SimpleNamespace(action_space=lambda: cfg["actions"]))
mem = ReplayMemory(args, cfg["capacity"], seed=1)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
for _ in range(400):
    agent.reset_noise(); agent.learn(mem); agent.flush()
torch.cuda.synchronize()
