# Diagnostic (needs a -DRB_STAMP build: bash tools/build_variant.sh stamp -DRB_STAMP; RAINBOW_AMD_LIB=...): the sampler
# workgroup's stage timeline inside the learn loop, with and without the hosted optimiser pass beside it.
import os, sys, types, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
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
lib = L.load()
lib.rb_debug_stamps.argtypes = [C.c_void_p]
acc = []
for it in range(80):
    agent.reset_noise()
    agent.learn(mem)
    if it >= 20 and it % 4 == 0:
        torch.cuda.synchronize()
        st = (C.c_longlong * 32)()
        lib.rb_debug_stamps(st)
        acc.append([st[i] - st[0] for i in range(6)])
a = np.array(acc, dtype=np.float64) * 0.01   # 100 MHz -> us
print("defer=%s  sampler stamps us (median): stage-top %.2f  descent+valid %.2f  after-loop %.2f  window/scalars %.2f  end %.2f"
      % ((agent._defer_update,) + tuple(np.median(a, axis=0)[1:])))
