import os, sys, types, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from rainbow_amd import _lib as L
from rainbow_amd.agent import Agent
from rainbow_amd.memory import ReplayMemory
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[os.environ.get("CFG", "pong-canonical-b32")]); cfg["capacity"] = 100000
args = bench.make_args(cfg, dev)
env = types.SimpleNamespace(action_space=lambda: cfg["actions"])
agent = Agent(args, env)
mem = ReplayMemory(args, cfg["capacity"], seed=7)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
lib = L.load()
lib.rb_debug_spans.argtypes = [C.c_void_p, C.c_int]
st = (C.c_longlong * 64)()
acc = []
for it in range(100):
    lib.rb_debug_spans(st, 1)
    agent.reset_noise(); agent.learn(mem)
    torch.cuda.synchronize()
    lib.rb_debug_spans(st, 0)
    acc.append(list(st))
a = np.array(acc[20:], dtype=np.float64) * 0.01
names = {0: "z:update", 1: "z:dW", 2: "z:dX", 3: "h:update", 4: "h:dW", 5: "h:dX"}
m = np.median(a, axis=0)
for base, label in ((0, "fc_z bwd launch"), (3, "fc_h bwd launch")):
    t0 = min(m[2 * (base + k)] for k in range(3) if m[2 * (base + k) + 1] > 0)
    print(label)
    for k in range(3):
        s_, e_ = m[2 * (base + k)], m[2 * (base + k) + 1]
        if e_ > 0:
            print("   %-9s first start +%.2f us   last end +%.2f us" % (names[base + k], s_ - t0, e_ - t0))
