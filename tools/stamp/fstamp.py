# in-kernel timeline of k_conv_fwd_full (RB_STAMP build): us since kernel start, wave 0 of workgroup 0
import os, sys, types, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from rainbow_amd import _lib as L
from rainbow_amd.agent import Agent
from rainbow_amd.memory import ReplayMemory
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["breakout-canonical-b256"]); cfg["capacity"] = 100000
args = bench.make_args(cfg, dev)
env = types.SimpleNamespace(action_space=lambda: cfg["actions"])
agent = Agent(args, env)
mem = ReplayMemory(args, cfg["capacity"], seed=7)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
lib = L.load()
lib.rb_debug_cstamps.argtypes = [C.c_void_p]
acc = []
for it in range(60):
    agent.reset_noise(); agent.learn(mem)
    torch.cuda.synchronize()
    st = (C.c_longlong * 64)()
    lib.rb_debug_cstamps(st)
    acc.append(list(st))
a = np.array(acc[20:], dtype=np.float64) * 0.01
m = np.median(a, axis=0)
order = [(57, "img0 before commit (weights staged, loads issued)"), (59, "img0 committed"), (49, "img0 barrier"), (50, "img0 mfma done (wave 0)"),
         (51, "img0 epilogue"), (52, "img0 end barrier"), (58, "img1 before commit"), (60, "img1 committed"), (53, "img1 barrier"),
         (54, "img1 mfma done"), (55, "img1 epilogue"), (56, "img1 end barrier")]
prev = m[48]
for i, name in order:
    print("%-52s t=%7.2f  (+%.2f)" % (name, m[i] - m[48], m[i] - prev)); prev = m[i]
