# in-kernel timeline of k_conv_fwd_multi (RB_STAMP build, RB_MSTAMP_KS selects the layer by kernel size): us since kernel start
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
names = {49: "img0 staged(+weights)", 50: "img0 mfma", 51: "img0 tile0 reduced", 52: "img0 done", 53: "img1 staged", 54: "img1 mfma",
         55: "img1 tile0 reduced", 56: "img1 done", 57: "last staged", 58: "last mfma", 59: "last tile0", 60: "last done", 61: "end"}
prev = m[48]
for i in range(49, 62):
    print("%-24s t=%7.2f  (+%.2f)" % (names[i], m[i] - m[48], m[i] - prev)); prev = m[i]
