import os, sys, types, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from rainbow_amd import _lib as L
from rainbow_amd.agent import Agent
from rainbow_amd.memory import ReplayMemory
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["pong-canonical-b32"]); cfg["capacity"] = 100000
args = bench.make_args(cfg, dev)
env = types.SimpleNamespace(action_space=lambda: cfg["actions"])
agent = Agent(args, env)
mem = ReplayMemory(args, cfg["capacity"], seed=7)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
lib = L.load()
lib.rb_debug_cstamps.argtypes = [C.c_void_p]
acc = []
for it in range(80):
    agent.reset_noise(); agent.learn(mem)
    torch.cuda.synchronize()
    st = (C.c_longlong * 64)()
    lib.rb_debug_cstamps(st)
    acc.append(list(st))
a = np.array(acc[20:], dtype=np.float64) * 0.01
for name, sb in (("dw conv1", 24), ("dw conv2", 32), ("dw conv3", 40)):
    m = np.median(a, axis=0)
    print("%s block0: staged +%.2f  mfma(+bias) +%.2f  store +%.2f" % (name, m[sb+1]-m[sb], m[sb+2]-m[sb+1], m[sb+3]-m[sb+2]))
for name, sb in (("conv1", 0), ("conv2", 8), ("conv3", 16)):
    m = np.median(a, axis=0)
    print("%s block0: staged +%.2f  mfma +%.2f  epilogue +%.2f | last block starts +%.2f ends +%.2f (us after block 0 start)" % (
        name, m[sb+1]-m[sb], m[sb+2]-m[sb+1], m[sb+3]-m[sb+2], m[sb+4]-m[sb], m[sb+5]-m[sb]))
