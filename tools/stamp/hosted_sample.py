# in-kernel timestamps of the sampler workgroup INSIDE the learn step (hosted optimiser pass beside it): RB_STAMP build
#   bash tools/build_variant.sh stamp -DRB_STAMP; RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so python tools/stamp/hosted_sample.py <config>
import os, sys, types, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from rainbow_amd import _lib as L
from rainbow_amd.agent import Agent
from rainbow_amd.memory import ReplayMemory
dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "breakout-canonical-b256"
cfg = dict(bench.CONFIGS[name])
args = bench.make_args(cfg, dev)
env = types.SimpleNamespace(action_space=lambda: cfg["actions"])
np.random.seed(123); torch.manual_seed(123)
agent = Agent(args, env)
mem = ReplayMemory(args, cfg["capacity"], seed=1000)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
lib = L.load()
lib.rb_debug_stamps.argtypes = [C.c_void_p]
acc = []
for it in range(80):
    for _ in range(4):
        agent.reset_noise(); agent.learn(mem)
    torch.cuda.synchronize()
    st = (C.c_longlong * 32)()
    lib.rb_debug_stamps(st)
    acc.append([st[i] - st[0] for i in range(9)])
a = np.array(acc[10:], dtype=np.float64) * 0.01   # 100 MHz -> us
print(name, "hosted sampler workgroup, us from its start (median): top staged %.2f  descent+valid %.2f  after-loop %.2f  window/scalars %.2f  end %.2f" % tuple(np.median(a, axis=0)[1:6]))
m = np.median(a, axis=0)
print(name, "hosted optimiser workgroups, us from the sampler workgroup's start (median): first starts %.2f  last starts %.2f  last one ends %.2f" % (m[6], m[7], m[8]))
print(name, "p90: top staged %.2f  descent+valid %.2f  after-loop %.2f  window/scalars %.2f  end %.2f" % tuple(np.percentile(a, 90, axis=0)[1:6]))
