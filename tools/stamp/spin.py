import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
import torch
x = torch.zeros(16, device="cuda"); torch.cuda.synchronize()
lib = C.CDLL(os.environ["RAINBOW_AMD_LIB"])
lib.rb_debug_spin_launch.argtypes = [C.c_void_p, C.c_int, C.c_int]
print("-- torch CUDA context live"); sys.stdout.flush()
lib.rb_debug_spin_launch(None, 2000, 14)
st = torch.cuda.Stream()
lib.rb_debug_spin_launch(C.c_void_p(st.cuda_stream), 2000, 14)
# with the agent constructed (library handles, side streams, pinned host mirrors)
import types, bench
from rainbow_amd.agent import Agent
from rainbow_amd.memory import ReplayMemory
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["pong-canonical-b32"]); cfg["capacity"] = 100000
args = bench.make_args(cfg, dev)
agent = Agent(args, types.SimpleNamespace(action_space=lambda: cfg["actions"]))
print("-- agent constructed"); sys.stdout.flush()
lib.rb_debug_spin_launch(None, 2000, 14)
mem = ReplayMemory(args, cfg["capacity"], seed=7)
print("-- replay constructed"); sys.stdout.flush()
lib.rb_debug_spin_launch(None, 2000, 14)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
for _ in range(50): agent.learn(mem)
torch.cuda.synchronize()
print("-- after learn steps"); sys.stdout.flush()
lib.rb_debug_spin_launch(None, 2000, 14)
