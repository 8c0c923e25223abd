# does the launch of ONE of our kernels, repeated, throttle the host?  (enqueue time per call vs wall time per call)
import os, sys, time, types
import torch
sys.path.insert(0, os.getcwd())
import bench
from rainbow_amd.agent import Agent
from rainbow_amd.memory import ReplayMemory
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["pong-canonical-b32"]); cfg["capacity"] = 100000
args = bench.make_args(cfg, dev)
agent = Agent(args, types.SimpleNamespace(action_space=lambda: cfg["actions"]))
mem = ReplayMemory(args, cfg["capacity"], seed=7)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
for _ in range(20): agent.learn(mem)
torch.cuda.synchronize()
o = mem.sample_device(32)
loss = torch.rand(32, device=dev) + 0.1
lib, st = agent._lib, agent._stream()
def timeit(name, f, n=3000):
    for _ in range(50): f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): f()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%-46s enqueue %6.2f us/call, wall %6.2f us/call" % (name, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6)); sys.stdout.flush()
ip, lp = o["tree_idxs"].data_ptr(), loss.data_ptr()
timeit("k_update (rb_replay_update_priorities)", lambda: lib.rb_replay_update_priorities(mem._h, ip, lp, 32, st))
timeit("k_noise (rb_learner_reset_noise)", lambda: lib.rb_learner_reset_noise(agent._h, 1, None, st))
timeit("k_sample (sample_device)", lambda: mem.sample_device(32, gather=False, stream=st))
x = torch.zeros(1 << 22, device=dev)
timeit("torch add_ (16 MB)", lambda: x.add_(1.0))
norm = torch.zeros(1, device=dev)
timeit("k_clip_grad (rb_learner_clip_grad)", lambda: lib.rb_learner_clip_grad(agent._h, 10.0, norm.data_ptr(), st))
def pair():
    lib.rb_replay_update_priorities(mem._h, ip, lp, 32, st); lib.rb_learner_reset_noise(agent._h, 1, None, st)
timeit("k_update + k_noise alternating (per pair)", pair)
def pair2():
    mem.sample_device(32, gather=False, stream=st); lib.rb_replay_update_priorities(mem._h, ip, lp, 32, st)
timeit("k_sample + k_update alternating (per pair)", pair2)
def trio():
    mem.sample_device(32, gather=False, stream=st); lib.rb_replay_update_priorities(mem._h, ip, lp, 32, st); lib.rb_learner_clip_grad(agent._h, 10.0, norm.data_ptr(), st)
timeit("k_sample + k_update + clip_grad (per trio)", trio)
timeit("whole learn step", lambda: agent.learn(mem), 1500)
