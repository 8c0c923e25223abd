# host-side cost of one learn step: enqueue time of N steps (before the final sync) against their wall time, and a cProfile of the loop
import os, sys, time, types, cProfile, pstats
import torch
sys.path.insert(0, os.getcwd())
import bench
from rainbow_amd.agent import Agent
from rainbow_amd.memory import ReplayMemory
name = os.environ.get("HOST_CONFIG", "pong-canonical-b32")
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[name]); cfg["capacity"] = 100000
args = bench.make_args(cfg, dev)
env = types.SimpleNamespace(action_space=lambda: cfg["actions"])
agent = Agent(args, env)
mem = ReplayMemory(args, cfg["capacity"], seed=7)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
if os.environ.get("HOST_STREAM") == "1":      # run on a created stream instead of the (null) default stream
    _st = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    torch.cuda.set_stream(_st)
    print("running on stream", hex(_st.cuda_stream))
for _ in range(200):
    agent.learn(mem)
torch.cuda.synchronize()
N = 2000
t0 = time.perf_counter()
for _ in range(N):
    agent.learn(mem)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("%s: enqueue %.1f us/step, wall %.1f us/step" % (name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
# the same with an idle GPU in front of every call: the pure launch cost, no back-pressure from a busy queue
ts = []
for _ in range(300):
    torch.cuda.synchronize()
    t = time.perf_counter(); agent.learn(mem); ts.append(time.perf_counter() - t)
torch.cuda.synchronize()
ts.sort()
print("enqueue with an idle GPU: median %.1f us/step (min %.1f)" % (ts[len(ts) // 2] * 1e6, ts[0] * 1e6))
print({k: os.environ.get(k) for k in ("HIP_LAUNCH_BLOCKING", "AMD_SERIALIZE_KERNEL", "AMD_SERIALIZE_COPY", "GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG", "HSA_ENABLE_IPC_MODE_LEGACY", "AMD_LOG_LEVEL")})
# host time inside the three C entry points of a step
class _Timed:
    def __init__(self, lib, names):
        self._lib, self.t = lib, {n: 0.0 for n in names}
    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if name not in self.t:
            return f
        def g(*a):
            t = time.perf_counter(); r = f(*a); self.t[name] += time.perf_counter() - t; return r
        return g
tl = _Timed(agent._lib, ["rb_learner_learn_windows", "rb_learner_clip_adam", "rb_replay_sample_fused_noise", "rb_replay_failed_samples",
                        "rb_learner_priority_written"])
agent._lib = tl; mem._lib = tl
for _ in range(N):
    agent.learn(mem)
torch.cuda.synchronize()
print("host us/step inside C calls:", {k: round(v / N * 1e6, 1) for k, v in tl.t.items()})
agent._lib = tl._lib; mem._lib = tl._lib
if hasattr(agent._lib, "rb_debug_host_timing"):
    try:
        agent._lib.rb_debug_host_timing(1)
        for _ in range(N):
            agent.learn(mem)
        torch.cuda.synchronize()
        agent._lib.rb_debug_host_timing(0)
        sys.stdout.flush()
    except AttributeError:
        pass
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    agent.learn(mem)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
