"""Where the host time of Agent.act goes: the C call alone (asynchronous), the call + stream synchronize, the whole Agent.act."""
import os, sys, time, types
import torch
sys.path.insert(0, os.getcwd())
import bench
from rainbow_amd import _lib as L
from rainbow_amd.agent import Agent
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["pong-canonical-b32"])
agent = Agent(bench.make_args(cfg, dev), types.SimpleNamespace(action_space=lambda: cfg["actions"]))
st = torch.rand(4, 84, 84, device=dev)
for _ in range(100):
    agent.act(st)
lib, h, s = agent._lib, agent._h, torch.cuda.current_stream(dev)
ap, qp, sp = agent._act_pin.data_ptr(), agent._q_pin.data_ptr(), st.data_ptr()
N = 3000
def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / N * 1e6
print("C call only (async, back to back): %.1f us" % timed(lambda: lib.rb_learner_act(h, sp, 1, ap, qp, s.cuda_stream)))
def call_sync():
    lib.rb_learner_act(h, sp, 1, ap, qp, s.cuda_stream)
    s.synchronize()
print("C call + stream.synchronize(): %.1f us" % timed(call_sync))
print("Agent.act: %.1f us" % timed(lambda: agent.act(st)))
import numpy as np
def call_poll():
    agent._act_np[0] = -7
    lib.rb_learner_act(h, sp, 1, ap, qp, s.cuda_stream)
    a = agent._act_np
    while a[0] == -7:
        pass
print("C call + poll of the pinned action word: %.1f us" % timed(call_poll))
