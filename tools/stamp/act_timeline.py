"""RB_STAMP build: per-phase timeline of the one-launch act path (act_path.h k_act_fused): for every phase, when the
workgroups are past its wait and when they finish it, relative to the launch's first start."""
import ctypes as C, os, sys, types
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from rainbow_amd import _lib as L
from rainbow_amd.agent import Agent
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS["pong-canonical-b32"])
args = bench.make_args(cfg, dev)
agent = Agent(args, types.SimpleNamespace(action_space=lambda: cfg["actions"]))
lib = L.load()
K, W = 14, 2048
buf = (C.c_longlong * (K * W * 8))()
lib.rb_debug_wgtrace.argtypes = [C.c_void_p, C.c_int]
st = torch.rand(4, 84, 84, device=dev)
for _ in range(30):
    agent.act(st)
lib.rb_debug_wgtrace(buf, 1)
agent.act(st)
lib.rb_debug_wgtrace(buf, 0)
a = np.frombuffer(buf, dtype=np.int64).reshape(K, W, 8).astype(np.float64)
done, past = a[10], a[11]
sel = done[:, 0] > 0
t0 = done[sel, 0].min()
print("workgroups %d, starts: median +%.2f last +%.2f us" % (sel.sum(), np.median(done[sel, 0] - t0) * 0.01, (done[sel, 0].max() - t0) * 0.01))
# only the workgroups WITH work in a phase wait for / arrive at its boundaries (round 6): the canonical stack's units per phase
units = [50, 24, 16, 256, (cfg["atoms"] * (cfg["actions"] + 1) + 3) // 4 if "atoms" in cfg else 90, 1]
for ph, nm in enumerate(["conv1", "conv2", "conv3", "fc_h", "fc_z", "head"]):
    w = np.arange(W) < units[ph]
    d, p = done[sel & w, 1 + ph], past[sel & w, 1 + ph]
    d, p = d[d > 0], p[p > 0]
    if len(d):
        print("%-6s past the wait: first +%.2f median +%.2f last +%.2f | phase done: first +%.2f median +%.2f last +%.2f"
              % (nm, (p.min() - t0) * 0.01, np.median(p - t0) * 0.01, (p.max() - t0) * 0.01, (d.min() - t0) * 0.01, np.median(d - t0) * 0.01, (d.max() - t0) * 0.01))
