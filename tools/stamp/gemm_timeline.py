"""Per-workgroup timeline of the hidden layer's tiled GEMM launches (fc_gemm.h) at batch 256, from an RB_STAMP build:
bash tools/build_variant.sh stamp -DRB_STAMP; RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so python tools/stamp/gemm_timeline.py"""
import ctypes as C
import os
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.getcwd())
import bench  # noqa: E402
from rainbow_amd import _lib as L  # noqa: E402
from rainbow_amd.agent import Agent  # noqa: E402
from rainbow_amd.memory import ReplayMemory  # noqa: E402

cfgname = sys.argv[1] if len(sys.argv) > 1 else "breakout-canonical-b256"
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[cfgname])
cfg["capacity"] = 100000
args = bench.make_args(cfg, dev)
env = types.SimpleNamespace(action_space=lambda: cfg["actions"])
agent = Agent(args, env)
mem = ReplayMemory(args, cfg["capacity"], seed=7)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
lib = L.load()
K, W = 14, 2048
buf = (C.c_longlong * (K * W * 8))()
lib.rb_debug_wgtrace.argtypes = [C.c_void_p, C.c_int]
for it in range(20):
    agent.reset_noise()
    agent.learn(mem)
torch.cuda.synchronize()
lib.rb_debug_wgtrace(buf, 1)
agent.reset_noise()
agent.learn(mem)
torch.cuda.synchronize()
lib.rb_debug_wgtrace(buf, 0)
a = np.frombuffer(buf, dtype=np.int64).reshape(K, W, 8).astype(np.float64)
us = 0.01


def cu_of(rows):
    hw = rows[:, 7].astype(np.int64)
    return (hw >> 16) * 4096 + ((hw >> 8) & 0xff)


rows = a[12][a[12][:, 0] > 0]
if len(rows):
    t0 = rows[:, 0].min()
    print("== k_fc_gemm_fwd: %d workgroups, span %.2f us; starts last +%.2f" % (len(rows), (rows[:, 6].max() - t0) * us, (rows[:, 0].max() - t0) * us))
    uniq, cnt = np.unique(cu_of(rows), return_counts=True)
    print("   %d distinct CUs, max %d per CU" % (len(uniq), cnt.max()))
    d = np.diff(rows[:, [0, 2, 3, 4]], axis=1) * us
    print("   all: prologue %.2f | loop %.2f | partial store + arrive %.2f (medians); loop p90 %.2f max %.2f; loop done at median +%.2f last +%.2f"
          % (np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), np.percentile(d[:, 1], 90), d[:, 1].max(),
             np.median(rows[:, 3] - t0) * us, (rows[:, 3].max() - t0) * us))
    last = rows[rows[:, 1] == 1]
    if len(last):
        print("   last arrivers (%d): arrive at median +%.2f last +%.2f | sum partials %.2f | epilogue %.2f | end median +%.2f last +%.2f"
              % (len(last), np.median(last[:, 4] - t0) * us, (last[:, 4].max() - t0) * us, np.median(last[:, 5] - last[:, 4]) * us,
                 np.median(last[:, 6] - last[:, 5]) * us, np.median(last[:, 6] - t0) * us, (last[:, 6].max() - t0) * us))
rows = a[13][a[13][:, 0] > 0]
if len(rows):
    t0 = rows[:, 0].min()
    print("== k_fc_gemm_bwd: %d workgroups, span %.2f us" % (len(rows), (rows[:, 6].max() - t0) * us))
    cu = cu_of(rows)
    uniq, cnt = np.unique(cu, return_counts=True)
    print("   %d distinct CUs, workgroups per CU: max %d mean %.2f" % (len(uniq), cnt.max(), cnt.mean()))
    for role, rn in ((0, "write-back / padding"), (1, "dW"), (2, "dX")):
        r = rows[(rows[:, 1] == role) & (rows[:, 6] > 0)]
        if not len(r):
            continue
        line = "   %-20s n %4d  start median +%.2f last +%.2f | end median +%.2f last +%.2f | duration median %.2f max %.2f" % (
            rn, len(r), np.median(r[:, 0] - t0) * us, (r[:, 0].max() - t0) * us, np.median(r[:, 6] - t0) * us, (r[:, 6].max() - t0) * us,
            np.median(r[:, 6] - r[:, 0]) * us, (r[:, 6] - r[:, 0]).max() * us)
        rr = r[r[:, 2] > 0]
        if len(rr):
            d = np.diff(rr[:, [0, 2, 3, 6]], axis=1) * us
            line += " | prologue %.2f loop %.2f epilogue %.2f (medians), loop max %.2f epilogue max %.2f" % (
                np.median(d[:, 0]), np.median(d[:, 1]), np.median(d[:, 2]), d[:, 1].max(), d[:, 2].max())
        print(line)
