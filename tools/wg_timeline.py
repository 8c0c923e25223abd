"""Per-workgroup timeline of the conv launches of one learn step (RB_STAMP build: bash tools/build_variant.sh stamp -DRB_STAMP;
RAINBOW_AMD_LIB=$PWD/rainbow_amd/librainbow_hip_stamp.so python tools/wg_timeline.py [config]).  For every traced kernel:
when its workgroups start / finish relative to the first start, the median duration of each phase, how many CUs it used and
how many workgroups shared a CU."""
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

cfgname = sys.argv[1] if len(sys.argv) > 1 else "pong-canonical-b32"
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
for it in range(40):
    agent.reset_noise()
    agent.learn(mem)
torch.cuda.synchronize()
lib.rb_debug_wgtrace(buf, 1)
agent.reset_noise()
agent.learn(mem)
torch.cuda.synchronize()
lib.rb_debug_wgtrace(buf, 0)
a = np.frombuffer(buf, dtype=np.int64).reshape(K, W, 8).astype(np.float64)
names = ["conv1_fwd", "conv2_fwd", "conv3_fwd", "conv3_dx", "conv2_dx", "conv_dw", "(fine)", "head: logits loaded | double-Q | softmaxes | projection | loss | dlogits"]
phase = ["stage weights", "wait", "stage input", "mfma", "epilogue", "signal"]
t0_all = None
for k in range(8):
    rows = a[k][a[k][:, 0] > 0]
    if not len(rows):
        continue
    if t0_all is None:
        t0_all = rows[:, 0].min()
    t0 = rows[:, 0].min()
    us = lambda x: x * 0.01
    print("== %s: %d workgroups; first start at +%.2f us of the first traced kernel" % (names[k], len(rows), us(t0 - t0_all)))
    print("   starts: median +%.2f  p90 +%.2f  last +%.2f | ends: median +%.2f  last +%.2f us"
          % (us(np.median(rows[:, 0]) - t0), us(np.percentile(rows[:, 0], 90) - t0), us(rows[:, 0].max() - t0),
             us(np.median(rows[:, 6]) - t0), us(rows[:, 6].max() - t0)))
    d = np.diff(rows[:, 0:7], axis=1)
    print("   phases (median / p90 us): " + "  ".join("%s %.2f/%.2f" % (phase[i], us(np.median(d[:, i])), us(np.percentile(d[:, i], 90))) for i in range(6)))
    print("   workgroup total: median %.2f  p90 %.2f  max %.2f us" % (us(np.median(rows[:, 6] - rows[:, 0])), us(np.percentile(rows[:, 6] - rows[:, 0], 90)), us((rows[:, 6] - rows[:, 0]).max())))
    hw = rows[:, 7].astype(np.int64)
    cu = (hw >> 16) * 4096 + ((hw >> 8) & 0xff)          # XCC id, (se, sh, cu) bits of HW_ID
    uniq, cnt = np.unique(cu, return_counts=True)
    print("   ran on %d distinct CUs; workgroups per CU: max %d, mean %.2f" % (len(uniq), cnt.max(), cnt.mean()))

# the weight-gradient launch by layer (block ranges: layer 0 first): whole-workgroup durations and end times
rows5 = a[5]
n5 = int((rows5[:, 0] > 0).sum())
if n5:
    B = cfg["batch_size"]
    ipb = max(1, -(-B // 32)) if B > 32 else 1
    groups = -(-B // ipb)
    nb = [3 * groups, 2 * groups, 2 * groups] if cfg.get("architecture", "canonical") == "canonical" else [n5, 0, 0]
    t0 = rows5[:n5, 0].min()
    lo = 0
    for li, n in enumerate(nb):
        r = rows5[lo:lo + n]; lo += n
        r = r[r[:, 0] > 0]
        if len(r):
            print("conv_dw layer %d: %3d workgroups; start median +%.2f | end median +%.2f last +%.2f | duration median %.2f max %.2f us"
                  % (li, len(r), np.median(r[:, 0] - t0) * 0.01, np.median(r[:, 6] - t0) * 0.01, (r[:, 6].max() - t0) * 0.01,
                     np.median(r[:, 6] - r[:, 0]) * 0.01, (r[:, 6] - r[:, 0]).max() * 0.01))

# conv1: which workgroups pay the long input stage?  (wg index = img * 16 + cotile * 8 + chunk)
rows_all = a[0]
idx = np.nonzero(rows_all[:, 0] > 0)[0]
if len(idx):
    st_in = (rows_all[idx, 3] - rows_all[idx, 2]) * 0.01
    img, chunk = idx // 16, idx % 8
    xcc = (rows_all[idx, 7].astype(np.int64) >> 16)
    lin = chunk + (chunk.max() + 1) * img
    print("conv1 input stage by chunk:", " ".join("%d: %.2f" % (c, np.median(st_in[chunk == c])) for c in range(chunk.max() + 1)))
    print("conv1 input stage by image group (16 images each):", " ".join("%.2f" % np.median(st_in[(img // 16) == g]) for g in range(img.max() // 16 + 1)))
    print("conv1 slow (> 4 us) workgroups: %d of %d; per image count of slow:" % ((st_in > 4).sum(), len(idx)),
          np.bincount(img[st_in > 4], minlength=img.max() + 1).tolist())
    print("dispatch id %% 8 -> XCC id (first 32 workgroups):", [(int(l) % 8, int(x)) for l, x in sorted(zip(lin, xcc))[:32]])
    agree = np.mean((lin % 8) == ((xcc - xcc[np.argmin(lin)]) % 8))
    print("fraction of workgroups with XCC == (dispatch id + const) %% 8: %.3f" % agree)
    first = img < 51
    d0 = np.diff(rows_all[idx][:, 0:7], axis=1) * 0.01
    for nm, sel in (("first 255 workgroups (images < 51)", first), ("the rest (second workgroup of its CU)", ~first)):
        print("conv1 %s: start +%.2f | " % (nm, np.median(rows_all[idx][sel, 0] - rows_all[idx][:, 0].min()) * 0.01)
              + "  ".join("%s %.2f" % (phase[i], np.median(d0[sel, i])) for i in range(6)) + " | total %.2f" % np.median(d0[sel].sum(axis=1)))

# noisy-linear backward launches (kernel ids 8 = output layer, 9 = hidden layer): per role (0 priority write-back, 1 dW tiles,
# 2 dX tiles) when the workgroups start and end relative to the launch's first start
for k, nm in ((8, "fc_z_bwd"), (9, "fc_h_bwd")):
    rows = a[k][a[k][:, 0] > 0]
    if not len(rows):
        continue
    t0 = rows[:, 0].min()
    print("== %s: %d workgroups, launch span %.2f us" % (nm, len(rows), (rows[:, 6].max() - t0) * 0.01))
    for role, rn in ((0, "write-back"), (1, "dW"), (2, "dX")):
        r = rows[(rows[:, 1] == role) & (rows[:, 6] > 0)]
        if len(r):
            print("   %-10s n %4d  start median +%.2f last +%.2f | end median +%.2f last +%.2f | duration median %.2f max %.2f us"
                  % (rn, len(r), np.median(r[:, 0] - t0) * 0.01, (r[:, 0].max() - t0) * 0.01, np.median(r[:, 6] - t0) * 0.01,
                     (r[:, 6].max() - t0) * 0.01, np.median(r[:, 6] - r[:, 0]) * 0.01, (r[:, 6] - r[:, 0]).max() * 0.01))
    r2 = rows[(rows[:, 1] == 2) & (rows[:, 6] > 0)]
    if len(r2) and len(r2) <= 64:
        print("   dX durations, sorted (us):", " ".join("%.2f" % x for x in sorted((r2[:, 6] - r2[:, 0]) * 0.01)))
    r = rows[(rows[:, 1] == 2) & (rows[:, 6] > 0) & (rows[:, 2] > 0)]
    if len(r):
        d = np.diff(r[:, [0, 2, 3, 4, 5, 6]], axis=1) * 0.01
        order = np.argsort(-(r[:, 6] - r[:, 0]))
        print("   dX phases (prologue | first iteration | rest of the loop | cross-wave sum | epilogue), slowest 4 and median:")
        for i in list(order[:4]):
            print("      " + "  ".join("%.2f" % x for x in d[i]))
        print("      median " + "  ".join("%.2f" % x for x in np.median(d, axis=0)))
