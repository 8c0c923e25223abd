import os, sys, types, ctypes as C, json
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from rainbow_amd import _lib as L
from rainbow_amd.memory import ReplayMemory
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[os.environ.get("SAMPLE_CONFIG", "pong-canonical-b32")])
args = bench.make_args(cfg, dev)
mem = ReplayMemory(args, cfg["capacity"], seed=7)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
lib = L.load()
B = cfg["batch_size"]
lib.rb_debug_stamps.argtypes = [C.c_void_p]
acc = []
for it in range(60):
    mem.sample_device(B, gather=False)
    torch.cuda.synchronize()
    st = (C.c_longlong * 32)()
    lib.rb_debug_stamps(st)
    acc.append([st[i] - st[0] for i in range(6)])
a = np.array(acc[10:], dtype=np.float64) * 0.01   # 100 MHz -> us
print("stamps us (median): stage-top %.2f  descent+valid %.2f  after-loop %.2f  window/scalars %.2f  end %.2f" % tuple(np.median(a, axis=0)[1:]))
# back-to-back timing
def stream_us(fn, n):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
print("sample back-to-back us", stream_us(lambda: mem.sample_device(B, gather=False), 2000))
o = mem.sample_device(B, gather=False)
loss = torch.rand(B, device=dev) + 0.1
print("update back-to-back us", stream_us(lambda: mem.update_priorities(o["tree_idxs"], loss), 2000))
