"""The PER-only loop of bench.py (sample_device + update_priorities, no learner) on its own: wall time per batch and the
stream time of its pieces, so that host-bound and GPU-bound can be told apart."""
import json, os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from rainbow_amd.memory import ReplayMemory  # noqa: E402
dev = torch.device("cuda", 0)
cfg = dict(bench.CONFIGS[os.environ.get("SAMPLE_CONFIG", "pong-canonical-b32")])
cfg["capacity"] = int(os.environ.get("PER_CAPACITY", cfg["capacity"]))
args = bench.make_args(cfg, dev)
mem = ReplayMemory(args, cfg["capacity"], seed=7)
bench.fill_replay(mem, cfg["capacity"], cfg["actions"], seed=0)
B = cfg["batch_size"]
loss = torch.rand(B, device=dev) + 0.1
DONATE = os.environ.get("PER_DONATE", "1") == "1"      # PER_DONATE=0: the library stages the caller-owned loss tensor at every call
def loop(n, gather=True, update=True):
    for _ in range(50):
        o = mem.sample_device(B, gather=gather)
        if update: mem.update_priorities(o["tree_idxs"], loss, donate=DONATE)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n):
        o = mem.sample_device(B, gather=gather)
        if update: mem.update_priorities(o["tree_idxs"], loss, donate=DONATE)
    mem.flush()
    t1 = time.perf_counter()
    torch.cuda.synchronize(dev)
    t2 = time.perf_counter()
    return (t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6
out = {}
for name, g, u in (("sample+gather+update", True, True), ("sample+update", False, True), ("sample+gather", True, False), ("sample", False, False)):
    w, h = loop(1000, g, u)
    out[name] = dict(wall_us=round(w, 2), host_enqueue_us=round(h, 2))
out["samples_per_s"] = round(B / out["sample+gather+update"]["wall_us"] * 1e6)
out["donate"] = DONATE
print(json.dumps(out))
