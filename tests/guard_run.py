"""Out-of-bounds WRITE hunt (run as a script, with RB_GUARD=1 in the environment: the library's own allocations are then
guarded too — include/rainbow_hip.h rb_debug_check_guards).  Drives the hot path through the C ABI with every
caller-owned buffer canaried (tests/guarded_mem.py) and exits non-zero if any guard band changed.
  python tests/guard_run.py emu            host-interpreted kernels, small shapes (CPU)
  python tests/guard_run.py hip [shapes]   librainbow_hip.so on cuda:0, BASELINE shapes (cfg2,cfg3,cfg4)
"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert os.environ.get("RB_GUARD") == "1", "run with RB_GUARD=1"

import scenarios  # noqa: E402
from oracle import learner_oracle as O  # noqa: E402

SHAPES_HIP = {
    "cfg2": dict(architecture="canonical", hidden=512, actions=6, atoms=51, batch=32, multi_step=3, discount=0.99,
                 history=4, v_min=-10.0, v_max=10.0),
    "cfg3": dict(architecture="canonical", hidden=512, actions=4, atoms=51, batch=256, multi_step=3, discount=0.99,
                 history=4, v_min=-10.0, v_max=10.0),
    "cfg4": dict(architecture="data-efficient", hidden=256, actions=6, atoms=51, batch=32, multi_step=20, discount=0.99,
                 history=4, v_min=-10.0, v_max=10.0),
}


def lib_guards(lib, label):
    from rainbow_amd import _lib as L
    nb, bad = C.c_int64(0), C.c_int64(0)
    L.check(lib, lib.rb_debug_check_guards(C.byref(nb), C.byref(bad)))
    print("  %-44s library blocks %4d, overwritten guard bands %d" % (label, nb.value, bad.value), flush=True)
    assert nb.value > 0, "library allocations are not guarded (RB_GUARD read too late?)"
    assert bad.value == 0, lib.rb_last_error().decode()


def mem_guards(mem, label):
    n, bad = mem.check()
    print("  %-44s caller blocks  %4d, overwritten guard bands %d" % (label, n, len(bad)), flush=True)
    assert not bad, "caller-owned buffers with overwritten guard bands: %s" % bad[:8]


def learn_steps(lib, mem, name, cfgd, steps, act_n, pair_exchange):
    from cabi_adapter import CAbiLearnAdapter
    from rainbow_amd import _lib as L
    scenarios.LEARN_CONFIGS[name] = cfgd
    cfg = O.Config(**cfgd)
    ads = [CAbiLearnAdapter(lib, mem, name) for _ in range(2 if pair_exchange else 1)]
    online, target = O.init_params(cfg, 41), O.init_params(cfg, 42)
    for ad in ads:
        ad.load(online, target)
    draws = O.noise_draw_count(cfg)
    rs = np.random.RandomState(9)
    ad = ads[0]
    for k in range(steps):
        ad.reset_noise_online(rs.randn(draws).astype(np.float32))
        out = ad.learn_step(scenarios.make_batch(cfgd, 50 + k), rs.randn(draws).astype(np.float32))
        assert np.isfinite(out["loss"]).all()
    st = rs.random_sample((act_n, cfgd["history"], 84, 84)).astype(np.float32)
    ad.act_batch(st, True)
    ad.act(st[0], False)
    lib_guards(lib, name + ": learn x%d, act_batch(%d)" % (steps, act_n))
    mem_guards(mem, name)
    if pair_exchange:     # the replica exchange (set_exchange / pack_factors / finish_grads) between two handles
        f, off, n = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L.check(lib, lib.rb_learner_exchange_layout(ad.h, C.byref(f), C.byref(off), C.byref(n)))
        local = [mem.empty((f.value,), np.float32) for _ in ads]
        allb = [mem.empty((2 * f.value,), np.float32) for _ in ads]
        for a, lo, al in zip(ads, local, allb):
            L.check(lib, lib.rb_learner_set_exchange(a.h, 2, mem.ptr(lo), mem.ptr(al)))
        for r, a in enumerate(ads):
            a.reset_noise_online(rs.randn(draws).astype(np.float32))
            a.learn_only(scenarios.make_batch(cfgd, 70 + r), rs.randn(draws).astype(np.float32))
        mem.sync()
        for a, al in zip(ads, allb):
            L.check(lib, lib.rb_learner_wait_factors(a.h, mem.stream))
            al[:f.value] = local[0]
            al[f.value:] = local[1]
            L.check(lib, lib.rb_learner_finish_grads(a.h, mem.stream))
            a.finish_step()
        lib_guards(lib, name + ": factored exchange of two handles")
        mem_guards(mem, name + " exchange")
    for a in ads:
        a.close()


def replay_ops(lib, mem, capacity, history, n, batch):
    from cabi_adapter import CAbiReplayAdapter
    rp = CAbiReplayAdapter(lib, mem, capacity, history, n, 0.99, 0.5)
    rs = np.random.RandomState(3)
    T = capacity + capacity // 3                                     # wraps the ring
    frames = rs.randint(0, 256, size=(T, 84 * 84)).astype(np.uint8)
    ts = (np.arange(T) % 97).astype(np.int32)
    acts, rews, nts = rs.randint(0, 4, T), rs.choice([-1.0, 0.0, 1.0], T), (ts != 96).astype(np.uint8)
    for lo in range(0, T, capacity // 2 + 1):                        # chunks <= capacity; the last ones wrap the ring
        hi = min(T, lo + capacity // 2 + 1)
        rp.append_batch(frames[lo:hi], ts[lo:hi], acts[lo:hi], rews[lo:hi], nts[lo:hi])
    for k in range(3):
        rp.append(scenarios.synth_state(rs, history, k % 2), 1, 0.5, k == 1)
    for k in range(3):
        o = rp.sample(batch, None, 0.4)
        rp.update_priorities(o["tree_idxs"], rs.random_sample(batch).astype(np.float32) + 0.01)
    rp.find(rs.random_sample(64) * float(rp.tree()[0]))
    rp.state_at(5)
    lib_guards(lib, "replay C=%d h=%d n=%d B=%d" % (capacity, history, n, batch))
    mem_guards(mem, "replay")
    rp.close()


def main():
    backend = sys.argv[1]
    if backend == "emu":
        from hipemu import loader
        from guarded_mem import GuardedNumpyMem
        lib, mem = loader.load(), GuardedNumpyMem()
        replay_ops(lib, mem, 300, 4, 3, 8)
        replay_ops(lib, mem, 6000, 2, 20, 16)
        learn_steps(lib, mem, "g-canon", dict(scenarios.LEARN_CONFIGS["canon"]), 2, 3, False)
        learn_steps(lib, mem, "g-dataeff", dict(scenarios.LEARN_CONFIGS["dataeff"]), 2, 3, True)
    else:
        import torch  # noqa: F401
        from guarded_mem import GuardedTorchMem
        from rainbow_amd import _lib
        lib, mem = _lib.load(), GuardedTorchMem()
        which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["cfg2", "cfg3", "cfg4"]
        replay_ops(lib, mem, 4096, 4, 3, 32)
        replay_ops(lib, mem, 100000, 4, 20, 256)
        for w in which:
            learn_steps(lib, mem, "g-" + w, SHAPES_HIP[w], 3, 4096 if w == "cfg2" else 64, w != "cfg3")
    print("guard run ok")


if __name__ == "__main__":
    main()
