"""Scripted, fully seeded scenarios shared by the golden generator (which drives the REAL
reference classes, tests/golden/make_golden.py) and the parity tests (which drive the
oracle, the host-interpreted kernels and the HIP kernels).  A scenario talks to a
backend through a small adapter protocol and returns a flat {name: ndarray} trace.

Adapter protocol (replay):
    append(state_f32[h,84,84], action, reward, terminal)
    sample(batch, unit_uniforms f64[A,B], beta) -> dict(tree_idxs i64[B], states u8[B,h,84,84],
           next_states u8[B,h,84,84], actions i64[B], returns f32[B], nonterminals f32[B,1],
           weights f32[B])
    update_priorities(tree_idxs i64[n], losses f32[n])
    find(values f64[n]) -> (probs f32[n], data_idx i64[n], tree_idx i64[n])
    tree() -> f32[tree_len] copy ; header() -> (index, full, max)
    state_at(i) -> f32[h,84,84]
"""
import zlib

import numpy as np

MAX_ATTEMPTS = 64


def crc(a):
    return np.uint32(zlib.crc32(np.ascontiguousarray(a).tobytes()))


def synth_state(rs, history, kind):
    """A [h,84,84] float32 state in [0,1].  kind 0: u8/255 (what env.py produces),
    kind 1: arbitrary floats (exercises the x*255 truncation, memory.py:106)."""
    if kind == 0:
        return (rs.randint(0, 256, size=(history, 84, 84)).astype(np.float32) / np.float32(255)).astype(np.float32)
    return rs.random_sample((history, 84, 84)).astype(np.float32)


def draw_unit_uniforms(batch, seed):
    """The U[0,1) doubles the sampler will consume, one row per rejection attempt."""
    return np.random.RandomState(seed).random_sample((MAX_ATTEMPTS, batch))


REPLAY_CONFIGS = {
    # name: (capacity, history, multi_step, discount, omega, terminal_prob)
    "small": (256, 4, 3, 0.99, 0.5, 0.05),
    "ragged": (100, 2, 5, 0.9, 0.7, 0.08),       # non power-of-two capacity, other h/n/omega
    "nstep20": (1000, 4, 20, 0.99, 0.5, 0.01),   # data-efficient window (n=20)
    "c4096": (4096, 4, 3, 0.99, 0.5, 0.02),      # C = 4096 with the benchmark batch sizes 32 and 256
}
# batch sizes of the three sampling phases (small rings cannot host 32 strata next to the write head)
REPLAY_BATCHES = {"small": (8, 16, 32), "ragged": (4, 6, 5), "nstep20": (8, 16, 32), "c4096": (32, 256, 256)}


def replay_scenario(backend, name):
    """Drives `backend` through appends / samples / priority updates with wrap-around,
    episode boundaries inside windows, a not-yet-full buffer and duplicate leaf updates."""
    capacity, history, n, discount, omega, p_term = REPLAY_CONFIGS[name]
    rs = np.random.RandomState({"small": 11, "ragged": 22, "nstep20": 33, "c4096": 44}[name])
    b1, b2, b3 = REPLAY_BATCHES[name]
    trace = {}
    step = [0]

    def rec(key, val):
        trace["%02d_%s" % (step[0], key)] = np.asarray(val)

    def do_appends(count):
        for _ in range(count):
            state = synth_state(rs, history, int(rs.randint(0, 2)))
            action = int(rs.randint(0, 6))
            reward = float(rs.choice([-1.0, 0.0, 1.0], p=[0.05, 0.9, 0.05]))
            terminal = bool(rs.random_sample() < p_term)
            backend.append(state, action, reward, terminal)
        idx, full, mx = backend.header()
        rec("hdr", np.array([idx, int(full)], dtype=np.int64))
        rec("max", np.float32(mx))
        rec("tree", backend.tree())
        step[0] += 1

    def do_sample(batch, beta, seed):
        uu = draw_unit_uniforms(batch, seed)
        out = backend.sample(batch, uu, beta)
        rec("tree_idxs", out["tree_idxs"])
        rec("actions", out["actions"])
        rec("returns", out["returns"])
        rec("nonterminals", out["nonterminals"])
        rec("weights", out["weights"])
        rec("states_crc", crc(out["states"]))
        rec("next_states_crc", crc(out["next_states"]))
        rec("states_sum", np.asarray(out["states"], dtype=np.int64).sum(axis=(2, 3)))
        rec("next_states_sum", np.asarray(out["next_states"], dtype=np.int64).sum(axis=(2, 3)))
        step[0] += 1
        return out

    def do_update(tree_idxs, seed):
        losses = (np.abs(np.random.RandomState(seed).randn(len(tree_idxs))) + 1e-3).astype(np.float32)
        backend.update_priorities(np.asarray(tree_idxs, dtype=np.int64), losses)
        idx, full, mx = backend.header()
        rec("max", np.float32(mx))
        rec("tree", backend.tree())
        step[0] += 1

    # 1. partially filled buffer
    do_appends(int(capacity * 0.6))
    out = do_sample(b1, 0.4, 101)
    do_update(out["tree_idxs"], 201)
    # 2. duplicate leaves in one update (last write wins, memory.py:45)
    dup = np.concatenate([out["tree_idxs"][:4], out["tree_idxs"][:4][::-1], out["tree_idxs"][:2]])
    do_update(dup, 202)
    # 3. wrap the ring, buffer becomes full
    do_appends(int(capacity * 0.7))
    for r in range(3):
        out = do_sample(b2, 0.4 + 0.2 * r, 110 + r)
        do_update(out["tree_idxs"], 210 + r)
    # 4. a few single appends between samples (write head moves under the sampler)
    do_appends(5)
    out = do_sample(b3, 1.0, 120)
    do_update(out["tree_idxs"], 220)
    # 5. tree search on hand-picked values: 0, total, beyond total (clamp), node boundaries
    tree = backend.tree()
    total = np.float64(tree[0])
    vals = np.concatenate([
        np.array([0.0, total, total * (1 + 1e-6), total * 0.5, np.float64(tree[1]), np.nextafter(np.float64(tree[1]), np.inf)]),
        np.random.RandomState(301).random_sample(58) * total,
    ])
    probs, data_idx, tree_idx = backend.find(vals)
    rec("find_values", vals)
    rec("find_probs", probs)
    rec("find_data_idx", data_idx)
    rec("find_tree_idx", tree_idx)
    step[0] += 1
    # 6. validation-iterator view at a few indices incl. 0 (negative wrap) and episode starts
    for i in [0, 1, history - 1, capacity // 2, capacity - 1]:
        rec("state_at_%d_crc" % i, crc(np.rint(backend.state_at(i) * 255).astype(np.uint8)))
    step[0] += 1
    return trace


# keys compared exactly (integers / indices / byte checksums) vs. with a float tolerance
def is_exact_key(key):
    k = key.split("_", 1)[1]
    return any(k.startswith(p) for p in ("tree_idxs", "actions", "hdr", "states_crc", "next_states_crc", "states_sum",
                                         "next_states_sum", "find_data_idx", "find_tree_idx", "state_at", "nonterminals",
                                         "find_values"))


# =============================================================================== learn step
LEARN_CONFIGS = {
    # canonical conv stack (k8s4,k4s2,k3s1) with a narrow hidden layer to keep fixtures small
    "canon": dict(architecture="canonical", hidden=64, actions=6, atoms=51, batch=8, multi_step=3, discount=0.99,
                  history=4, v_min=-10.0, v_max=10.0),
    # data-efficient stack (k5s5,k5s5), n=20
    "dataeff": dict(architecture="data-efficient", hidden=32, actions=4, atoms=51, batch=6, multi_step=20,
                    discount=0.99, history=4, v_min=-10.0, v_max=10.0),
    # other atom count / support / action count / history
    "atoms21": dict(architecture="data-efficient", hidden=48, actions=3, atoms=21, batch=5, multi_step=1,
                    discount=0.9, history=2, v_min=-5.0, v_max=5.0),
}
LEARN_CONFIGS["k10"] = dict(architecture="data-efficient", hidden=32, actions=3, atoms=51, batch=4, multi_step=3,
                            discount=0.99, history=4, v_min=-10.0, v_max=10.0)   # K = 1 and K = 10
# north_star: "loss curves matching reference within tolerance" — a 200-step trajectory of the reference (agent.py:61-100)
# with injected noise and fixed per-step batches, one update_target_net (agent.py:102-103) on the way
LEARN_CONFIGS["k200"] = dict(architecture="data-efficient", hidden=32, actions=3, atoms=51, batch=4, multi_step=3,
                             discount=0.99, history=4, v_min=-10.0, v_max=10.0)
LEARN_HYPER = dict(lr=6.25e-5, adam_eps=1.5e-4, norm_clip=10.0)   # main.py:43-46 defaults
LEARN_STEPS = 3
LEARN_STEPS_BY = {"k10": 10, "k200": 200}    # other configs: LEARN_STEPS
LEARN_FULL_RECORD = {"k10": (0, 9), "k200": (29, 99, 199)}   # steps whose gradient / parameter summaries are kept (default: all)
LEARN_SYNC_AT = {"k200": (100,)}             # backend.sync_target() (agent.py:102-103) BEFORE these steps


def make_batch(cfg, seed):
    """Synthetic learn() inputs incl. the projection's corner cases: clamped returns (|R| beyond
    the support), terminal rows (all mass on two atoms) and integer-b rows (agent.py:85-86)."""
    rs = np.random.RandomState(seed)
    B, h = cfg["batch"], cfg["history"]
    states = rs.randint(0, 256, size=(B, h, 84, 84)).astype(np.uint8)
    next_states = rs.randint(0, 256, size=(B, h, 84, 84)).astype(np.uint8)
    states[0, 0] = 0                      # a blanked frame
    actions = rs.randint(0, cfg["actions"], size=B).astype(np.int64)
    returns = rs.uniform(-3, 3, size=B).astype(np.float32)
    nonterminals = (rs.random_sample(B) < 0.7).astype(np.float32)
    returns[0] = 0.0; nonterminals[0] = 0.0                     # b exactly integer (mid support)
    returns[1] = cfg["v_max"] + 2.0                             # clamp high: b = Z-1 for every atom
    returns[2] = cfg["v_min"] - 2.0; nonterminals[2] = 1.0      # clamp low / partially
    if B > 3:
        returns[3] = 1.0; nonterminals[3] = 0.0
    weights = rs.uniform(0.2, 1.0, size=B).astype(np.float32)
    weights[rs.randint(0, B)] = 1.0
    return dict(states=states, next_states=next_states, actions=actions, returns=returns,
                nonterminals=nonterminals, weights=weights)


def summarize(a):
    """Full tensor when small, else a strided sample (keeps fixtures small)."""
    a = np.asarray(a, dtype=np.float32).ravel()
    if a.size <= 4096:
        return a.copy()
    stride = a.size // 2048 + 1
    return a[::stride].copy()


def learn_scenario(backend, name, oracle_mod, steps=None):
    """backend protocol:
         load(online_params, target_params)            # {state-dict name: f32 array}
         reset_noise_online(raw_normals f32[D])
         learn_step(batch, target_raw_normals) -> dict(loss f32[B], grad_norm float, grads {name: f32 array (clipped)})
         params() -> {name: array}                      # online, after the optimiser step
         act(state_f32[h,84,84], noisy) -> (action, q)
         sync_target()                                  # Agent.update_target_net (only scenarios listed in LEARN_SYNC_AT)
    oracle_mod supplies Config/init_params/noise_draw_count only (pure bookkeeping)."""
    c = LEARN_CONFIGS[name]
    cfg = oracle_mod.Config(**c)
    seed0 = {"canon": 1000, "dataeff": 2000, "atoms21": 3000, "k10": 4000, "k200": 5000}[name]
    online = oracle_mod.init_params(cfg, seed0)
    target = oracle_mod.init_params(cfg, seed0 + 1)
    backend.load(online, target)
    draws = oracle_mod.noise_draw_count(cfg)
    trace = {}
    full = LEARN_FULL_RECORD.get(name)
    for k in range(steps if steps is not None else LEARN_STEPS_BY.get(name, LEARN_STEPS)):
        if k in LEARN_SYNC_AT.get(name, ()):
            backend.sync_target()
        rs = np.random.RandomState(seed0 + 10 + k)
        backend.reset_noise_online(rs.randn(draws).astype(np.float32))
        batch = make_batch(c, seed0 + 20 + k)
        out = backend.learn_step(batch, rs.randn(draws).astype(np.float32))
        trace["s%d_loss" % k] = np.asarray(out["loss"], dtype=np.float32)
        trace["s%d_grad_norm" % k] = np.float32(out["grad_norm"])
        if full is not None and k not in full:
            continue
        for pname, g in out["grads"].items():
            trace["s%d_grad/%s" % (k, pname)] = summarize(g)
            trace["s%d_gradnorm/%s" % (k, pname)] = np.float32(np.sqrt(np.sum(np.asarray(g, dtype=np.float64) ** 2)))
        for pname, p in backend.params().items():
            trace["s%d_param/%s" % (k, pname)] = summarize(p)
    if steps is not None:          # shortened run (a test that only needs the first steps): no acting epilogue
        return trace
    st = synth_state(np.random.RandomState(seed0 + 99), c["history"], 0)
    a_noisy, q_noisy = backend.act(st, True)
    a_eval, q_eval = backend.act(st, False)
    trace["act_noisy"] = np.array([a_noisy], dtype=np.int64)
    trace["act_eval"] = np.array([a_eval], dtype=np.int64)
    trace["q_noisy"] = np.float32(q_noisy)
    trace["q_eval"] = np.float32(q_eval)
    return trace


def sampler_gives_up_check(lib, mem):
    """A ring too small for the batch (stratum 0 lies inside the write head's exclusion zone) has NO valid batch; the
    reference would spin forever (memory.py:128-132).  The bounded device loop must then emit zero importance weights
    (zero-gradient step, no inf/NaN from a zero-priority leaf), flag the header and count the failure in the pinned host
    word.  Shared by the emulator test and the GPU test (`mem` = NumpyMem / TorchMem)."""
    import ctypes as C
    from cabi_adapter import CAbiReplayAdapter
    from rainbow_amd import _lib as L
    ad = CAbiReplayAdapter(lib, mem, 16, 4, 3, 0.99, 0.5)
    rs = np.random.RandomState(0)
    for _ in range(16):
        ad.append(synth_state(rs, 4, 0), 1, 0.0, False)
    B, attempts = 8, 12
    uu = mem.upload(rs.random_sample((attempts, B)))
    outs = dict(tree_idx=mem.empty((B,), np.int64), actions=mem.empty((B,), np.int64), returns=mem.empty((B,), np.float32),
                nonterm=mem.empty((B,), np.float32), weights=mem.upload(np.full(B, 7.0, np.float32)))
    n0 = C.c_int64(-1)
    L.check(lib, lib.rb_replay_failed_samples(ad.h, C.byref(n0)))
    assert n0.value == 0
    for k in range(2):
        L.check(lib, lib.rb_replay_sample(ad.h, B, 0.5, mem.ptr(uu), attempts, mem.ptr(outs["tree_idx"]), None, None,
                                          mem.ptr(outs["actions"]), mem.ptr(outs["returns"]), mem.ptr(outs["nonterm"]),
                                          mem.ptr(outs["weights"]), mem.stream))
        mem.sync()
        hdr = ad.raw_header()
        assert hdr.last_status == 1 and hdr.last_attempts == attempts
        assert np.array_equal(mem.download(outs["weights"]), np.zeros(B, np.float32))
        L.check(lib, lib.rb_replay_failed_samples(ad.h, C.byref(n0)))
        assert n0.value == k + 1
    # ... and the failed draw leaves NO trace: (a) ReplayMemory.update_priorities on its (illegal) indices is dropped,
    tree0 = ad.tree()
    losses = mem.upload(np.full(B, 2.5, np.float32))
    L.check(lib, lib.rb_replay_update_priorities(ad.h, mem.ptr(outs["tree_idx"]), mem.ptr(losses), B, mem.stream))
    mem.sync()
    assert np.array_equal(ad.tree(), tree0), "update_priorities after a failed draw changed the sum-tree"
    nd = C.c_int64(-1)
    L.check(lib, lib.rb_replay_dropped_updates(ad.h, C.byref(nd)))
    assert nd.value == 1, "the dropped write-back was not counted"
    assert np.array_equal(mem.download(outs["tree_idx"]), np.full(B, -1, np.int64)), "a failed draw must mark its index buffer"
    # (b) a learner whose priority sink is this replay skips the fused write-back, the optimiser update and the
    # device-resident step number (Adam's momentum would otherwise move the parameters on a zero gradient)
    from cabi_adapter import CAbiLearnAdapter
    from oracle import learner_oracle as O
    c = LEARN_CONFIGS["canon"]
    assert c["batch"] == B
    cfg = O.Config(**c)
    la = CAbiLearnAdapter(lib, mem, "canon")
    la.load(O.init_params(cfg, 31), O.init_params(cfg, 32))
    if not isinstance(la.adam_m, np.ndarray):
        la.adam_m.fill_(0.25)                 # non-zero momentum: an un-skipped Adam pass WOULD move the parameters
    else:
        la.adam_m[:] = 0.25
    step_dev = mem.empty((1,), np.int64)
    L.check(lib, lib.rb_learner_set_step_counter(la.h, mem.ptr(step_dev)))
    L.check(lib, lib.rb_learner_set_priority_sink(la.h, ad.h, mem.ptr(outs["tree_idx"])))
    p0, m0 = mem.download(la.p_on.detach() if hasattr(la.p_on, "detach") else la.p_on), mem.download(la.adam_m)
    la.reset_noise_online(rs.randn(O.noise_draw_count(cfg)).astype(np.float32))
    batch = make_batch(c, 17)
    batch["weights"] = np.zeros(B, np.float32)                    # what the sampler wrote
    out = la.learn_step(batch, rs.randn(O.noise_draw_count(cfg)).astype(np.float32))
    assert out["grad_norm"] == 0.0
    assert np.array_equal(mem.download(la.p_on.detach() if hasattr(la.p_on, "detach") else la.p_on), p0), "parameters moved"
    assert np.array_equal(mem.download(la.adam_m), m0), "Adam moments moved"
    assert int(mem.download(step_dev)[0]) == 0, "device step counter advanced"
    assert np.array_equal(ad.tree(), tree0), "fused priority write-back ran on an illegal batch"
    L.check(lib, lib.rb_learner_set_priority_sink(la.h, None, None))
    la.close()
    # (c) the counter can be cleared once the failure has been reported
    L.check(lib, lib.rb_replay_reset_failed_samples(ad.h))
    L.check(lib, lib.rb_replay_failed_samples(ad.h, C.byref(n0)))
    assert n0.value == 0
    # the host mirror of the write position needs no device round trip and follows a raw header restore
    idx, full = C.c_int64(-1), C.c_int32(-1)
    L.check(lib, lib.rb_replay_position(ad.h, C.byref(idx), C.byref(full)))
    assert (idx.value, full.value) == (0, 1)
    return ad


def earlier_valid_batch_survives_failed_draw_check(lib, mem):
    """update_priorities of an EARLIER, valid batch after a LATER draw gave up (memory.py:157-159 has no notion of a failed draw:
    the reference would still be spinning in memory.py:128-132).  Only the write-back whose indices the failed draw itself
    produced is dropped (its index buffer is marked, rb_replay_dropped_updates counts it); the valid batch's priorities land in
    the tree exactly as the oracle's SumTree puts them, although the header's status word says 'failed' by then.  Both the
    one-wave sorted path (rb_replay_update_priorities, n <= 64) and the fused update + sample launch are exercised."""
    import ctypes as C
    from cabi_adapter import CAbiReplayAdapter
    from oracle.replay_oracle import ReplayOracle
    from rainbow_amd import _lib as L
    cap, h, n = 32, 4, 3
    ad = CAbiReplayAdapter(lib, mem, cap, h, n, 0.99, 0.5)
    ora = ReplayOracle(cap, history=h, discount=0.99, multi_step=n, priority_weight=0.5, priority_exponent=0.5)
    rs = np.random.RandomState(4)
    for _ in range(cap + 5):
        st = synth_state(rs, h, 0)
        ad.append(st, 1, 0.0, False)
        ora.append(st, 1, 0.0, False)
    uu2 = rs.random_sample((32, 2))
    good = ad.sample(2, uu2, 0.5)                         # two strata of 16 leaves: a valid batch exists (asserted inside)
    want = ora.sample_with_uniforms(2, uu2)
    assert np.array_equal(good["tree_idxs"], want["tree_idxs"])
    B, attempts = 8, 6                                    # eight strata of 4 leaves: one lies inside the write head's zone
    outs = dict(tree_idx=mem.empty((B,), np.int64), actions=mem.empty((B,), np.int64), returns=mem.empty((B,), np.float32),
                nonterm=mem.empty((B,), np.float32), weights=mem.empty((B,), np.float32))
    uu8 = mem.upload(rs.random_sample((attempts, B)))

    def failing_draw():
        L.check(lib, lib.rb_replay_sample(ad.h, B, 0.5, mem.ptr(uu8), attempts, mem.ptr(outs["tree_idx"]), None, None,
                                          mem.ptr(outs["actions"]), mem.ptr(outs["returns"]), mem.ptr(outs["nonterm"]),
                                          mem.ptr(outs["weights"]), mem.stream))
        mem.sync()
        assert ad.raw_header().last_status == 1
        assert np.array_equal(mem.download(outs["tree_idx"]), np.full(B, -1, np.int64))

    failing_draw()
    loss = np.array([2.5, 0.7], np.float32)
    ad.update_priorities(good["tree_idxs"], loss)         # the EARLIER valid batch: applied
    ora.update_priorities(want["tree_idxs"], loss)
    np.testing.assert_allclose(ad.tree(), ora.transitions.tree, rtol=4e-7)
    nd = C.c_int64(-1)
    L.check(lib, lib.rb_replay_dropped_updates(ad.h, C.byref(nd)))
    assert nd.value == 0
    tree1 = ad.tree()
    bad = mem.upload(np.full(B, 1.5, np.float32))          # the failed draw's own buffer: dropped and counted
    L.check(lib, lib.rb_replay_update_priorities(ad.h, mem.ptr(outs["tree_idx"]), mem.ptr(bad), B, mem.stream))
    mem.sync()
    assert np.array_equal(ad.tree(), tree1)
    L.check(lib, lib.rb_replay_dropped_updates(ad.h, C.byref(nd)))
    assert nd.value == 1
    # the same through the fused update + sample launch (the draw inside it fails again; the update in front of it applies)
    loss2 = np.array([0.3, 4.0], np.float32)
    ad.update_sample(good["tree_idxs"], loss2, B, rs.random_sample((attempts, B)), 0.5, check_status=False)
    ora.update_priorities(want["tree_idxs"], loss2)
    np.testing.assert_allclose(ad.tree(), ora.transitions.tree, rtol=4e-7)
    assert ad.raw_header().last_status == 1
    L.check(lib, lib.rb_replay_dropped_updates(ad.h, C.byref(nd)))
    assert nd.value == 1
    ad.close()


def update_sample_twin_check(make_adapter, capacity=6000, history=4, n=3, rounds=6, seed=11):
    """rb_replay_update_sample (update_priorities of step k + sample of step k + 1 as ONE launch) against the two calls on a twin
    replay: tree, header and every output of the batch bit-identical, round after round (the one-launch sampler searches from
    an LDS tree top that the update patched).  Batches: sorted (the sampler's own indices, the one-wave update), shuffled (the
    hashed body inside the launch), 200 leaves (hashed), 300 leaves (the two-launch fallback of the entry point)."""
    rs = np.random.RandomState(seed)
    a, b = make_adapter(capacity, history, n), make_adapter(capacity, history, n)
    total = capacity + capacity // 4
    term = rs.random_sample(total) < 0.02
    ts = np.zeros(total, dtype=np.int32)
    t = 0
    for i in range(total):
        ts[i] = t
        t = 0 if term[i] else t + 1
    actions = rs.randint(0, 6, total).astype(np.int32)
    rewards = rs.choice([-1.0, 0.0, 1.0], size=total).astype(np.float32)
    pool = rs.randint(0, 256, size=(32, 84, 84)).astype(np.uint8)
    step = min(2000, capacity // 2)
    for lo in range(0, total, step):
        hi = min(total, lo + step)
        frames = pool[(np.arange(lo, hi) * 5) % 32]
        for ad in (a, b):
            ad.append_batch(frames, ts[lo:hi], actions[lo:hi], rewards[lo:hi], (~term[lo:hi]).astype(np.uint8))
    tree_start = a.bufs.tree_start
    spread = rs.randint(0, capacity, 1024) + tree_start
    vals = (rs.random_sample(1024) * 4 + 1e-3).astype(np.float32)
    for ad in (a, b):
        ad.update_leaves(spread, vals)
    B = 32
    uu = rs.random_sample((16, B))
    prev = a.sample(B, uu, 0.5)
    assert np.array_equal(prev["tree_idxs"], b.sample(B, uu, 0.5)["tree_idxs"])
    upd_idx = prev["tree_idxs"]
    for r in range(rounds):
        kind = ("sorted", "sorted", "shuffled", "many", "fallback", "sorted")[r % 6]
        if kind == "shuffled":
            upd_idx = upd_idx[rs.permutation(len(upd_idx))]
        elif kind == "many":
            upd_idx = np.sort(rs.randint(0, capacity, 200)) + tree_start
        elif kind == "fallback":
            upd_idx = rs.randint(0, capacity, 300) + tree_start
        loss = (rs.random_sample(len(upd_idx)) * 2 + 1e-3).astype(np.float32)
        uu = rs.random_sample((16, B))
        a.update_priorities(upd_idx, loss)
        want = a.sample(B, uu, 0.5)
        got = b.update_sample(upd_idx, loss, B, uu, 0.5)
        assert np.array_equal(a.tree(), b.tree()), (r, kind)
        ha, hb = a.raw_header(), b.raw_header()
        assert (ha.max, ha.total, ha.last_attempts, ha.last_status, ha.rng_counter) == (hb.max, hb.total, hb.last_attempts, hb.last_status, hb.rng_counter)
        for k in want:
            assert np.array_equal(np.asarray(want[k]), np.asarray(got[k])), (r, kind, k)
        upd_idx = want["tree_idxs"]
    a.close(); b.close()
