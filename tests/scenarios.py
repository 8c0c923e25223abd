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
}
# batch sizes of the three sampling phases (small rings cannot host 32 strata next to the write head)
REPLAY_BATCHES = {"small": (8, 16, 32), "ragged": (4, 6, 5), "nstep20": (8, 16, 32)}


def replay_scenario(backend, name):
    """Drives `backend` through appends / samples / priority updates with wrap-around,
    episode boundaries inside windows, a not-yet-full buffer and duplicate leaf updates."""
    capacity, history, n, discount, omega, p_term = REPLAY_CONFIGS[name]
    rs = np.random.RandomState({"small": 11, "ragged": 22, "nstep20": 33}[name])
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
