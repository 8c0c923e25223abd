"""GPU parity tests of the replay path through the C ABI of librainbow_hip.so (hand-written HIP):
bit-exact indices / bytes against the REAL reference's golden vectors, plus size-independent
properties at the benchmark's full 1M capacity."""
import pickle
import types

import numpy as np
import pytest
import torch

import scenarios
from helpers import F32_ULP_RTOL, assert_trace_matches, load_golden
from oracle.replay_oracle import ReplayOracle, tree_geometry

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    from rainbow_amd import _lib
    return _lib.load()   # raises if the HIP library is missing: no fallback


@pytest.mark.parametrize("name", sorted(scenarios.REPLAY_CONFIGS))
def test_replay_hip_matches_reference_golden(hip, name):
    from cabi_adapter import CAbiReplayAdapter, TorchMem
    capacity, history, n, discount, omega, _ = scenarios.REPLAY_CONFIGS[name]
    ad = CAbiReplayAdapter(hip, TorchMem(), capacity, history, n, discount, omega)
    trace = scenarios.replay_scenario(ad, name)
    assert_trace_matches(trace, load_golden("replay_%s.npz" % name), label="hip/" + name)
    ad.close()


def _args(**kw):
    base = dict(device=torch.device("cuda:0"), history_length=4, discount=0.99, multi_step=3, priority_weight=0.4,
                priority_exponent=0.5)
    base.update(kw)
    return types.SimpleNamespace(**base)


def _fill(mem, oracle, count, seed, p_term=0.02):
    rs = np.random.RandomState(seed)
    for _ in range(count):
        st = rs.randint(0, 256, size=(mem.history, 84, 84)).astype(np.float32) / np.float32(255)
        a, r, term = int(rs.randint(0, 6)), float(rs.choice([-1.0, 0.0, 1.0])), bool(rs.random_sample() < p_term)
        mem.append(torch.from_numpy(st).cuda(), a, r, term)
        if oracle is not None:
            oracle.append(st, a, r, term)


def test_python_class_sample_tuple_matches_oracle(hip):
    """ReplayMemory.sample() keeps the reference's 7-tuple contract."""
    from rainbow_amd.memory import ReplayMemory
    mem = ReplayMemory(_args(), 512, seed=3)
    ora = ReplayOracle(512)
    _fill(mem, ora, 700, seed=5)
    uu = np.random.RandomState(9).random_sample((32, 16))
    o = mem.sample_device(16, torch.from_numpy(uu))
    want = ora.sample_with_uniforms(16, uu)
    assert np.array_equal(o["tree_idxs"].cpu().numpy(), want["tree_idxs"])
    assert np.array_equal(o["states"].cpu().numpy(), want["states"])
    assert np.array_equal(o["next_states"].cpu().numpy(), want["next_states"])
    assert np.array_equal(o["actions"].cpu().numpy(), want["actions"])
    np.testing.assert_allclose(o["returns"].cpu().numpy(), want["returns"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(o["weights"].cpu().numpy(), want["weights"], rtol=F32_ULP_RTOL)
    tup = mem.sample(16)   # device Philox path, reference tuple layout
    assert isinstance(tup[0], np.ndarray) and tup[0].dtype == np.int64 and tup[0].shape == (16,)
    assert tup[1].dtype == torch.float32 and tuple(tup[1].shape) == (16, 4, 84, 84) and tup[1].is_cuda
    assert tup[2].dtype == torch.int64 and tup[3].dtype == torch.float32
    assert tuple(tup[5].shape) == (16, 1) and tuple(tup[6].shape) == (16,)
    assert float(tup[1].max()) <= 1.0 and float(tup[6].max()) == 1.0
    # stratification: one sample per equal-mass segment => tree indices are non-decreasing
    assert np.all(np.diff(tup[0]) >= 0)
    mem.update_priorities(tup[0], np.abs(np.random.RandomState(1).randn(16)).astype(np.float32) + 1e-3)
    torch.cuda.synchronize()


def test_append_batch_equals_sequential_appends(hip):
    from rainbow_amd.memory import ReplayMemory
    cap, n = 1000, 1700     # wraps the ring
    rs = np.random.RandomState(4)
    frames = rs.randint(0, 256, size=(n, 84, 84)).astype(np.uint8)
    actions = rs.randint(0, 6, size=n)
    rewards = rs.choice([-1.0, 0.0, 1.0], size=n).astype(np.float32)
    terms = rs.random_sample(n) < 0.03
    a = ReplayMemory(_args(), cap, seed=1)
    b = ReplayMemory(_args(), cap, seed=1)
    ora = ReplayOracle(cap)
    for i in range(n):
        ora.append_frame(frames[i], int(actions[i]), float(rewards[i]), bool(terms[i]))
    # sequential path: states whose last frame quantises back to `frames`
    for i in range(0, n):
        st = np.zeros((4, 84, 84), dtype=np.float32)
        st[-1] = (frames[i].astype(np.float32) + 0.5) / 255.0
        a.append(torch.from_numpy(st).cuda(), int(actions[i]), float(rewards[i]), bool(terms[i]))
    for lo in range(0, n, 640):
        hi = min(n, lo + 640)
        b.append_batch(torch.from_numpy(frames[lo:hi]).cuda(), actions[lo:hi], rewards[lo:hi], terms[lo:hi])
    da, db = a._dump(), b._dump()
    for key in ("tree", "frames", "timestep", "action", "reward", "nonterminal"):
        assert np.array_equal(da[key], db[key]), key
    assert np.array_equal(da["tree"], ora.transitions.tree)
    assert np.array_equal(da["frames"].reshape(cap, 84, 84), ora.transitions.frames)
    assert np.array_equal(da["timestep"], ora.transitions.timestep)
    assert a.transitions.index == ora.transitions.index and a.transitions.full == ora.transitions.full


def test_pickle_round_trip(hip):
    from rainbow_amd.memory import ReplayMemory
    mem = ReplayMemory(_args(), 256, seed=2)
    _fill(mem, None, 300, seed=8)
    o = mem.sample_device(8)
    mem.update_priorities(o["tree_idxs"], torch.rand(8, device="cuda") + 0.1)
    blob = pickle.dumps(mem)             # main.py:94-100
    back = pickle.loads(blob)            # main.py:118
    d0, d1 = mem._dump(), back._dump()
    for key in d0:
        assert np.array_equal(np.frombuffer(d0[key], dtype=np.uint8) if isinstance(d0[key], bytes) else d0[key],
                              np.frombuffer(d1[key], dtype=np.uint8) if isinstance(d1[key], bytes) else d1[key]), key
    uu = np.random.RandomState(3).random_sample((32, 8))
    x = mem.sample_device(8, torch.from_numpy(uu))["tree_idxs"].cpu().numpy()
    y = back.sample_device(8, torch.from_numpy(uu))["tree_idxs"].cpu().numpy()
    assert np.array_equal(x, y)


def test_full_size_tree_properties(hip):
    """BASELINE config 2 capacity (1M): size-independent invariants instead of an oracle replay."""
    from rainbow_amd.memory import ReplayMemory
    cap = 1_000_000
    mem = ReplayMemory(_args(), cap, seed=6)
    g = torch.Generator(device="cuda").manual_seed(0)
    chunk = 50_000
    rs = np.random.RandomState(0)
    for lo in range(0, cap + chunk, chunk):    # 1.05M appends: full, write head mid-buffer
        fr = torch.randint(0, 256, (chunk, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
        mem.append_batch(fr, rs.randint(0, 6, chunk), rs.choice([-1.0, 0.0, 1.0], size=chunk, p=[0.05, 0.9, 0.05]),
                         rs.random_sample(chunk) < 1e-3)
    hdr = mem._header()
    assert hdr.full == 1 and hdr.index == chunk
    # non-uniform priorities through the public update path, 1024 leaves at a time
    levels, tree_start, tree_len = tree_geometry(cap)
    for r in range(40):
        idx = torch.randint(0, cap, (1024,), device="cuda", generator=g) + tree_start
        mem.update_priorities(idx, torch.rand(1024, device="cuda", generator=g) * 3 + 1e-3)
    tree = mem._grab("tree")
    leaves = tree[tree_start:]
    # every internal node is fl32(left + right) of its children  (memory.py:25,39)
    internal = np.arange(0, tree_start)
    have = 2 * internal + 2 < tree_len
    p = internal[have]
    assert np.array_equal(tree[p], tree[2 * p + 1] + tree[2 * p + 2])
    assert abs(float(tree[0]) - leaves.astype(np.float64).sum()) <= 1e-4 * float(tree[0])
    # find(v) lands on a leaf whose float64 prefix-sum bracket contains v (up to fp32 tree rounding)
    vals = np.random.RandomState(2).random_sample(4096) * float(tree[0])
    v_d = torch.from_numpy(vals).cuda()
    probs, di, ti = (torch.empty(4096, dtype=torch.float32, device="cuda"), torch.empty(4096, dtype=torch.int64, device="cuda"),
                     torch.empty(4096, dtype=torch.int64, device="cuda"))
    from rainbow_amd import _lib as L
    L.check(hip, hip.rb_replay_find(mem._h, v_d.data_ptr(), 4096, probs.data_ptr(), di.data_ptr(), ti.data_ptr(), mem._stream()))
    di = di.cpu().numpy()
    ora = ReplayOracle.__new__(ReplayOracle)   # oracle search on the device's own tree: indices must be identical
    from oracle.replay_oracle import SumTreeOracle
    st = SumTreeOracle.__new__(SumTreeOracle)
    st.capacity, st.levels, st.tree_start, st.tree_len, st.tree = cap, levels, tree_start, tree_len, tree
    _, want_di, want_ti = st.find(vals)
    assert np.array_equal(di, want_di)
    assert np.array_equal(ti.cpu().numpy(), want_ti)
    csum = np.cumsum(leaves.astype(np.float64))
    tol = 1e-5 * float(tree[0])
    assert np.all(vals <= csum[di] + tol) and np.all(vals >= csum[di] - leaves[di] - tol)
    # a sampled batch: valid distance to the write head, IS weights normalised, states are ring frames
    o = mem.sample_device(256)
    torch.cuda.synchronize()
    idx = o["tree_idxs"].cpu().numpy() - tree_start
    assert np.all((hdr.index - idx) % cap > 3) and np.all((idx - hdr.index) % cap >= 4)
    assert float(o["weights"].max()) == 1.0 and float(o["weights"].min()) > 0.0
    s = o["states"].cpu().numpy()
    for b in (0, 100, 255):
        assert np.array_equal(s[b, 3], mem._grab("frames", int(idx[b]), 1)[0])
