"""GPU parity tests of the replay path through the C ABI of librainbow_hip.so (hand-written HIP):
bit-exact indices / bytes against the REAL reference's golden vectors, plus size-independent
properties at the benchmark's full 1M capacity."""
import pickle
import types

import numpy as np
import pytest
import torch

import scenarios
from helpers import F32_ULP_RTOL, assert_trace_matches, load_golden, oracle_view_of_device_replay
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


def test_sampler_gives_up_harmlessly_on_device(hip):
    """The bounded rejection loop on a ring that has no valid batch: zero importance weights, header flag, and the failure
    count in the pinned host word the kernel bumps (system-scope load + store) — the emulator runs the same check."""
    from cabi_adapter import TorchMem
    scenarios.sampler_gives_up_check(hip, TorchMem()).close()


def test_earlier_valid_batch_survives_a_failed_draw_on_device(hip):
    """update_priorities of an earlier valid batch after a later draw gave up is APPLIED (== the oracle's tree); only the
    write-back on the failed draw's own (marked) index buffer is dropped and counted (rb_replay_dropped_updates)."""
    from cabi_adapter import TorchMem
    scenarios.earlier_valid_batch_survives_failed_draw_check(hip, TorchMem())


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


@pytest.mark.parametrize("capacity,n", [(6000, 3), (100000, 20)])
def test_update_and_sample_in_one_launch_equals_the_two_calls_on_device(hip, capacity, n):
    """k_update_sample on hardware: tree, header and batch bit-identical to rb_replay_update_priorities + rb_replay_sample on a
    twin replay, six rounds (sorted / shuffled / 200 / 300 leaves) — the same scenario the host interpreter runs."""
    from cabi_adapter import CAbiReplayAdapter, TorchMem
    scenarios.update_sample_twin_check(lambda c, h, nn: CAbiReplayAdapter(hip, TorchMem(), c, h, nn, 0.99, 0.5), capacity=capacity, n=n)


def test_lazy_update_priorities_equals_immediate(hip, monkeypatch):
    """ReplayMemory.update_priorities is lazy (the next sample_device issues write-back + draw as ONE launch); everything that
    looks at the tree in between applies it first.  Twin memories — one lazy, one with RAINBOW_AMD_LAZY_PRIORITIES=0 — through
    sample / update / append / header / find / a numpy-operand update: batches, trees and headers bit-identical throughout."""
    from rainbow_amd import _lib as L
    from rainbow_amd.memory import ReplayMemory
    monkeypatch.setenv("RAINBOW_AMD_LAZY_PRIORITIES", "0")
    eager = ReplayMemory(_args(), 4096, seed=3)
    monkeypatch.setenv("RAINBOW_AMD_LAZY_PRIORITIES", "1")
    lazy = ReplayMemory(_args(), 4096, seed=3)
    assert lazy._lazy and not eager._lazy
    for m in (eager, lazy):
        _fill(m, None, 4500, seed=5)
    rs = np.random.RandomState(9)

    def tree_of(m):
        return m._grab("tree")

    def same():
        he, hl = eager._header(), lazy._header()
        assert (he.index, he.full, he.max, he.total, he.last_attempts, he.last_status) == (hl.index, hl.full, hl.max, hl.total, hl.last_attempts, hl.last_status)
        assert np.array_equal(tree_of(eager), tree_of(lazy))

    B = 32
    for r in range(8):
        uu = torch.from_numpy(rs.random_sample((8, B))).cuda()
        oe, ol = eager.sample_device(B, unit_uniforms=uu), lazy.sample_device(B, unit_uniforms=uu)
        assert lazy._pending is None
        for k in oe:
            assert torch.equal(oe[k], ol[k]), (r, k)
        loss = torch.from_numpy((rs.random_sample(B) + 1e-2).astype(np.float32)).cuda()
        if r % 4 == 3:         # the reference's call site: numpy operands (agent.py:100)
            eager.update_priorities(oe["tree_idxs"].cpu().numpy(), loss.cpu().numpy())
            lazy.update_priorities(ol["tree_idxs"].cpu().numpy(), loss.cpu().numpy())
        else:
            eager.update_priorities(oe["tree_idxs"], loss)
            # the reference's update is immediate (memory.py:157-159): a caller may reuse its tensors as soon as the call
            # returns.  The lazy write-back therefore keeps its own copy of device operands — overwrite the caller's at once
            idx_l, loss_l = ol["tree_idxs"].clone(), loss.clone()
            lazy.update_priorities(idx_l, loss_l)
            loss_l.fill_(77.0)
            idx_l.zero_()
        assert lazy._pending is not None and eager._pending is None
        if r == 1:             # an append in between applies the pending write-back first (it rewrites leaves and ancestors)
            for m in (eager, lazy):
                _fill(m, None, 3, seed=100 + r)
            assert lazy._pending is None
        elif r == 2:           # so does the header
            same()
            assert lazy._pending is None
        elif r == 4:           # ... and the raw handle (external C callers)
            v = torch.from_numpy(rs.random_sample(64) * lazy.transitions.total()).cuda()
            outs = []
            for m in (eager, lazy):
                p, di, ti = (torch.empty(64, dtype=torch.float32, device="cuda"), torch.empty(64, dtype=torch.int64, device="cuda"),
                             torch.empty(64, dtype=torch.int64, device="cuda"))
                L.check(hip, hip.rb_replay_find(m._h, v.data_ptr(), 64, p.data_ptr(), di.data_ptr(), ti.data_ptr(), m._stream()))
                outs.append((p, ti))
            assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
        elif r == 5:           # two updates in a row: the first is applied before the second is recorded
            loss2 = loss * 0.5 + 0.1
            eager.update_priorities(oe["tree_idxs"], loss2)
            lazy.update_priorities(ol["tree_idxs"], loss2)
    same()


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


def _fill_full(mem, cap, chunk, seed, actions=6):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rs = np.random.RandomState(seed)
    for lo in range(0, cap + chunk, chunk):    # cap + chunk appends: full, write head mid-buffer
        fr = torch.randint(0, 256, (chunk, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
        mem.append_batch(fr, rs.randint(0, actions, chunk), rs.choice([-1.0, 0.0, 1.0], size=chunk, p=[0.05, 0.9, 0.05]),
                         rs.random_sample(chunk) < 1e-3)
    levels, tree_start, tree_len = tree_geometry(cap)
    for r in range(40):        # non-uniform priorities through the public update path, 1024 leaves at a time
        idx = torch.randint(0, cap, (1024,), device="cuda", generator=g) + tree_start
        mem.update_priorities(idx, torch.rand(1024, device="cuda", generator=g) * 3 + 1e-3)
    return g


@pytest.fixture(scope="module")
def full_mem(hip):
    """BASELINE config 2/3 replay: 1M capacity (7 GB of HBM), filled once for every full-size test of this module."""
    from rainbow_amd.memory import ReplayMemory
    cap, chunk = 1_000_000, 50_000
    mem = ReplayMemory(_args(), cap, seed=6)
    _fill_full(mem, cap, chunk, 0)
    yield mem, chunk
    del mem
    torch.cuda.empty_cache()


def test_full_size_tree_properties(hip, full_mem):
    """BASELINE config 2 capacity (1M): size-independent invariants instead of an oracle replay."""
    mem, chunk = full_mem
    cap = mem.capacity
    hdr = mem._header()
    assert hdr.full == 1 and hdr.index == chunk
    levels, tree_start, tree_len = tree_geometry(cap)
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
    from oracle.replay_oracle import SumTreeOracle     # oracle search on the device's own tree: indices must be identical
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


def _check_sampler_against_oracle(hip, mem, batches, seed, beta=0.6):
    """sample_device with INJECTED uniforms on a full-size device replay against the oracle's draw on the downloaded tree:
    tree indices bit-exact (memory.py:64-82,124-131 — this is the sampler's multi-level descent, rb_descend_levels<5>/<4>
    on the 1M tree, <5>/<1> on the 100k tree), == rb_replay_find on the same sample values, and every scalar output of
    the batch (actions, n-step returns, nonterminals, IS weights, window slots + blanking) against the oracle."""
    from rainbow_amd import _lib as L
    ora = oracle_view_of_device_replay(mem, beta=beta)
    tr = ora.transitions
    rs = np.random.RandomState(seed)
    mem.priority_weight = beta
    for B in batches:
        for rep in range(3):
            uu = rs.random_sample((32, B))
            o = mem.sample_device(B, torch.from_numpy(uu))
            torch.cuda.synchronize()
            assert mem._header().last_status == 0
            probs, idxs, tree_idxs, attempts = ora.draw_indices(B, uu)
            got_ti = o["tree_idxs"].cpu().numpy()
            assert np.array_equal(got_ti, tree_idxs), "B=%d rep=%d" % (B, rep)
            assert mem._header().last_attempts == attempts
            # the plain one-level-per-step search (k_find) on the very sample values of the accepted attempt
            seg = np.float32(tr.total()) / np.float32(B)
            samples = (0.0 + np.float64(seg) * uu[attempts - 1]) + np.arange(B, dtype=np.int64) * np.float64(seg)
            v_d = torch.from_numpy(samples).cuda()
            pr, di, ti = (torch.empty(B, dtype=torch.float32, device="cuda"), torch.empty(B, dtype=torch.int64, device="cuda"),
                          torch.empty(B, dtype=torch.int64, device="cuda"))
            L.check(hip, hip.rb_replay_find(mem._h, v_d.data_ptr(), B, pr.data_ptr(), di.data_ptr(), ti.data_ptr(), mem._stream()))
            assert np.array_equal(ti.cpu().numpy(), got_ti)
            assert np.array_equal(pr.cpu().numpy(), probs)
            sc = ora.batch_scalars(idxs, probs)
            assert np.array_equal(o["actions"].cpu().numpy(), sc["actions"])
            np.testing.assert_allclose(o["returns"].cpu().numpy(), sc["returns"], rtol=1e-6, atol=1e-7)
            assert np.array_equal(o["nonterminals"].cpu().numpy(), sc["nonterminals"][:, 0])
            np.testing.assert_allclose(o["weights"].cpu().numpy(), sc["weights"], rtol=F32_ULP_RTOL)
            # pixels: every stack slot is the ring frame the oracle's window names, or zeros when blanked
            s, ns = o["states"].cpu().numpy(), o["next_states"].cpu().numpy()
            h, n = mem.history, mem.n
            for b in (0, B // 2, B - 1):
                for t in range(h):
                    for arr, slot in ((s, t), (ns, n + t)):
                        want = (np.zeros((84, 84), np.uint8) if sc["blank"][b, slot]
                                else mem._grab("frames", int(sc["ring"][b, slot]), 1)[0])
                        assert np.array_equal(arr[b, t], want), (B, b, t, slot)


def test_sampler_indices_match_oracle_at_1m(hip, full_mem):
    """the 5-levels-per-trip descent of the sampler is pinned to the oracle on the 1M tree."""
    mem, _ = full_mem
    _check_sampler_against_oracle(hip, mem, (32, 256), seed=77)


def test_per_loop_at_1m_keeps_the_tree_exact(hip, full_mem):
    """The PER-only loop of BASELINE's second metric on the 1M replay, as the bench runs it (device RNG, lazy write-backs: every
    update_priorities rides in the next sampler launch — k_update_sample on the 20-level tree, its one-wave sorted update and
    the LDS tree top it patches), with appends in between (they apply the pending write-back first): afterwards EVERY internal
    node is fl32(left + right) of its children, the leaves are what a numpy replay of the same write-backs holds (last write
    wins), and header.max / total agree.  Runs last among the full-size tests (it moves the write head)."""
    mem, _ = full_mem
    cap = mem.capacity
    levels, tree_start, tree_len = tree_geometry(cap)
    leaves = mem._grab("tree")[tree_start:].copy()
    vmax = float(mem._header().max)
    g = torch.Generator(device="cuda").manual_seed(123)
    rs = np.random.RandomState(5)
    B = 32
    pending = None
    for it in range(300):
        o = mem.sample_device(B)                       # applies the previous round's write-back in the same launch
        if pending is not None:
            idx, pr = pending
            for j in range(B):                          # memory.py:45,158: p = loss ** 0.5, last write wins
                leaves[idx[j] - tree_start] = pr[j]
            vmax = max(vmax, float(pr.max()))
        loss = torch.rand(B, device="cuda", generator=g) * 2 + 1e-3
        idx = o["tree_idxs"].cpu().numpy()
        assert np.all(np.diff(idx) >= 0), "a stratified draw never decreases"
        mem.update_priorities(o["tree_idxs"], loss)
        assert mem._pending is not None
        pending = (idx, np.power(loss.cpu().numpy(), np.float32(0.5)))
        if it % 25 == 24:                               # an append in between: the pending write-back goes first
            fr = torch.randint(0, 256, (7, 84, 84), dtype=torch.uint8, device="cuda", generator=g)
            idx0 = mem.transitions.index
            mem.append_batch(fr, rs.randint(0, 6, 7), np.zeros(7), np.zeros(7, dtype=bool))
            assert mem._pending is None
            for j in range(B):
                leaves[idx[j] - tree_start] = pending[1][j]
            vmax = max(vmax, float(pending[1].max()))
            pending = None
            for k in range(7):                          # memory.py:52-61: new transitions enter with the running maximum
                leaves[(idx0 + k) % cap] = np.float32(vmax)
    mem.flush()
    if pending is not None:
        for j in range(B):
            leaves[pending[0][j] - tree_start] = pending[1][j]
        vmax = max(vmax, float(pending[1].max()))
    tree = mem._grab("tree")
    np.testing.assert_allclose(tree[tree_start:], leaves, rtol=F32_ULP_RTOL)      # (loss ** 0.5: 1 ulp, helpers.py)
    internal = np.arange(0, tree_start)
    p = internal[2 * internal + 2 < tree_len]
    assert np.array_equal(tree[p], tree[2 * p + 1] + tree[2 * p + 2])
    hdr = mem._header()
    assert hdr.total == tree[0] and hdr.last_status == 0
    np.testing.assert_allclose(hdr.max, vmax, rtol=F32_ULP_RTOL)


def test_sampler_indices_match_oracle_at_100k_nstep20(hip):
    """BASELINE config 4's replay: C = 100k (tree depth 17: one 5-level trip + one 1-level trip), n = 20 windows."""
    from rainbow_amd.memory import ReplayMemory
    cap = 100_000
    mem = ReplayMemory(_args(multi_step=20), cap, seed=8)
    _fill_full(mem, cap, 20_000, 3)
    _check_sampler_against_oracle(hip, mem, (32, 256), seed=78)


def test_device_rng_sampler_frequencies(hip):
    """The PRODUCTION random path of the sampler (device Philox, no injected uniforms): over 10^4 batches on a small tree
    every stratum yields exactly one sample per batch (memory.py:125-129) and the empirical leaf frequencies follow
    priority / total (each leaf's count is Binomial: within 5 sigma, and a chi-square over the leaves)."""
    from rainbow_amd.memory import ReplayMemory
    cap, B, rounds = 256, 16, 10_000
    mem = ReplayMemory(_args(), cap, seed=12345)
    _fill(mem, None, cap + 40, seed=3, p_term=0.0)          # full; write head at 40
    tree_start = tree_geometry(cap)[1]
    rs = np.random.RandomState(4)
    pri = (rs.random_sample(cap) * 2 + 0.05).astype(np.float32)
    mem.update_priorities(np.arange(cap) + tree_start, pri)    # leaves = pri ** 0.5
    tree = mem._grab("tree")
    leaves = tree[tree_start:].astype(np.float64)
    total = float(tree[0])
    out = torch.empty(rounds, B, dtype=torch.int64, device="cuda")
    attempts = np.empty(rounds, dtype=np.int64)
    for r in range(rounds):
        o = mem.sample_device(B)
        out[r].copy_(o["tree_idxs"])
        if r % 500 == 0:
            assert mem._header().last_status == 0
    torch.cuda.synchronize()
    idx = out.cpu().numpy() - tree_start
    # one sample per stratum: sample i lies in [i*seg, (i+1)*seg) => its leaf's prefix bracket intersects the stratum
    csum = np.cumsum(leaves)
    seg = total / B
    lo, hi = csum[idx] - leaves[idx], csum[idx]
    i = np.arange(B)[None, :]
    tol = 1e-5 * total
    assert np.all(hi >= i * seg - tol) and np.all(lo <= (i + 1) * seg + tol)
    assert np.all(np.diff(idx, axis=1) >= 0)
    # frequencies: a batch is rejected as a whole when any leaf is within the write head's exclusion zone
    # (memory.py:131), which conditions the distribution; compare on leaves whose stratum never contains an excluded leaf
    head = 40
    bad = np.zeros(cap, dtype=bool)
    bad[[(head - k) % cap for k in range(0, 4)]] = True         # (index - idx) % C <= n
    bad[[(head + k) % cap for k in range(0, 4)]] = True         # (idx - index) % C < h
    counts = np.bincount(idx.ravel(), minlength=cap).astype(np.float64)
    # P(leaf j drawn by stratum i) = overlap(j, i) / seg; rejection is independent across strata, so conditioning on
    # "no stratum drew an excluded leaf" leaves the clean strata's distributions unchanged
    edges = np.concatenate([[0.0], csum])
    expect = np.zeros(cap)
    clean = np.ones(cap, dtype=bool)
    for s in range(B):
        a, b = s * seg, (s + 1) * seg
        ov = np.clip(np.minimum(edges[1:], b) - np.maximum(edges[:-1], a), 0, None)
        touched = ov > 0
        if np.any(bad & touched):
            clean[touched] = False
        else:
            expect += ov / seg
    sel = clean & (expect > 0)
    assert sel.sum() > cap // 2
    n_exp = expect[sel] * rounds
    sigma = np.sqrt(np.maximum(n_exp * (1 - np.minimum(expect[sel], 1.0)), 1.0))
    z = (counts[sel] - n_exp) / sigma
    assert np.max(np.abs(z)) < 5.0, float(np.max(np.abs(z)))
    chi2 = float(np.sum(z ** 2))
    dof = int(sel.sum())
    assert abs(chi2 - dof) < 6.0 * np.sqrt(2.0 * dof), (chi2, dof)
