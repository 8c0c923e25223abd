"""Kernel-logic tests of the replay path on the host interpreter (CPU).  These run the SAME
kernel sources as librainbow_hip.so (compiled with -DRB_HOST_INTERP) against the reference's
golden vectors; the GPU parity tests proper are tests/test_replay_gpu.py."""
import numpy as np
import pytest

import scenarios
from cabi_adapter import CAbiReplayAdapter, NumpyMem
from helpers import assert_trace_matches, load_golden
from hipemu import loader


@pytest.fixture(scope="module")
def emu():
    return loader.load()


@pytest.mark.parametrize("name", sorted(scenarios.REPLAY_CONFIGS))
def test_replay_kernels_match_reference_golden(emu, name):
    capacity, history, n, discount, omega, _ = scenarios.REPLAY_CONFIGS[name]
    ad = CAbiReplayAdapter(emu, NumpyMem(), capacity, history, n, discount, omega)
    trace = scenarios.replay_scenario(ad, name)
    assert_trace_matches(trace, load_golden("replay_%s.npz" % name), label="emu/" + name)
    ad.close()


@pytest.mark.parametrize("capacity", [6000, 10000, 20000, 40000, 70000])
def test_sampler_deep_tree_matches_oracle(emu, capacity):
    """Trees deeper than the LDS-cached top (4095 nodes = depths 0..11): the sampler's D-levels-per-round-trip search
    (rb_descend_levels<D>) and the bulk append / ancestor rebuild against the oracle, indices bit-exact.  Levels left
    under the LDS top and the instantiations they run: 6000 -> 2 (<2>); 10000 -> 3 (<3>); 20000 -> 4 (<4>);
    40000 -> 5 (<5>); 70000 -> 6 (<5> then <1>, the shape of BASELINE config 4's 100k tree; the 1M tree's <5> + <4> is
    pinned on the GPU, tests/test_replay_gpu.py::test_sampler_indices_match_oracle_at_1m)."""
    from oracle.replay_oracle import ReplayOracle
    rs = np.random.RandomState(capacity)
    ad = CAbiReplayAdapter(emu, NumpyMem(), capacity, 4, 3, 0.99, 0.5)
    ora = ReplayOracle(capacity)
    n = capacity + capacity // 3
    term = rs.random_sample(n) < 0.01
    ts = np.zeros(n, dtype=np.int32)
    t = 0
    for i in range(n):
        ts[i] = t
        t = 0 if term[i] else t + 1
    actions = rs.randint(0, 6, n).astype(np.int32)
    rewards = rs.choice([-1.0, 0.0, 1.0], size=n).astype(np.float32)
    frame_pool = rs.randint(0, 256, size=(64, 84, 84)).astype(np.uint8)
    for lo in range(0, n, 4000):
        hi = min(n, lo + 4000)
        frames = frame_pool[(np.arange(lo, hi) * 7) % 64]
        ad.append_batch(frames, ts[lo:hi], actions[lo:hi], rewards[lo:hi], (~term[lo:hi]).astype(np.uint8))
        ora.transitions.bulk_append(ts[lo:hi], frames, actions[lo:hi], rewards[lo:hi], ~term[lo:hi])
    ora.t = t
    tree_start = ora.transitions.tree_start
    for r in range(6):
        idx = rs.randint(0, capacity, 1024) + tree_start
        vals = (rs.random_sample(1024) * 4 + 1e-3).astype(np.float32)
        ad.update_leaves(idx, vals)
        ora.transitions.set_leaves(idx, vals)
    assert np.array_equal(ad.tree(), ora.transitions.tree)
    hdr = ad.raw_header()
    assert hdr.index == ora.transitions.index and bool(hdr.full) == ora.transitions.full
    assert hdr.max == ora.transitions.max and hdr.total == ora.transitions.total()
    for B in (32, 256):
        uu = rs.random_sample((16, B))
        got = ad.sample(B, uu, 0.6)
        ora.priority_weight = 0.6
        want = ora.sample_with_uniforms(B, uu)
        assert np.array_equal(got["tree_idxs"], want["tree_idxs"])
        assert np.array_equal(got["states"], want["states"]) and np.array_equal(got["next_states"], want["next_states"])
        assert np.array_equal(got["actions"], want["actions"])
        np.testing.assert_allclose(got["weights"], want["weights"], rtol=4 * 2.0 ** -23)
        assert got["attempts"] == want["attempts"]
    vals = rs.random_sample(512) * float(ora.transitions.total())
    p, di, ti = ad.find(vals)
    wp, wdi, wti = ora.transitions.find(vals)
    assert np.array_equal(ti, wti) and np.array_equal(p, wp)
    ad.close()


@pytest.mark.parametrize("capacity", [64, 1000, 6000, 70000])
def test_sorted_small_batch_update_matches_oracle(emu, capacity):
    """k_update's one-wave path for SORTED batches of at most 64 leaves (what ReplayMemory.sample hands to update_priorities:
    stratified draws never decrease) — replay_internal.h rb_update_sorted_wave — against SegmentTree.update (memory.py:44-48):
    runs of duplicate leaves (last write wins), sibling pairs, clusters that merge a few levels up, one leaf, 64 leaves, the
    first and the last leaf of the tree; then the same leaves shuffled (the hashed workgroup path) on a twin.  Tree, max and
    total bit-exact after every call."""
    from oracle.replay_oracle import ReplayOracle
    rs = np.random.RandomState(capacity + 1)
    ad = CAbiReplayAdapter(emu, NumpyMem(), capacity, 4, 3, 0.99, 0.5)
    twin = CAbiReplayAdapter(emu, NumpyMem(), capacity, 4, 3, 0.99, 0.5)
    ora, ora2 = ReplayOracle(capacity), ReplayOracle(capacity)
    tree_start = ora.transitions.tree_start
    every = np.arange(capacity) + tree_start
    for lo in range(0, capacity, 1024):                    # a non-trivial tree first (update_leaves takes up to 1024)
        idx = every[lo:lo + 1024]
        vals = (rs.random_sample(len(idx)) * 3 + 1e-3).astype(np.float32)
        for a in (ad, twin):
            a.update_leaves(idx, vals)
        for o in (ora, ora2):
            o.transitions.set_leaves(idx, vals)
    assert np.array_equal(ad.tree(), ora.transitions.tree)
    cases = [np.array([0]), np.array([capacity - 1]), np.array([0, 1]), np.array([0, capacity - 1]),
             np.array([5, 5, 5, 6, 6, 7]) % capacity, np.arange(64) % capacity, np.full(64, capacity // 2)]
    for n in (1, 2, 7, 32, 33, 64):
        cases.append(np.sort(rs.randint(0, capacity, n)))                                   # spread out
        c0 = int(rs.randint(0, capacity))
        cases.append(np.sort((c0 + rs.randint(0, min(capacity, 2 * n), n)) % capacity))     # clustered: paths meet early
    unsorted_calls = 0
    for data_idx in cases:
        idx = np.sort(np.asarray(data_idx, dtype=np.int64)) + tree_start
        vals = (rs.random_sample(len(idx)) * 5 + 1e-3).astype(np.float32)
        ad.update_leaves(idx, vals)
        ora.transitions.set_leaves(idx, vals)
        assert np.array_equal(ad.tree(), ora.transitions.tree), data_idx
        hdr = ad.raw_header()
        assert hdr.max == ora.transitions.max and hdr.total == ora.transitions.total()
        # the same leaves in another order: the hashed workgroup path (and another winner among duplicates)
        perm = rs.permutation(len(idx))
        unsorted_calls += int(not np.all(np.diff(idx[perm]) >= 0))
        twin.update_leaves(idx[perm], vals[perm])
        ora2.transitions.set_leaves(idx[perm], vals[perm])
        assert np.array_equal(twin.tree(), ora2.transitions.tree), data_idx
    assert unsorted_calls >= 8
    # priorities (loss ** omega) through the sorted path
    idx = np.sort(rs.randint(0, capacity, 32)) + tree_start
    loss = (rs.random_sample(32) + 1e-2).astype(np.float32)
    ad.update_priorities(idx, loss)
    ora.update_priorities(idx, loss)
    np.testing.assert_allclose(ad.tree(), ora.transitions.tree, rtol=4 * 2.0 ** -23)
    ad.close(); twin.close()


@pytest.mark.parametrize("capacity,n", [(6000, 3), (2000, 20)])
def test_update_and_sample_in_one_launch_equals_the_two_calls(emu, capacity, n):
    """k_update_sample (replay.hip) against rb_replay_update_priorities + rb_replay_sample on a twin (scenarios.py)."""
    scenarios.update_sample_twin_check(lambda c, h, nn: CAbiReplayAdapter(emu, NumpyMem(), c, h, nn, 0.99, 0.5), capacity=capacity, n=n)


def test_create_rejects_what_the_reference_cannot_run(emu):
    """Odd capacities crash the reference's sum-tree walk (IndexError in _propagate_index) and a window
    longer than 64 slots does not fit the sampler's masks: both are refused with an error code and a message."""
    import ctypes as C
    from rainbow_amd import _lib as L
    for cap, h, n in ((501, 4, 3), (0, 4, 3), (512, 0, 3), (512, 4, 0), (512, 16, 49)):
        handle = C.c_void_p()
        rc = emu.rb_replay_create(C.byref(handle), cap, h, n, 0.99, 0.5, 1)
        assert rc != 0 and not handle.value, (cap, h, n)
        assert emu.rb_last_error()
    with pytest.raises(L.RainbowError):
        L.check(emu, emu.rb_replay_create(C.byref(C.c_void_p()), 501, 4, 3, 0.99, 0.5, 1))


def test_earlier_valid_batch_survives_a_failed_draw(emu):
    """scenarios.earlier_valid_batch_survives_failed_draw_check on the host interpreter (the GPU runs the same check)."""
    scenarios.earlier_valid_batch_survives_failed_draw_check(emu, NumpyMem())


def test_sampler_gives_up_harmlessly(emu):
    """scenarios.sampler_gives_up_check on the host interpreter (the same check runs on the GPU in test_replay_gpu.py), plus:
    the host mirror of the write position follows a raw header restore."""
    import ctypes as C
    from rainbow_amd import _lib as L
    ad = scenarios.sampler_gives_up_check(emu, NumpyMem())
    idx, full = C.c_int64(-1), C.c_int32(-1)
    hdr = ad.raw_header()
    hdr.index = 6
    raw = np.frombuffer(bytes(hdr), dtype=np.uint8).copy()
    L.check(emu, emu.rb_copy_to_device(ad.bufs.header_dev, raw.ctypes.data, raw.nbytes, None))
    L.check(emu, emu.rb_replay_position(ad.h, C.byref(idx), C.byref(full)))
    assert (idx.value, full.value) == (6, 1)
    ad.close()
