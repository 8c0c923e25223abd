"""Shared comparison helpers for the parity tests."""
import os

import numpy as np

import scenarios

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# float32 quantities that pass through a power function (p = loss^w, w = (N p)^-beta) are only
# defined to ~1 ulp even for the reference itself (numpy's SIMD powf differs from libm by machine);
# sums of such values in the tree inherit that.  4 float32 ulps relative:
F32_ULP_RTOL = 4 * 2.0 ** -23


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name)) as z:
        return {k: z[k] for k in z.files}


def assert_trace_matches(trace, golden, rtol=F32_ULP_RTOL, label=""):
    assert set(trace.keys()) == set(golden.keys()), (sorted(set(trace) ^ set(golden)))
    for key in sorted(golden.keys()):
        got, want = np.asarray(trace[key]), np.asarray(golden[key])
        assert got.shape == want.shape, "%s %s: shape %s vs %s" % (label, key, got.shape, want.shape)
        if scenarios.is_exact_key(key):
            assert np.array_equal(got, want), "%s %s: exact mismatch\n got %s\nwant %s" % (label, key, got.ravel()[:16], want.ravel()[:16])
        else:
            np.testing.assert_allclose(got.astype(np.float64), want.astype(np.float64), rtol=rtol, atol=0,
                                       err_msg="%s %s" % (label, key))


def assert_learn_trace_matches(trace, golden, label="", grad_rtol=2e-4, grad_atol_rel=5e-6, param_atol=2e-7):
    """Float tolerances of the learn step (stated): loss / norms 1e-5 relative;
    gradient elements rtol 2e-4 + 5e-6*max|g| (fp32 accumulation-order noise on cancelling sums);
    post-Adam parameters 2e-7 absolute (lr 6.25e-5 times an O(1e-3) relative update error);
    greedy actions exact."""
    assert set(trace.keys()) == set(golden.keys()), sorted(set(trace) ^ set(golden))[:10]
    for key in sorted(golden.keys()):
        got, want = np.asarray(trace[key]), np.asarray(golden[key])
        assert got.shape == want.shape, "%s %s: shape %s vs %s" % (label, key, got.shape, want.shape)
        if key.startswith("act_"):
            assert np.array_equal(got, want), "%s %s" % (label, key)
        elif "_grad/" in key:
            atol = grad_atol_rel * float(np.max(np.abs(want))) + 1e-12
            np.testing.assert_allclose(got, want, rtol=grad_rtol, atol=atol, err_msg="%s %s" % (label, key))
        elif "_param/" in key:
            np.testing.assert_allclose(got, want, rtol=0, atol=param_atol, err_msg="%s %s" % (label, key))
        elif key.endswith("grad_norm"):
            # the reference's own clip_grad_norm_ total is a float32 reduction over 3.2 M-element tensors: measured
            # 1.5e-5 .. 2.3e-5 away from the exact (float64) norm of its own gradients at the BASELINE shapes, while the
            # device norm is within 1e-7 of exact (tools/diag_shape.py) — so the bound is the reference's noise
            np.testing.assert_allclose(got, want, rtol=5e-5, atol=1e-7, err_msg="%s %s" % (label, key))
        else:  # loss, q values
            np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-7, err_msg="%s %s" % (label, key))


def oracle_view_of_device_replay(mem, beta=None):
    """A ReplayOracle whose sum-tree and scalar columns ARE the device's (downloaded), without the frame store: lets the
    oracle's index draw / window / scalar code (oracle.replay_oracle.draw_indices, batch_scalars) run against a
    full-size (1M) device replay, whose 7 GB of frames no host copy is made of."""
    from oracle.replay_oracle import ReplayOracle, SumTreeOracle, tree_geometry
    ora = ReplayOracle.__new__(ReplayOracle)
    ora.capacity, ora.history, ora.n = mem.capacity, mem.history, mem.n
    ora.discount = mem.discount
    ora.priority_weight = mem.priority_weight if beta is None else beta
    ora.priority_exponent = mem.priority_exponent
    ora.n_step_scaling = np.array([mem.discount ** i for i in range(mem.n)], dtype=np.float32)
    st = SumTreeOracle.__new__(SumTreeOracle)
    st.capacity = mem.capacity
    st.levels, st.tree_start, st.tree_len = tree_geometry(mem.capacity)
    st.tree = mem._grab("tree")
    st.timestep, st.action, st.reward = mem._grab("timestep"), mem._grab("action"), mem._grab("reward")
    st.nonterminal = mem._grab("nonterminal").astype(bool)
    st.frames = None
    hdr = mem._header()
    st.index, st.full, st.max = int(hdr.index), bool(hdr.full), np.float32(hdr.max)
    ora.transitions = st
    return ora
