"""Kernel-logic tests of the learn step on the host interpreter (CPU): the same kernel sources as
librainbow_hip.so, against the REAL reference's golden vectors (tests/golden/learn_*.npz)."""
import numpy as np
import pytest

import scenarios
from cabi_adapter import CAbiLearnAdapter, NumpyMem
from helpers import assert_learn_trace_matches, load_golden
from hipemu import loader
from oracle import learner_oracle as O


@pytest.fixture(scope="module")
def emu():
    return loader.load()


@pytest.mark.parametrize("name", ["atoms21", "dataeff", "canon"])
def test_learn_step_matches_reference_golden(emu, name):
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    trace = scenarios.learn_scenario(ad, name, O)
    assert_learn_trace_matches(trace, load_golden("learn_%s.npz" % name), label="emu/" + name)
    ad.close()


def test_generic_gemm_fallback_still_matches_golden(emu, monkeypatch):
    """RB_GENERIC_GEMM_ONLY=1 forces every contraction through gemm_core.h (the fallback used when the
    streamed noisy-linear kernels' alignment preconditions do not hold)."""
    monkeypatch.setenv("RB_GENERIC_GEMM_ONLY", "1")
    name = "atoms21"
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    trace = scenarios.learn_scenario(ad, name, O)
    assert_learn_trace_matches(trace, load_golden("learn_%s.npz" % name), label="emu-generic/" + name)
    ad.close()


def test_unit_conversion_is_exact():
    """rb_unit's multiply + Newton step equals the correctly rounded x/255 for every byte (memory.py:137)."""
    x = np.arange(256, dtype=np.float32)
    inv = np.float32(1.0 / 255.0)
    q = (x * inv).astype(np.float32)
    r = (np.float64(-255.0) * q.astype(np.float64) + x.astype(np.float64)).astype(np.float32)
    q2 = (r.astype(np.float64) * np.float64(inv) + q.astype(np.float64)).astype(np.float32)
    assert np.array_equal(q2, x / np.float32(255))
