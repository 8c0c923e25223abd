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
