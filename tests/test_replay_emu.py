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
