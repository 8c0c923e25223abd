"""Pins the CPU oracle against golden vectors produced by the REAL reference
(tests/golden/make_golden.py).  CPU-only."""
import pytest

import scenarios
from adapters import OracleReplayAdapter
from helpers import assert_trace_matches, load_golden


@pytest.mark.parametrize("name", sorted(scenarios.REPLAY_CONFIGS))
def test_replay_oracle_matches_reference_golden(name):
    capacity, history, n, discount, omega, _ = scenarios.REPLAY_CONFIGS[name]
    trace = scenarios.replay_scenario(OracleReplayAdapter(capacity, history, n, discount, omega), name)
    assert_trace_matches(trace, load_golden("replay_%s.npz" % name), label="oracle/" + name)


from adapters import OracleLearnAdapter  # noqa: E402
from helpers import assert_learn_trace_matches  # noqa: E402
from oracle import learner_oracle as O  # noqa: E402


@pytest.mark.parametrize("name", sorted(scenarios.LEARN_CONFIGS))
def test_learn_oracle_matches_reference_golden(name):
    trace = scenarios.learn_scenario(OracleLearnAdapter(name), name, O)
    assert_learn_trace_matches(trace, load_golden("learn_%s.npz" % name), label="oracle/" + name)
