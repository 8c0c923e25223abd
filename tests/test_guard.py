"""Out-of-bounds write hunt: the hot path driven through the C ABI with guard bands around every
caller-owned buffer (tests/guarded_mem.py) AND around every allocation the library owns (RB_GUARD=1,
rb_debug_check_guards).  The run happens in a child process because RB_GUARD is read when the library first allocates."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, timeout):
    env = dict(os.environ, RB_GUARD="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "guard_run.py"), *argv], env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0 and "guard run ok" in p.stdout, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    return p.stdout


def test_canaries_catch_a_stray_write():
    from guarded_mem import GuardedNumpyMem
    m = GuardedNumpyMem()
    a = m.empty((10,), np.float32)
    b = m.empty((7,), np.int64)
    assert m.check() == (2, [])
    np.lib.stride_tricks.as_strided(a, shape=(11,), strides=(4,))[10] = 1.0      # one element past the end
    assert m.check()[1] == ["float32(10,)"]
    np.lib.stride_tricks.as_strided(b[0:1], shape=(2,), strides=(-8,))[1] = 3    # one element before the start
    assert sorted(m.check()[1]) == ["float32(10,)", "int64(7,)"]


def test_host_interpreted_kernels_stay_inside_their_buffers():
    out = _run("emu", timeout=1500)
    assert "overwritten guard bands 0" in out


@pytest.mark.gpu
def test_hip_kernels_stay_inside_their_buffers_at_baseline_shapes():
    """3 learn steps at BASELINE cfg 2 / 3 / 4 shapes, act_batch(4096), single act, the factored exchange between two
    handles, replay append / sample / update / find at C = 4096 and 100k (n = 20, B = 256)."""
    out = _run("hip", timeout=1500)
    assert "overwritten guard bands 0" in out and "g-cfg3" in out
