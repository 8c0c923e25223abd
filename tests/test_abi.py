"""CPU checks of the C-ABI boundary: the shipped library loads and exports every symbol that
include/rainbow_hip.h declares (no compute without a GPU), and the product path refuses to run
without the HIP library / a GPU (no silent fallback)."""
import ctypes
import os
import re
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    g.build()
    return os.path.join(ROOT, "rainbow_amd", "librainbow_hip.so")


def test_header_and_binding_table_agree():
    from rainbow_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "rainbow_hip.h")).read()
    declared = set(re.findall(r"\b(rb_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), sorted(declared ^ set(_lib.SIGNATURES))


def test_library_exports_every_declared_symbol(built):
    from rainbow_amd import _lib
    lib = ctypes.CDLL(built)
    _lib.declare(lib, strict=True)
    assert lib.rb_abi_version() == 1


def test_layout_queries_work_without_gpu(built):
    from rainbow_amd import _lib
    lib = _lib.declare(ctypes.CDLL(built))
    cfg = _lib.LearnerConfig(batch=32, atoms=51, actions=6, history=4, hidden=512, architecture=0, multi_step=3,
                             v_min=-10.0, v_max=10.0, discount=0.99)
    n_params, n_noise = ctypes.c_int64(0), ctypes.c_int64(0)
    assert lib.rb_learner_sizes(ctypes.byref(cfg), ctypes.byref(n_params), ctypes.byref(n_noise)) == 0
    # 6,868,842 reference parameters + alignment padding
    assert 6868842 <= n_params.value < 6868842 + 64 * 32
    n = ctypes.c_int32(0)
    assert lib.rb_learner_param_layout(ctypes.byref(cfg), None, ctypes.byref(n)) == 0 and n.value == 22
    bad = _lib.LearnerConfig(batch=0, atoms=51, actions=6, history=4, hidden=512, architecture=0, multi_step=3,
                             v_min=-10.0, v_max=10.0, discount=0.99)
    assert lib.rb_learner_sizes(ctypes.byref(bad), None, None) < 0
    assert b"batch" in lib.rb_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="CPU-only check")
def test_product_path_fails_loudly_without_gpu():
    from rainbow_amd.memory import ReplayMemory
    args = types.SimpleNamespace(device=torch.device("cpu"), history_length=4, discount=0.99, multi_step=3,
                                 priority_weight=0.4, priority_exponent=0.5)
    with pytest.raises(RuntimeError, match="HBM"):
        ReplayMemory(args, 128)
