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


def test_no_flat_or_scratch_instructions_in_the_gfx950_code():
    """The device code of every translation unit, compiled to gfx950 assembly: no FLAT instruction (a pointer the compiler
    could not place in an address space: the load also probes the LDS and scratch apertures) and no scratch segment (a
    spilling kernel slows every kernel that shares the stream with it) — both crept in twice in round 4
    (a select between an LDS value and a global one in the sampler; the hosted optimiser pass over its hosting kernel's
    register budget) and cost up to 10 us per step before anybody looked."""
    import shutil
    import subprocess
    import tempfile
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not (os.path.exists(hipcc) or shutil.which(hipcc)):
        pytest.skip("no hipcc")
    csrc = os.path.join(ROOT, "rainbow_amd", "csrc")
    procs = []
    with tempfile.TemporaryDirectory() as tmp:
        for tu in ("learner", "replay", "common"):
            out = os.path.join(tmp, tu + ".s")
            procs.append((tu, out, subprocess.Popen([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                                                     "--cuda-device-only", "-S", os.path.join(csrc, tu + ".hip"), "-o", out],
                                                    stderr=subprocess.DEVNULL)))
        for tu, out, pr in procs:
            assert pr.wait() == 0, tu
            kernel, offenders = None, []
            for line in open(out):
                m = re.match(r"^([_A-Za-z0-9]+):", line)
                if m:
                    kernel = m.group(1)
                body = line.split(";")[0]
                if re.search(r"\b(flat_load|flat_store|flat_atomic|scratch_load|scratch_store)", body):
                    offenders.append((kernel, body.split()[0]))
                m = re.match(r"^\s*\.amdhsa_private_segment_fixed_size\s+(\d+)", line)
                if m and int(m.group(1)) != 0:
                    offenders.append((kernel, "scratch segment of %s bytes" % m.group(1)))
            assert not offenders, (tu, offenders[:5])
