import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(tryfirst=True)
def pytest_cmdline_main(config):
    """The CPU suite (`-m "not gpu"`) is dominated by kernels running on the single-threaded host interpreter (tests/hipemu):
    spread it over worker processes with pytest-xdist when that is installed and the caller did not choose a worker count
    (RB_TEST_SERIAL=1 keeps one process).  The GPU suite stays in ONE process: one device, and the round-end harness records
    which shared objects that process loaded.  Everything the workers would otherwise build concurrently — the interpreter
    build of the kernels, the in-tree library test_abi.py inspects — is built here first, in the controller."""
    if hasattr(config, "workerinput") or os.environ.get("RB_TEST_SERIAL") == "1":
        return None
    if (config.option.markexpr or "").strip() != "not gpu" or not config.pluginmanager.hasplugin("xdist"):
        return None
    if getattr(config.option, "numprocesses", None) not in (None, 0) or getattr(config.option, "collectonly", False):
        return None
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests", "hipemu"))
        import build_emu
        build_emu.build()
        import __graft_entry__
        __graft_entry__.build()
    except Exception as e:          # a build failure is reported by the tests that need the artefact
        sys.stderr.write("conftest: pre-build failed: %r\n" % (e,))
    config.option.numprocesses = max(1, min(4, (os.cpu_count() or 2) // 2))
    return None


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with -m gpu; without a GPU they are skipped, never faked.
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
