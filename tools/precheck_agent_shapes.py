"""CPU pre-check of tests/test_learner_gpu.py::test_agent_default_flag_set_with_hosted_optimiser_pass_vs_oracle: the oracle side
of the test (no GPU) with the Agent's initial parameters reproduced from torch's CPU generator; prints the hidden-layer ReLU
margin of every step, so that an ill-conditioned seed is found here and not on the GPU box.  usage: python tools/precheck_agent_shapes.py [shape ...]"""
import ctypes as C
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_learner_gpu as T  # noqa: E402
from rainbow_amd import _lib as L  # noqa: E402
from rainbow_amd.agent import _query_layout, init_parameters_flat  # noqa: E402

lib = L.declare(C.CDLL(os.path.join(ROOT, "rainbow_amd", "librainbow_hip.so")))
for shape in (sys.argv[1:] or sorted(T.AGENT_SHAPES)):
    arch, hidden, B, A, n, cap, appends, _seed = T.AGENT_SHAPES[shape]
    args = T._args(device="cpu", architecture=arch, hidden_size=hidden, batch_size=B, multi_step=n)
    cfg = L.LearnerConfig(batch=B, atoms=51, actions=A, history=4, hidden=hidden, architecture=0 if arch == "canonical" else 1,
                          multi_step=n, v_min=-10.0, v_max=10.0, discount=0.99)
    n_params, n_noise = C.c_int64(0), C.c_int64(0)
    L.check(lib, lib.rb_learner_sizes(C.byref(cfg), C.byref(n_params), C.byref(n_noise)))
    layout = _query_layout(lib, cfg, lib.rb_learner_param_layout)
    torch.manual_seed(5)
    torch.randint(0, 2 ** 31 - 1, (1,))                      # Agent.__init__ draws the library seed first
    flat = init_parameters_flat(layout, n_params.value, 0.1)
    online = {name: flat[off:off + int(np.prod(s))].view(s).numpy().copy() for name, off, s in layout}
    t0 = time.time()
    want, _tree = T.agent_shape_oracle(shape, online, args)
    print(shape, "margins", ["%.1e" % w["margin"] for w in want], "norms", ["%.3g" % w["norm"] for w in want],
          "%.0f s" % (time.time() - t0), flush=True)
