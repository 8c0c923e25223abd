"""Host-side logic of the drop-in classes that needs no GPU."""
import numpy as np
import torch


def _f(x):
    return x.sign() * x.abs().sqrt()          # model.py:32-34


def test_eps_in_recovery_is_exact():
    """weight_epsilon = ger(eps_out, eps_in) (model.py:39) is what a reference checkpoint stores; load_state_dict must get
    the factorised vectors back bit-for-bit (a plain division is off by one ulp in ~10 % of the columns)."""
    from rainbow_amd.agent import Agent
    g = torch.Generator().manual_seed(5)
    plain_wrong = 0
    for (n_out, n_in) in ((64, 3136), (357, 512), (32, 576), (3, 7)):
        e_in, e_out = _f(torch.randn(n_in, generator=g)), _f(torch.randn(n_out, generator=g))
        e_w = torch.outer(e_out, e_in)
        got = Agent._recover_eps_in(e_w, e_out)
        assert torch.equal(got, e_in), (n_out, n_in, int((got != e_in).sum()))
        assert torch.equal(torch.outer(e_out, got), e_w)
        j = int(torch.argmax(e_out.abs()))
        plain_wrong += int(((e_w[j] / e_out[j]) != e_in).sum())
    assert plain_wrong > 0, "the naive quotient is expected to miss some columns (else this test proves nothing)"
    # all-zero eps_out (cannot come out of randn, but must not divide by zero)
    z = Agent._recover_eps_in(torch.zeros(4, 9), torch.zeros(4))
    assert torch.equal(z, torch.zeros(9))


def test_bench_work_tables_match_survey():
    """bench.py's per-step algorithmic figures are SURVEY §8d's: 3.352 / 26.68 / 0.481 GFLOP and the byte totals."""
    import bench
    want = {"pong-canonical-b32": (3.352e9, 6868842), "breakout-canonical-b256": (26.68e9, 6764190),
            "data-efficient-b32": (0.481e9, 828842)}
    for name, (flops, n_params) in want.items():
        fl, nb = bench.step_work(bench.CONFIGS[name], n_params)
        assert abs(fl / flops - 1) < 2e-3, (name, fl)
        assert nb > 40 * n_params
        tab = bench.kernel_table(bench.CONFIGS[name], n_params)
        assert {"clip_adam", "fc_h_fwd", "fc_h_bwd", "conv1_fwd", "conv2_fwd", "conv2_dx", "conv_dw_all"} <= set(tab)


def test_initial_parameters_follow_the_reference_distributions():
    """SURVEY a14 (model.py:25-30) and the Conv2d defaults behind model.py:56-62: sigma constants EXACT
    (std_init / sqrt(in) for weights, std_init / sqrt(out) for biases, as float32 of the Python-double quotient), mu and
    conv draws inside their uniform bounds and filling them (a uniform on [-b, b] has std b / sqrt(3)), padding untouched.
    The layout comes from the built library (a host-side query: no GPU)."""
    import ctypes
    import math
    import os
    import __graft_entry__ as g
    from rainbow_amd import _lib
    from rainbow_amd.agent import _query_layout, init_parameters_flat
    g.build()
    lib = _lib.declare(ctypes.CDLL(os.path.join(g.ROOT, "rainbow_amd", "librainbow_hip.so")))
    for arch, hidden, actions, hist in ((0, 512, 6, 4), (1, 256, 4, 4), (0, 64, 3, 2)):
        cfg = _lib.LearnerConfig(batch=32, atoms=51, actions=actions, history=hist, hidden=hidden, architecture=arch,
                                 multi_step=3, v_min=-10.0, v_max=10.0, discount=0.99)
        n_params = ctypes.c_int64(0)
        assert lib.rb_learner_sizes(ctypes.byref(cfg), ctypes.byref(n_params), None) == 0
        layout = _query_layout(lib, cfg, lib.rb_learner_param_layout)
        torch.manual_seed(11)
        flat = init_parameters_flat(layout, n_params.value, 0.1)
        covered = torch.zeros(n_params.value, dtype=torch.bool)
        shapes = {n: s for n, _o, s in layout}
        fan_in = None
        for name, off, shape in layout:
            numel = int(np.prod(shape))
            v = flat[off:off + numel]
            covered[off:off + numel] = True
            if name.startswith("convs"):
                if name.endswith("weight"):
                    fan_in = int(np.prod(shape[1:]))
                bound = 1.0 / math.sqrt(fan_in)
            else:
                layer, kind = name.split(".")
                out_f, in_f = shapes[layer + ".weight_mu"]
                if kind == "weight_sigma":
                    assert torch.equal(v, torch.full((numel,), 0.1 / math.sqrt(in_f), dtype=torch.float32)), name
                    continue
                if kind == "bias_sigma":
                    assert torch.equal(v, torch.full((numel,), 0.1 / math.sqrt(out_f), dtype=torch.float32)), name
                    continue
                bound = 1.0 / math.sqrt(in_f)
            assert float(v.abs().max()) <= bound, name
            if numel >= 512:          # the draw fills its interval: std of U(-b, b) = b / sqrt(3), max close to b
                assert abs(float(v.std()) / (bound / math.sqrt(3)) - 1) < 0.1, name
                assert float(v.abs().max()) > 0.95 * bound, name
                assert abs(float(v.mean())) < 4 * bound / math.sqrt(3 * numel), name
        assert float(flat[~covered].abs().sum()) == 0.0          # alignment padding stays zero
