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
