"""Per-tensor comparison of one learn step at a BASELINE shape against the oracle (float64 norms): where a norm / gradient
difference comes from.  usage: python tools/diag_shape.py cfg3-canonical-h512-b256-a4"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import scenarios
from test_learner_gpu import BASELINE_SHAPES
from oracle import learner_oracle as O
from cabi_adapter import CAbiLearnAdapter, TorchMem
from rainbow_amd import _lib
shape = sys.argv[1]
cfgd = BASELINE_SHAPES[shape]
scenarios.LEARN_CONFIGS[shape] = cfgd
cfg = O.Config(**cfgd)
ad = CAbiLearnAdapter(_lib.load(), TorchMem(), shape)
ad.hy = dict(ad.hy, norm_clip=1e9)
online, target = O.init_params(cfg, 901), O.init_params(cfg, 902)
ad.load(online, target)
draws = O.noise_draw_count(cfg)
rs = np.random.RandomState(55)
raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
ad.reset_noise_online(raw_on)
batch = scenarios.make_batch(cfgd, 700)
got = ad.learn_step(batch, raw_tg)
want = O.learn(cfg, online, target, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg), batch)
tot_g = tot_w = 0.0
for k, g in want["grads"].items():
    a, b = got["grads"][k].astype(np.float64), g.astype(np.float64)
    ng, nw = np.sqrt((a * a).sum()), np.sqrt((b * b).sum())
    tot_g += (a * a).sum(); tot_w += (b * b).sum()
    print("%-22s |g| %.6e ours %.6e rel %+.2e  max|d| %.2e  max|g| %.2e  mean signed rel %+.2e" % (
        k, nw, ng, ng / nw - 1, np.abs(a - b).max(), np.abs(b).max(), ((a - b) * b).sum() / (b * b).sum()))
print("total f64: want %.8e ours %.8e rel %+.2e ; device norm %.8e ; oracle clip_grads norm %.8e" % (
    np.sqrt(tot_w), np.sqrt(tot_g), np.sqrt(tot_g / tot_w) - 1, got["grad_norm"], O.clip_grads(want["grads"], 1e9)[0]))
print("loss rel", np.abs(got["loss"] / want["loss"] - 1).max())
