"""Diagnosis: the world-N exchange test's per-rank gradients (before any exchange) against the per-rank oracle, step by step
(usage: python tools/diag_world8.py [world])."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenarios  # noqa: E402
from cabi_adapter import CAbiLearnAdapter, TorchMem  # noqa: E402
from oracle import learner_oracle as O  # noqa: E402
from rainbow_amd import _lib as L  # noqa: E402
from test_learner_gpu import BASELINE_SHAPES  # noqa: E402

hip = L.load()
shape = "cfg2-canonical-h512-b32-a6"
cfgd = BASELINE_SHAPES[shape]
scenarios.LEARN_CONFIGS[shape] = cfgd
cfg = O.Config(**cfgd)
hy = scenarios.LEARN_HYPER
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ads = [CAbiLearnAdapter(hip, TorchMem(), shape) for _ in range(world)]
online, target = O.init_params(cfg, 611), O.init_params(cfg, 612)
for ad in ads:
    ad.load(online, target)
adam = O.AdamOracle(online, hy["lr"], hy["adam_eps"])
draws = O.noise_draw_count(cfg)
for k in range(2):
    per = []
    for r, ad in enumerate(ads):
        rs = np.random.RandomState(100 + 10 * k + r)
        raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
        batch = scenarios.make_batch(cfgd, 200 + 10 * k + r)
        ad.reset_noise_online(raw_on)
        ad.learn_only(batch, raw_tg)
        per.append(O.learn(cfg, online, target, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg), batch))
    torch.cuda.synchronize()
    for r, ad in enumerate(ads):
        got = ad._unflat(ad.mem.download(ad.grads))
        loss = ad.mem.download(ad._loss)
        worst = []
        for n, g in per[r]["grads"].items():
            d = np.abs(got[n] - g).max() / (np.abs(g).max() + 1e-30)
            worst.append((d, n))
        worst.sort(reverse=True)
        print("step %d rank %d: loss maxrel %.2e   worst grad tensors: %s" % (
            k, r, np.abs(loss - per[r]["loss"]).max() / np.abs(per[r]["loss"]).max(),
            ", ".join("%s %.1e" % (n, d) for d, n in worst[:3])), flush=True)
    mean = ads[0].grads.clone()
    for ad in ads[1:]:
        mean += ad.grads
    mean /= world
    for ad in ads:
        ad.grads.copy_(mean)
        L.check(hip, hip.rb_learner_grads_modified(ad.h))
    outs = [ad.finish_step() for ad in ads]
    gmean = {n: sum(pr["grads"][n].astype(np.float64) for pr in per).astype(np.float32) / np.float32(world)
             for n in per[0]["grads"]}
    total, clipped = O.clip_grads(gmean, hy["norm_clip"])
    online = adam.step(clipped)
    p = ads[0].params()
    print("step %d: params max abs diff %.2e, norm %g vs %g" % (k, max(np.abs(p[n] - online[n]).max() for n in p),
                                                                outs[0]["grad_norm"], total))
