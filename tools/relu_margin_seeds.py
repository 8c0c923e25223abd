"""usage: python tools/relu_margin_seeds.py [n_seeds] — VERDICT r5 weak 1: how restrictive is the guard `hidden_relu_margin > 3e-8` that the
fixed-seed batch-32 parity tests assert?  For n seeds: random-init canonical parameters (model.py:25-30), fresh noise, a random batch of
32 uint8 frame stacks; the smallest |pre-activation| of the two hidden layers in the differentiated forward (oracle, CPU).  Prints the
count above the guard and the distribution."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import learner_oracle as O  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
cfg = O.Config(batch=32, atoms=51, actions=6, history=4, hidden=512, architecture="canonical", multi_step=3)
margins = []
for seed in range(n):
    rs = np.random.RandomState(1000 + seed)
    params = {k: torch.as_tensor(v) for k, v in O.init_params(cfg, 5000 + seed).items()}
    noise = O.make_noise(cfg, rs.randn(O.noise_draw_count(cfg)).astype(np.float32))
    x = torch.as_tensor(rs.randint(0, 256, size=(32, 4, 84, 84)).astype(np.float32) / np.float32(255))
    probe = {}
    with torch.no_grad():
        O.forward(cfg, params, noise, x, log=True, probe=probe)
    margins.append(probe["hidden_relu_margin"])
m = np.array(margins)
print("seeds %d; margin > 3e-8 (the tests' guard): %d; > 1e-8: %d; > 4e-9 (the f32 dot product's own rounding noise): %d" %
      (n, int((m > 3e-8).sum()), int((m > 1e-8).sum()), int((m > 4e-9).sum())))
print("min %.2e  median %.2e  max %.2e  (32768 pre-activations per seed)" % (m.min(), np.median(m), m.max()))
