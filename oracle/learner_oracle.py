"""CPU oracle for the Rainbow learn step — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional float32 restatement (torch-CPU tensors + autograd, explicit parameter
dictionaries, no nn.Module) of the arithmetic in the reference's model.py and
agent.py:61-100 (/root/reference; cited per function).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Parity status: PINNED.  tests/golden/learn_*.npz hold losses, gradient norms, gradient
samples and post-Adam parameters produced by the REAL reference Agent.learn()
(tests/golden/make_golden_learn.py) for seeded parameters, batches and injected noise;
tests/test_oracle_golden.py replays them through this oracle.

Floating point: everything is float32; results agree with the reference to accumulation
order (GEMM/conv summation order is not specified by torch), i.e. ~1e-6 relative.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

LAYERS = ("fc_h_v", "fc_h_a", "fc_z_v", "fc_z_a")   # reset_noise order, model.py:82-85 / :64-67


class Config:
    def __init__(self, batch=32, atoms=51, actions=6, history=4, hidden=512, architecture="canonical",
                 multi_step=3, v_min=-10.0, v_max=10.0, discount=0.99, noisy_std=0.1):
        self.batch, self.atoms, self.actions, self.history, self.hidden = batch, atoms, actions, history, hidden
        self.architecture, self.multi_step = architecture, multi_step
        self.v_min, self.v_max, self.discount, self.noisy_std = v_min, v_max, discount, noisy_std

    @property
    def convs(self):
        """(out_ch, kernel, stride) per conv layer and the flattened feature size (model.py:55-63)."""
        if self.architecture == "canonical":
            return [(32, 8, 4), (64, 4, 2), (64, 3, 1)], 3136
        return [(32, 5, 5), (64, 5, 5)], 576

    def linear_shapes(self):
        _, feat = self.convs
        return {"fc_h_v": (self.hidden, feat), "fc_h_a": (self.hidden, feat),
                "fc_z_v": (self.atoms, self.hidden), "fc_z_a": (self.actions * self.atoms, self.hidden)}


def param_shapes(cfg):
    """State-dict names and shapes in nn.Module registration order (model.py:56-67, 16-21)."""
    shapes = []
    cin = cfg.history
    convs, _ = cfg.convs
    for i, (cout, k, _s) in enumerate(convs):
        shapes.append(("convs.%d.weight" % (2 * i), (cout, cin, k, k)))
        shapes.append(("convs.%d.bias" % (2 * i), (cout,)))
        cin = cout
    for name, (out_f, in_f) in cfg.linear_shapes().items():
        shapes.append((name + ".weight_mu", (out_f, in_f)))
        shapes.append((name + ".weight_sigma", (out_f, in_f)))
        shapes.append((name + ".bias_mu", (out_f,)))
        shapes.append((name + ".bias_sigma", (out_f,)))
    return shapes


def init_params(cfg, seed):
    """Seeded parameters with the reference's init *distributions* (model.py:25-30; torch's
    Conv2d default U(+-1/sqrt(fan_in))).  Used to build identical networks everywhere."""
    rs = np.random.RandomState(seed)
    out = {}
    for name, shape in param_shapes(cfg):
        if name.startswith("convs"):
            fan_in = int(np.prod(shape[1:])) if len(shape) == 4 else None
            if fan_in is None:  # bias: fan_in of the matching weight
                w = out[name.replace("bias", "weight")]
                fan_in = int(np.prod(w.shape[1:]))
            bound = 1.0 / math.sqrt(fan_in)
            out[name] = rs.uniform(-bound, bound, size=shape).astype(np.float32)
        else:
            layer, kind = name.split(".")
            out_f, in_f = cfg.linear_shapes()[layer]
            if kind in ("weight_mu", "bias_mu"):
                bound = 1.0 / math.sqrt(in_f)                                   # model.py:26-29
                out[name] = rs.uniform(-bound, bound, size=shape).astype(np.float32)
            elif kind == "weight_sigma":
                out[name] = np.full(shape, cfg.noisy_std / math.sqrt(in_f), dtype=np.float32)   # model.py:28
            else:
                out[name] = np.full(shape, cfg.noisy_std / math.sqrt(out_f), dtype=np.float32)  # model.py:30
    return out


def noise_draw_count(cfg):
    return sum(in_f + out_f for (out_f, in_f) in cfg.linear_shapes().values())


def scale_noise(x):
    """f(x) = sign(x) * sqrt(|x|)  (model.py:32-34)."""
    x = np.asarray(x, dtype=np.float32)
    return (np.sign(x) * np.sqrt(np.abs(x))).astype(np.float32)


def make_noise(cfg, raw_normals):
    """Splits N(0,1) draws in the reference's consumption order — per layer randn(in) then
    randn(out), layers fc_h_v, fc_h_a, fc_z_v, fc_z_a (model.py:36-38, 82-85) — into the
    factorised vectors {layer: (f(eps_in), f(eps_out))}."""
    raw = np.asarray(raw_normals, dtype=np.float32)
    assert raw.size == noise_draw_count(cfg)
    out, p = {}, 0
    for layer in LAYERS:
        out_f, in_f = cfg.linear_shapes()[layer]
        e_in = scale_noise(raw[p:p + in_f]); p += in_f
        e_out = scale_noise(raw[p:p + out_f]); p += out_f
        out[layer] = (e_in, e_out)
    return out


def _t(a):
    return torch.as_tensor(np.ascontiguousarray(a))


def noisy_linear(x, params, layer, noise):
    """NoisyLinear.forward (model.py:42-46); noise=None is eval mode (mu only)."""
    w, b = params[layer + ".weight_mu"], params[layer + ".bias_mu"]
    if noise is not None:
        e_in, e_out = (_t(v) for v in noise[layer])
        eps_w = torch.outer(e_out, e_in)                                       # model.py:39
        w = w + params[layer + ".weight_sigma"] * eps_w                        # model.py:44
        b = b + params[layer + ".bias_sigma"] * e_out
    return x @ w.t() + b


def forward(cfg, params, noise, x, log=False, probe=None, hidden_mask=None):
    """DQN.forward (model.py:69-80).  x float32 [B,h,84,84] in [0,1]; params: torch tensors.
    probe (optional dict): receives 'hidden_relu_margin' = the smallest |pre-activation| of the two hidden layers (see learn)
    and 'hidden_pre' = the two pre-activation matrices (numpy).
    hidden_mask (optional, test hook): boolean [B, 2H] — the hidden layers' ReLU decisions (value stream | advantage stream) are
    TAKEN from it instead of from the sign of this forward's own pre-activations (see learn)."""
    convs, feat = cfg.convs
    for i, (_c, _k, stride) in enumerate(convs):
        x = F.relu(F.conv2d(x, params["convs.%d.weight" % (2 * i)], params["convs.%d.bias" % (2 * i)], stride=stride))
    x = x.reshape(-1, feat)                                                    # model.py:71
    pre_v, pre_a = noisy_linear(x, params, "fc_h_v", noise), noisy_linear(x, params, "fc_h_a", noise)
    if probe is not None:
        probe["hidden_relu_margin"] = float(torch.minimum(pre_v.detach().abs().min(), pre_a.detach().abs().min()))
        probe["hidden_pre"] = np.concatenate([pre_v.detach().numpy(), pre_a.detach().numpy()], axis=1)
    if hidden_mask is not None:
        mk = _t(np.asarray(hidden_mask)).to(torch.float32)
        H = pre_v.shape[1]
        h_v, h_a = pre_v * mk[:, :H], pre_a * mk[:, H:]                        # relu with the given decisions (value and gradient)
    else:
        h_v, h_a = F.relu(pre_v), F.relu(pre_a)
    v = noisy_linear(h_v, params, "fc_z_v", noise)                             # model.py:72
    a = noisy_linear(h_a, params, "fc_z_a", noise)                             # model.py:73
    v = v.reshape(-1, 1, cfg.atoms)
    a = a.reshape(-1, cfg.actions, cfg.atoms)
    q = v + a - a.mean(1, keepdim=True)                                        # model.py:75
    return F.log_softmax(q, dim=2) if log else F.softmax(q, dim=2)             # model.py:76-79


def support(cfg):
    return torch.linspace(cfg.v_min, cfg.v_max, cfg.atoms)                     # agent.py:18


def project(cfg, pns_a, returns, nonterminals):
    """C51 projection (agent.py:79-92).  Deterministic per-row accumulation instead of the
    reference's two flat index_add_ calls (same sums up to addition order)."""
    B, Z = pns_a.shape
    z = support(cfg)
    delta_z = (cfg.v_max - cfg.v_min) / (Z - 1)                                # agent.py:19 (python double)
    Tz = returns.unsqueeze(1) + nonterminals.reshape(B, 1) * (cfg.discount ** cfg.multi_step) * z.unsqueeze(0)  # agent.py:79
    Tz = Tz.clamp(min=cfg.v_min, max=cfg.v_max)                                # agent.py:80
    b = (Tz - cfg.v_min) / delta_z                                             # agent.py:82
    l, u = b.floor().to(torch.int64), b.ceil().to(torch.int64)                 # agent.py:83
    l = torch.where((u > 0) & (l == u), l - 1, l)                              # agent.py:85
    u = torch.where((l < (Z - 1)) & (l == u), u + 1, u)                        # agent.py:86
    m = torch.zeros(B, Z, dtype=torch.float32)
    lo = pns_a * (u.float() - b)                                               # agent.py:91
    hi = pns_a * (b - l.float())                                               # agent.py:92
    for i in range(B):
        m[i].index_add_(0, l[i], lo[i])
        m[i].index_add_(0, u[i], hi[i])
    return m, l, u, b


def learn(cfg, online, target, noise_online, noise_target, batch, hidden_mask=None):
    """Agent.learn up to and including backward (agent.py:63-96).
    online/target: {name: np.float32 array}; batch: dict(states u8[B,h,84,84], next_states u8,
    actions i64[B], returns f32[B], nonterminals f32[B] or [B,1], weights f32[B]).
    Returns numpy results incl. UNCLIPPED gradients, and `hidden_relu_margin`: the smallest |pre-activation| of the two
    hidden layers in the forward that is differentiated.  relu'(x) is a step at 0: where a pre-activation (a 3136-term f32 dot
    product, rounding noise ~1e-7) is closer to zero than that noise, its SIGN — and with it one sample's whole contribution
    to that unit's gradients, up to 1/B of their magnitude — depends on the summation order; the reference itself (model.py:72-73,
    torch's GEMM blocking) is only defined up to that order there.  Parity tests on fixed seeds require a margin above the
    noise so that a gradient mismatch can never be this discontinuity.
    hidden_mask (optional, test hook; boolean [B, 2H]): the ReLU decisions of the differentiated forward are taken from it (what
    another implementation of the same forward decided) instead of from this forward's own signs.  The result then carries
    `hidden_mask_flips` = how many of its own decisions differ and `hidden_mask_flip_abs` = the largest |pre-activation| among
    those: a parity test at batch 256 (262 144 pre-activations per step, the smallest ~3e-8, a split-K GEMM's summation noise
    ~5e-8) uses it to show that every difference it tolerates IS this discontinuity — decisions that differ only where the
    pre-activation is inside the rounding noise — and nothing else."""
    B = batch["states"].shape[0]
    p_on = {k: _t(v).clone().requires_grad_(True) for k, v in online.items()}
    p_tg = {k: _t(v) for k, v in target.items()}
    states = _t(batch["states"]).to(torch.float32).div(255)                    # memory.py:137
    next_states = _t(batch["next_states"]).to(torch.float32).div(255)          # memory.py:138
    actions = _t(batch["actions"]).to(torch.int64)
    returns = _t(batch["returns"]).to(torch.float32)
    nonterminals = _t(batch["nonterminals"]).to(torch.float32).reshape(B)
    weights = _t(batch["weights"]).to(torch.float32)

    probe = {}
    log_ps = forward(cfg, p_on, noise_online, states, log=True, probe=probe, hidden_mask=hidden_mask)   # agent.py:66
    log_ps_a = log_ps[torch.arange(B), actions]                                # agent.py:67
    with torch.no_grad():
        pns = forward(cfg, p_on, noise_online, next_states)                    # agent.py:71
        a_star = (support(cfg).expand_as(pns) * pns).sum(2).argmax(1)          # agent.py:72-73
        pns_t = forward(cfg, p_tg, noise_target, next_states)                  # agent.py:75
        pns_a = pns_t[torch.arange(B), a_star]                                 # agent.py:76
        m, l, u, b = project(cfg, pns_a, returns, nonterminals)
    loss = -torch.sum(m * log_ps_a, 1)                                         # agent.py:94
    (weights * loss).mean().backward()                                         # agent.py:96
    grads = {k: v.grad.numpy().copy() for k, v in p_on.items()}
    return dict(loss=loss.detach().numpy().copy(), m=m.numpy().copy(), a_star=a_star.numpy().copy(),
                pns_a=pns_a.numpy().copy(), log_ps_a=log_ps_a.detach().numpy().copy(), grads=grads,
                l=l.numpy().copy(), u=u.numpy().copy(), hidden_relu_margin=probe["hidden_relu_margin"],
                **(_mask_flips(probe["hidden_pre"], hidden_mask) if hidden_mask is not None else {}))


def _mask_flips(pre, mask):
    own = pre > 0
    diff = own != np.asarray(mask, dtype=bool)
    return dict(hidden_mask_flips=int(diff.sum()), hidden_mask_flip_abs=float(np.abs(pre[diff]).max()) if diff.any() else 0.0)


def clip_grads(grads, max_norm):
    """clip_grad_norm_ (agent.py:97): 2-norm of the per-tensor 2-norms; scale by
    min(1, max_norm / (total + 1e-6)).  Returns (total_norm, clipped grads)."""
    norms = torch.stack([torch.linalg.vector_norm(_t(g)) for g in grads.values()])
    total = torch.linalg.vector_norm(norms)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return float(total), {k: (_t(g) * coef).numpy() for k, g in grads.items()}


class AdamOracle:
    """torch.optim.Adam(lr, eps) as the reference constructs it (agent.py:46), stepping a
    dict of numpy parameters with externally supplied (already clipped) gradients."""

    def __init__(self, params, lr, eps):
        self.t = {k: _t(v).clone().requires_grad_(True) for k, v in params.items()}
        self.opt = torch.optim.Adam(list(self.t.values()), lr=lr, eps=eps)

    def step(self, grads):
        for k, p in self.t.items():
            p.grad = _t(grads[k]).clone()
        self.opt.step()                                                        # agent.py:98
        return {k: p.detach().numpy().copy() for k, p in self.t.items()}


def act(cfg, params, noise, state):
    """Agent.act / evaluate_q (agent.py:53-55, 110-112) -> (argmax action, its expected value)."""
    with torch.no_grad():
        p = {k: _t(v) for k, v in params.items()}
        ps = forward(cfg, p, noise, _t(state).to(torch.float32).unsqueeze(0))
        q = (ps * support(cfg)).sum(2)
        return int(q.argmax(1).item()), float(q.max(1)[0].item())
