"""Drop-in `Agent` for the reference's main.py / test.py (replaces /root/reference/agent.py + model.py).

Same constructor and methods as the reference class (agent.py:12-118).  The networks are not
nn.Modules: online / target parameters, gradients and factorised noise are flat float32 HBM
buffers (torch tensors used purely as memory owners) and every forward / backward kernel is
hand-written HIP behind librainbow_hip.so, the clip + Adam update included (`self.optimiser` is a
torch.optim.Adam whose step() runs the library's one-pass kernel, agent.py:46,97-98); PyTorch-ROCm
owns memory, streams and torch.distributed.

learn(mem):  rainbow_amd.memory.ReplayMemory  -> fully device-resident step, no host sync:
                 sample (+ noise) -> 3 forwards + projection + loss + backward (+ priority
                 write-back) -> [RCCL all-reduce] -> global-norm clip + Adam
             any other object with the reference's sample()/update_priorities() -> compat path.

Multi-GPU (BASELINE config 5): one process per GPU, identical parameters, each replica owns its
replay; between backward and clip every replica obtains the mean gradient of the global batch
(rainbow_amd/dist.py: all-gather of the FC gradient FACTORS + all-reduce of the conv gradients by
default, or one all-reduce of the flat buffer).  Enabled automatically when torch.distributed is
initialised.
"""
import ctypes as C
import math
import os
import weakref

import numpy as np
import torch

from . import _lib as L
from . import dist as rdist
from .memory import ReplayMemory

_LAYERS = ("fc_h_v", "fc_h_a", "fc_z_v", "fc_z_a")


def _query_layout(lib, cfg, fn):
    n = C.c_int32(0)
    L.check(lib, fn(C.byref(cfg), None, C.byref(n)))
    descs = (L.TensorDesc * n.value)()
    L.check(lib, fn(C.byref(cfg), descs, C.byref(n)))
    return [(d.name.decode(), int(d.offset), tuple(d.shape[i] for i in range(d.ndim))) for d in descs[:n.value]]


def init_parameters_flat(layout, n_params, noisy_std):
    """The reference's initial parameter distributions over the flat layout (a CPU float32 tensor; `layout` = the
    (name, offset, shape) triples of rb_learner_param_layout): Conv2d's default U(+-1/sqrt(fan_in)) for weight and bias
    (model.py:56-62, torch.nn.Conv2d.reset_parameters), and NoisyLinear.reset_parameters (model.py:25-30):
    weight_mu, bias_mu ~ U(+-1/sqrt(in)); weight_sigma = std_init/sqrt(in); bias_sigma = std_init/sqrt(out).
    Drawn from torch's CPU generator."""
    flat = torch.zeros(n_params, dtype=torch.float32)
    shapes = dict((n, s) for n, _o, s in layout)
    fan_in = None
    for name, off, shape in layout:
        numel = int(np.prod(shape))
        if name.startswith("convs"):
            if name.endswith("weight"):
                fan_in = int(np.prod(shape[1:]))
            bound = 1.0 / math.sqrt(fan_in)                   # the bias of a layer follows its weight in the layout
            flat[off:off + numel] = torch.empty(numel).uniform_(-bound, bound)
        else:
            kind = name.split(".")[1]
            out_f, in_f = shapes[name.split(".")[0] + ".weight_mu"]
            if kind in ("weight_mu", "bias_mu"):
                bound = 1.0 / math.sqrt(in_f)                 # model.py:26,28
                flat[off:off + numel] = torch.empty(numel).uniform_(-bound, bound)
            elif kind == "weight_sigma":
                flat[off:off + numel] = noisy_std / math.sqrt(in_f)      # model.py:27
            else:
                flat[off:off + numel] = noisy_std / math.sqrt(out_f)     # model.py:29
    return flat


_ACT_PENDING = -7          # preset of the pinned action word while an act launch is in flight

# torch.cuda.current_stream(device).cuda_stream builds a Stream object per call (~4 us: a tenth of one act()); the raw handle
# of the same current stream is one C call (what torch's own compiled-kernel launchers use)
_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_handle(device):
    if _RAW_STREAM is not None:
        return _RAW_STREAM(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


class _FlatAdam(torch.optim.Adam):
    """torch.optim.Adam over the single flat parameter tensor (agent.py:46).  The state is ordinary Adam state
    (step / exp_avg / exp_avg_sq, so state_dict() and load_state_dict() work as usual), but step() runs the library's
    one-pass kernel (rb_learner_clip_adam), optionally with clip_grad_norm_ folded in (max_norm)."""

    def __init__(self, agent, **kw):
        super().__init__([agent._params], **kw)
        self._agent = weakref.ref(agent)
        p = agent._params
        self.state[p] = dict(step=torch.tensor(0.0, dtype=torch.float32),
                             exp_avg=torch.zeros_like(p, memory_format=torch.preserve_format),
                             exp_avg_sq=torch.zeros_like(p, memory_format=torch.preserve_format))

    def state_dict(self):
        ag = self._agent()
        if ag is not None:
            ag.flush()               # a deferred optimiser pass writes the moments
            ag._sync_step()          # the device counts the steps (deferred pass): refresh the host mirror first
        return super().state_dict()

    def load_state_dict(self, state_dict):
        ag = self._agent()
        if ag is not None:
            ag.flush()
        super().load_state_dict(state_dict)
        if ag is not None and ag._step_dev is not None:      # the device-resident step number follows the loaded one
            ag._step_dev.fill_(int(float(self.state[ag._params]["step"])))

    @torch.no_grad()
    def step(self, closure=None, max_norm=float("inf")):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.step_direct(max_norm)
        return loss

    def step_direct(self, max_norm=float("inf"), stream=None, defer=False):
        """What step() does, callable without torch.optim's step wrapper (profiler record + hook dispatch: ~20 us per call,
        a tenth of a batch-32 learn step's host time).  Agent.learn uses this; step() stays for API compatibility."""
        ag = self._agent()
        if ag is None:
            raise RuntimeError("the Agent that owns this optimiser is gone")
        g = self.param_groups[0]
        if g["amsgrad"] or g["weight_decay"] != 0 or g["maximize"]:
            raise NotImplementedError("rb_learner_clip_adam implements plain Adam (the reference's configuration)")
        st = self.state[g["params"][0]]
        st["step"] += 1
        b1, b2 = g["betas"]
        # device-resident step counter (deferred optimiser pass): step = 0 tells the kernel to read it and form the bias
        # corrections itself; the host copy above only mirrors it (Agent._sync_step refreshes it)
        step = 0 if ag._step_dev is not None else int(st["step"].item())
        fn = ag._lib.rb_learner_clip_adam_deferred if defer else ag._lib.rb_learner_clip_adam
        rc = fn(
            ag._h, float(max_norm), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), float(g["lr"]), float(b1),
            float(b2), float(g["eps"]), step,
            ag._norm_buf.data_ptr() if math.isfinite(max_norm) else None, ag._stream() if stream is None else stream)
        if rc != 0:
            L.check(ag._lib, rc)


class Agent:
    def __init__(self, args, env):
        self.device = torch.device(args.device)
        if self.device.type != "cuda":
            raise RuntimeError("rainbow_amd.Agent runs on MI355X: args.device must be a cuda (ROCm) device, got %s"
                               % self.device)
        self._lib = L.load()
        self.action_space = env.action_space()                       # agent.py:14
        self.atoms = args.atoms
        self.Vmin, self.Vmax = args.V_min, args.V_max
        self.support = torch.linspace(args.V_min, args.V_max, self.atoms).to(device=self.device)   # agent.py:18
        self.delta_z = (args.V_max - args.V_min) / (self.atoms - 1)
        self.batch_size = args.batch_size
        self.n = args.multi_step
        self.discount = args.discount
        self.norm_clip = args.norm_clip
        self.training = True

        arch = getattr(args, "architecture", "canonical")
        self._cfg = L.LearnerConfig(batch=int(args.batch_size), atoms=int(args.atoms), actions=int(self.action_space),
                                    history=int(args.history_length), hidden=int(args.hidden_size),
                                    architecture=0 if arch == "canonical" else 1, multi_step=int(args.multi_step),
                                    v_min=float(args.V_min), v_max=float(args.V_max), discount=float(args.discount))
        self._defer_update = False
        self._update_pending = False
        self._hosted_job = L.NoiseJob()
        n_params, n_noise = C.c_int64(0), C.c_int64(0)
        L.check(self._lib, self._lib.rb_learner_sizes(C.byref(self._cfg), C.byref(n_params), C.byref(n_noise)))
        self._layout = _query_layout(self._lib, self._cfg, self._lib.rb_learner_param_layout)
        self._noise_layout = _query_layout(self._lib, self._cfg, self._lib.rb_learner_noise_layout)
        d = self.device
        self._params = torch.zeros(n_params.value, dtype=torch.float32, device=d)          # online, flat
        self.target_params = torch.zeros(n_params.value, dtype=torch.float32, device=d)
        self._grads = torch.zeros(n_params.value, dtype=torch.float32, device=d)
        self.noise = torch.zeros(n_noise.value, dtype=torch.float32, device=d)
        self.target_noise = torch.zeros(n_noise.value, dtype=torch.float32, device=d)
        self._h = C.c_void_p()
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        with torch.cuda.device(d):
            L.check(self._lib, self._lib.rb_learner_create(
                C.byref(self._h), C.byref(self._cfg), self._params.data_ptr(), self.target_params.data_ptr(),
                self._grads.data_ptr(), self.noise.data_ptr(), self.target_noise.data_ptr(), seed))

        self._init_parameters(float(getattr(args, "noisy_std", 0.1)))
        self._noise_pending = False
        self.reset_noise()                                           # NoisyLinear.__init__ (model.py:23)
        if getattr(args, "model", None):                             # agent.py:26-36
            if os.path.isfile(args.model):
                # load_state_dict restores the epsilon buffers of the checkpoint too (they are registered buffers,
                # model.py:19,22): the noise drawn above is overwritten, exactly as in the reference
                self.load_state_dict(torch.load(args.model, map_location="cpu"))
                print("Loading pretrained model: " + args.model)
            else:
                raise FileNotFoundError(args.model)
        self.update_target_net()                                     # agent.py:41

        self._params.requires_grad_(True)
        self._params.grad = self._grads
        # (hipGraph replay of the step was built in rounds 1-3 and measured 4.5 % SLOWER than eager launches on this platform
        # at this launch count, DESIGN.md §6: removed)
        self._step_dev = None
        kw = dict(lr=args.learning_rate, eps=args.adam_eps)
        if os.environ.get("RAINBOW_AMD_FUSED_ADAM", "1") != "1":     # k_clip_scale + PyTorch's own (fused) Adam
            try:
                self.optimiser = torch.optim.Adam([self._params], fused=True, **kw)
            except (TypeError, RuntimeError):
                self.optimiser = torch.optim.Adam([self._params], **kw)
        else:
            # RAINBOW_AMD_DEFER_UPDATE (default on): learn() leaves its clip + Adam pass pending
            # and the NEXT learn()'s sampler launch hosts it (include/rainbow_hip.h RB_LEARNER_DEFER_UPDATE) — anything
            # else that touches the parameters runs it first (the library's entry points do so themselves; the public
            # tensors `params`, `grads`, `_norm` and the optimiser state go through flush()).  Needs the device-resident
            # step number (rb_learner_set_step_counter: the learn call increments it, the pass reads it).
            self._defer_update = os.environ.get("RAINBOW_AMD_DEFER_UPDATE", "1") == "1"
            if self._defer_update:
                self._step_dev = torch.zeros(1, dtype=torch.int64, device=d)
                L.check(self._lib, self._lib.rb_learner_set_step_counter(self._h, self._step_dev.data_ptr()))
            self.optimiser = _FlatAdam(self, **kw)                   # agent.py:46
        self._noise_jobs = {}
        self._zero_copy_ok = None
        # the whole step as ONE C call (rb_learner_train_step) when nothing needs the interpreter in between
        self._one_call = os.environ.get("RAINBOW_AMD_ONE_CALL", "1") == "1"
        self._ts = self._ts_mem = self._ts_out = None
        self._sink_watched = set()
        # priority write-back as one extra workgroup of the learner's backward launch (see rb_learner_set_priority_sink)
        self._fuse_update = True
        self._loss = torch.zeros(self.batch_size, dtype=torch.float32, device=d)
        self._norm_buf = torch.zeros(1, dtype=torch.float32, device=d)
        self._act_pin = torch.zeros(2 * self.batch_size, dtype=torch.int32).pin_memory()     # written by the device
        self._q_pin = torch.zeros(2 * self.batch_size, dtype=torch.float32).pin_memory()
        self._act_np, self._q_np = self._act_pin.numpy(), self._q_pin.numpy()
        # the single-state path (act / evaluate_q): (action, q) as ONE 8-byte pinned pair — the head stores both with a single
        # system-scope store (csrc/act_path.h rb_head_act_body) instead of q, a fence, then the action
        self._aq_pin = torch.zeros(2, dtype=torch.int32).pin_memory()
        self._aq_act = self._aq_pin.numpy()
        self._aq_q = self._aq_act.view(np.float32)
        self._aq_ptrs = (self._aq_pin.data_ptr(), self._aq_pin.data_ptr() + 4)
        self._pin_ptrs = None
        self._world = rdist.world_size()
        self._dist = rdist.active()
        self._exchange = None
        if self._dist:        # identical replicas: rank 0's initial parameters everywhere
            rdist.broadcast_parameters(self._params.detach(), 0)
            self.update_target_net()
            if rdist.mode() == "factored" and isinstance(self.optimiser, _FlatAdam):
                self._exchange = rdist.FactoredExchange(self._lib, self._h, self._grads)
        # RAINBOW_AMD_FUSED_DW=1 (single device + the library's own Adam): the hidden layer's weight gradient (93 % of the
        # gradient bytes) is never written to HBM — the backward computes it for the norm only, the clip + Adam pass
        # recomputes each tile while it streams that tile's parameters (include/rainbow_hip.h RB_LEARNER_FUSE_FC_H_DW);
        # self._grads then holds every OTHER gradient after learn().  51 MB less HBM traffic per step at the canonical
        # shape, measured 216.7 -> 215.3 us per step on MI355X: an opt-in, the default materialises every gradient as the
        # reference does.
        self._fused_dw = (isinstance(self.optimiser, _FlatAdam) and not self._dist
                          and os.environ.get("RAINBOW_AMD_FUSED_DW", "0") == "1")
        if self._fused_dw:
            self._defer_update = False
        # RAINBOW_AMD_IMPLICIT_SIGMA (default on with the deferred pass, single device): the hidden layer's sigma gradient is
        # g_mu * (eps_out x eps_in) element for element, so the backward does not store it and the hosted optimiser pass forms
        # it itself while it updates the (mu, sigma) pairs (include/rainbow_hip.h RB_LEARNER_IMPLICIT_SIGMA): 25.7 MB less HBM
        # traffic per step, bit-identical parameters; `grads` (which flushes) materialises it.
        self._implicit_sigma = (getattr(self, "_defer_update", False) and not self._dist
                                and os.environ.get("RAINBOW_AMD_IMPLICIT_SIGMA", "1") == "1")
        L.check(self._lib, self._lib.rb_learner_set_flags(
            self._h, (L.LEARNER_FUSE_FC_H_DW if self._fused_dw else 0) | (L.LEARNER_DEFER_UPDATE if self._defer_update else 0)
            | (L.LEARNER_IMPLICIT_SIGMA if self._implicit_sigma else 0)))

    # The flat tensors the library borrows.  With a deferred optimiser pass pending they are one update behind: reading them
    # through these names runs the pass first.
    @property
    def params(self):
        self.flush()
        return self._params

    @property
    def grads(self):
        self.flush()
        return self._grads

    @property
    def _norm(self):
        self.flush()
        return self._norm_buf

    def flush(self):
        """Run the optimiser pass the last learn() left pending (RAINBOW_AMD_DEFER_UPDATE), if any."""
        if self._update_pending:
            self._update_pending = False
            L.check(self._lib, self._lib.rb_learner_flush(self._h, self._stream()))

    # ------------------------------------------------------------------ plumbing
    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.rb_learner_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _stream(self):
        return current_stream_handle(self.device)

    def _view(self, flat, name):
        for n, off, shape in self._layout:
            if n == name:
                return flat.detach()[off:off + int(np.prod(shape))].view(shape)
        raise KeyError(name)

    def _noise_view(self, flat, name):
        for n, off, shape in self._noise_layout:
            if n == name:
                return flat[off:off + shape[0]]
        raise KeyError(name)

    def _init_parameters(self, noisy_std):
        with torch.no_grad():
            self._params.copy_(init_parameters_flat(self._layout, self._params.numel(), noisy_std))

    # ------------------------------------------------------------------ reference API
    def reset_noise(self, raw_normals=None):
        """agent.py:49-50 (online net only).  raw_normals: optional float32 device tensor of N(0,1) draws in the
        reference's order (parity hook)."""
        if raw_normals is None:
            # lazily: main.py calls reset_noise() right before learn() every 4th step (main.py:151,164); the draw is
            # deferred so learn() can resample online + target noise in ONE launch.  act()/evaluate_q() flush it first.
            self._noise_pending = True
            return
        self._noise_pending = False
        self._raw_on = raw_normals.to(device=self.device, dtype=torch.float32).contiguous()
        L.check(self._lib, self._lib.rb_learner_reset_noise(self._h, 0, self._raw_on.data_ptr(), self._stream()))

    def _flush_noise(self):
        if self._noise_pending:
            self._noise_pending = False
            L.check(self._lib, self._lib.rb_learner_reset_noise(self._h, 0, None, self._stream()))

    def _reset_target_noise(self, raw_normals=None):
        ptr = None
        if raw_normals is not None:
            self._raw_tg = raw_normals.to(device=self.device, dtype=torch.float32).contiguous()
            ptr = self._raw_tg.data_ptr()
        L.check(self._lib, self._lib.rb_learner_reset_noise(self._h, 1, ptr, self._stream()))   # agent.py:74

    def _forward_single(self, state):
        """One state through the act path; the action / value land in PINNED host memory the kernel writes directly
        (no device-to-host copy): they are final after the stream synchronize below."""
        if self._noise_pending:
            self._flush_noise()
        st = state
        if st.dtype != torch.float32 or st.device != self.device or not st.is_contiguous():     # (env.py hands over exactly this)
            st = state.to(device=self.device, dtype=torch.float32).contiguous()
        pins = self._aq_ptrs
        # launch + wait in ONE C call: completion = the pinned action word changes (the head writes action and q as one 8-byte word);
        # the library polls it in a compiled loop (a Python loop over a numpy scalar sees the store a microsecond or two late),
        # falls back to a stream synchronise, and retries once when the one-launch path reports an expired in-launch wait
        rc = self._lib.rb_learner_act_wait(self._h, st.data_ptr(), 1 if self.training else 0, pins[0], pins[1], None, None,
                                           current_stream_handle(self.device))
        if rc != 0:
            L.check(self._lib, rc)

    def act(self, state):
        """agent.py:53-55: greedy action on the expected value of the (noisy) online distribution."""
        self._forward_single(state)
        return int(self._aq_act[0])

    def act_batch(self, states):
        """Vectorised actors: `states` f32 [n, h, 84, 84] on the device (processed 2*batch_size at a time).
        Returns the n greedy actions (numpy int64), i.e. [self.act(s) for s in states] in one forward."""
        self._flush_noise()
        st = states.to(device=self.device, dtype=torch.float32).contiguous()
        n = int(st.shape[0])
        cap = int(self._act_np.shape[0])          # 2 * batch_size images fit the learner's activation buffers
        out = np.empty(n, dtype=np.int64)
        for lo in range(0, n, cap):
            m = min(cap, n - lo)
            L.check(self._lib, self._lib.rb_learner_act_batch(self._h, st[lo:lo + m].data_ptr(), m,
                                                              1 if self.training else 0, self._act_pin.data_ptr(),
                                                              self._q_pin.data_ptr(), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
            out[lo:lo + m] = self._act_np[:m]
        return out

    def act_e_greedy(self, state, epsilon=0.001):
        """agent.py:58-59."""
        return np.random.randint(0, self.action_space) if np.random.random() < epsilon else self.act(state)

    def evaluate_q(self, state):
        """agent.py:110-112."""
        self._forward_single(state)
        return float(self._aq_q[1])

    def evaluate_q_batch(self, states, chunk=512):
        """Batched agent.py:110-112: `states` f32 [n, h, 84, 84] on the device -> numpy float32 [n],
        [self.evaluate_q(s) for s in states] as ONE forward per `chunk` states (6 launches for a 500-state validation
        memory, test.py:38-39, instead of 500 single-state launch chains with a stream sync each)."""
        self._flush_noise()
        st = states.to(device=self.device, dtype=torch.float32).contiguous()
        n = int(st.shape[0])
        out = np.empty(n, dtype=np.float32)
        chunk = max(1, min(int(chunk), 4096))
        if self._q_pin.numel() < min(n, chunk):
            self._act_pin = torch.zeros(min(n, chunk), dtype=torch.int32).pin_memory()
            self._q_pin = torch.zeros(min(n, chunk), dtype=torch.float32).pin_memory()
            self._act_np, self._q_np = self._act_pin.numpy(), self._q_pin.numpy()
        for lo in range(0, n, chunk):
            m = min(chunk, n - lo)
            L.check(self._lib, self._lib.rb_learner_act_batch(self._h, st[lo:lo + m].data_ptr(), m,
                                                              1 if self.training else 0, self._act_pin.data_ptr(),
                                                              self._q_pin.data_ptr(), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
            out[lo:lo + m] = self._q_np[:m]
        return out

    def evaluate_q_memory(self, val_mem, chunk=512):
        """test.py:38-39 in one call: Q of every state the reference's `for state in val_mem` visits — ALL `capacity` slots
        (memory.py:166-168 walks data indices 0 .. capacity-1 whether or not they were ever written; unwritten slots are
        blank states) — built on the device by rb_replay_states_at."""
        n = val_mem.capacity
        qs = np.empty(n, dtype=np.float32)
        for lo in range(0, n, chunk):
            hi = min(n, lo + chunk)
            qs[lo:hi] = self.evaluate_q_batch(val_mem.states_at(torch.arange(lo, hi, device=self.device)), chunk)
        return qs

    def learn(self, mem, _target_raw_normals=None, _unit_uniforms=None):
        """agent.py:61-100.  With a rainbow_amd ReplayMemory the whole step (sample .. priority update) is
        device-resident and launched eagerly.  The injected-randomness arguments are parity-test hooks."""
        if isinstance(mem, ReplayMemory):
            self._raise_on_failed_samples(mem)
        self._learn_eager(mem, _target_raw_normals, _unit_uniforms)

    def _raise_on_failed_samples(self, mem):
        """The device sampler is bounded where the reference's rejection loop (memory.py:128-132) would spin forever.  A
        launch that gave up left NO trace on the device (zero importance weights; the priority write-back, the optimiser
        update and the device-resident step number are skipped when the header says so): here the host side is put back
        in step — the optimiser's step count is rolled back by the number of skipped updates, the counter is cleared —
        and the failure is raised.  The caller may append more transitions and call learn() again."""
        if self._step_dev is None:
            # by-value step numbers (RAINBOW_AMD_DEFER_UPDATE=0): the host's step count must be exact BEFORE the next
            # optimiser pass is issued, so the sampler launches in flight are waited for here (the default mode counts the
            # steps on the device and needs no such sync: a failure is then reported one learn() late, nothing else)
            torch.cuda.current_stream(self.device).synchronize()
        failed = mem.failed_samples()
        if not failed:
            return
        if isinstance(self.optimiser, _FlatAdam) and getattr(self, "_sink_mem", None) is mem:
            st = self.optimiser.state[self._params]
            st["step"] -= float(min(failed, int(st["step"].item())))
        mem.reset_failed_samples()
        raise RuntimeError("ReplayMemory: %d sampler launch(es) found no valid batch in %d attempts (replay too small for "
                           "batch %d?); those steps were skipped on the device (no priority write-back, no optimiser update)"
                           % (failed, mem.MAX_ATTEMPTS, self.batch_size))

    def _sync_step(self):
        """Host mirror of the device-resident optimiser step number (state_dict / checkpoint read it)."""
        if self._step_dev is not None and isinstance(self.optimiser, _FlatAdam):
            self.optimiser.state[self._params]["step"].fill_(float(self._step_dev.item()))

    def _learn_one_call(self, mem, stream):
        """The whole step through rb_learner_train_step (one C call): same entry points, same arguments, same order as
        _learn_eager's three calls — without the interpreter between the launches (include/rainbow_hip.h says why)."""
        B = self.batch_size
        ts = self._ts
        if mem._pending is not None:       # a caller's update_priorities() since the last draw: applied before this one
            mem.flush()
        if ts is None or self._ts_mem is not mem or mem._out.get(B) is not self._ts_out:
            o = mem._buffers(B)
            frames, windows, wlen = mem.frame_source()
            if getattr(self, "_sink_mem", None) is not mem or self._sink_idx is not o["tree_idxs"]:
                L.check(self._lib, self._lib.rb_learner_set_priority_sink(self._h, mem._h, o["tree_idxs"].data_ptr()))
                self._sink_mem, self._sink_idx = mem, o["tree_idxs"]
                self._watch_sink(mem)
            ts = L.TrainStep(replay=mem._h, batch=B, max_attempts=mem.MAX_ATTEMPTS, window_len=wlen,
                             tree_idx_dev=o["tree_idxs"].data_ptr(), actions_dev=o["actions"].data_ptr(),
                             returns_dev=o["returns"].data_ptr(), nonterminals_dev=o["nonterminals"].data_ptr(),
                             weights_dev=o["weights"].data_ptr(), frames_dev=frames, windows_dev=windows,
                             loss_dev=self._loss.data_ptr(), norm_dev=self._norm_buf.data_ptr())
            self._ts, self._ts_mem, self._ts_out = ts, mem, o
        if float(mem.priority_weight) != mem._neg_beta_val:
            mem._sync_beta()
        which = 2 if self._noise_pending else 1
        self._noise_pending = False
        job = self._noise_jobs.get(which)
        if job is None:
            job = L.NoiseJob()
            L.check(self._lib, self._lib.rb_learner_noise_job(self._h, which, C.byref(job)))
            self._noise_jobs[which] = job
        g = self.optimiser.param_groups[0]
        st = self.optimiser.state[self._params]        # looked up every step: load_state_dict replaces these tensors
        st["step"] += 1
        ts.exp_avg_dev, ts.exp_avg_sq_dev = st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
        ts.priority_weight = float(mem.priority_weight)
        ts.noise_job = C.addressof(job)
        ts.lr, ts.eps, ts.max_norm = float(g["lr"]), float(g["eps"]), float(self.norm_clip)
        ts.beta1, ts.beta2 = g["betas"]
        ts.step = 0 if self._step_dev is not None else int(st["step"].item())
        if self._exchange is not None:     # replicas: the library's own RCCL all-gather sits between backward and clip
            rc = self._lib.rb_learner_train_step_dist(self._h, C.byref(ts), self._exchange.comm, stream)
        else:
            rc = self._lib.rb_learner_train_step(self._h, C.byref(ts), stream)
        if rc != 0:
            L.check(self._lib, rc)
        self._update_pending = self._defer_update
        if not self._lib.rb_learner_priority_written(self._h):       # (cannot happen with a sink set; keeps agent.py:100)
            # immediately: a lazy write-back would read tree_idxs / _loss after the next step has overwritten them
            mem.update_priorities(self._ts_out["tree_idxs"], self._loss, _immediate=True)

    def _learn_eager(self, mem, _target_raw_normals=None, _unit_uniforms=None):
        B = self.batch_size
        device_mem = isinstance(mem, ReplayMemory)
        stream = self._stream()                       # ONE lookup per step (torch.cuda.current_stream costs ~4 us a call)
        if self._zero_copy_ok is None:                # a property of the learner's configuration: asked once
            self._zero_copy_ok = bool(self._lib.rb_learner_zero_copy_ok(self._h))
        zero_copy = device_mem and self._zero_copy_ok and mem.history == self._cfg.history and mem.n == self.n
        lib_comm = self._exchange is not None and getattr(self._exchange, "comm", None) is not None
        if (zero_copy and self._one_call and _target_raw_normals is None and _unit_uniforms is None and self._fuse_update
                and ((self._exchange is None and not self._dist) or lib_comm)
                and isinstance(self.optimiser, _FlatAdam) and math.isfinite(float(self.norm_clip))):
            g = self.optimiser.param_groups[0]
            if not (g["amsgrad"] or g["weight_decay"] != 0 or g["maximize"]):
                return self._learn_one_call(mem, stream)
        noise_job = None
        overwrite_target = False
        if device_mem and _target_raw_normals is not None and self._update_pending and not self._noise_pending:
            # parity runs (injected normals) keep the product's launch shape: the sampler launch still hosts the pending
            # optimiser pass, carried by a target-noise job whose device-RNG draw the injected normals overwrite below
            noise_job = self._noise_jobs.get(1)
            if noise_job is None:
                noise_job = L.NoiseJob()
                L.check(self._lib, self._lib.rb_learner_noise_job(self._h, 1, C.byref(noise_job)))
                self._noise_jobs[1] = noise_job
            overwrite_target = True
        elif device_mem and _target_raw_normals is None:
            # the target-noise draw of this step (agent.py:74) — plus the deferred online draw (main.py:151) when one is
            # pending — rides along in the sampler's launch: it does not depend on the batch, only has to precede the
            # forwards, and the draws are the same Philox epochs as separate launches (online first, then target)
            which = 2 if self._noise_pending else 1
            self._noise_pending = False
            noise_job = self._noise_jobs.get(which)
            if noise_job is None:
                noise_job = L.NoiseJob()
                L.check(self._lib, self._lib.rb_learner_noise_job(self._h, which, C.byref(noise_job)))
                self._noise_jobs[which] = noise_job
        if device_mem:
            hosted = 0
            if noise_job is not None and self._update_pending:
                # the previous call's clip + Adam pass rides in this sampler launch (RB_LEARNER_DEFER_UPDATE)
                hosted = self._lib.rb_learner_attach_pending(self._h, C.byref(noise_job), B, C.byref(self._hosted_job))
                if hosted < 0:
                    L.check(self._lib, hosted)
            o = mem.sample_device(B, _unit_uniforms, gather=not zero_copy,
                                  noise_job=self._hosted_job if hosted == 1 else noise_job, stream=stream)   # agent.py:63
            if hosted == 1:
                L.check(self._lib, self._lib.rb_learner_pending_launched(self._h))
                self._update_pending = False
            idxs, states, next_states = o["tree_idxs"], o["states"], o["next_states"]
            actions, returns, nonterminals, weights = o["actions"], o["returns"], o["nonterminals"], o["weights"]
        else:   # foreign replay with the reference's API: float32 /255 states come back; re-quantise (exact for k/255)
            idxs, s, actions, returns, ns, nonterminals, weights = mem.sample(B)
            d = self.device
            states = s.to(d).mul(255).round_().to(torch.uint8).contiguous()
            next_states = ns.to(d).mul(255).round_().to(torch.uint8).contiguous()
            actions = actions.to(device=d, dtype=torch.int64).contiguous()
            returns = returns.to(device=d, dtype=torch.float32).contiguous()
            nonterminals = nonterminals.to(device=d, dtype=torch.float32).reshape(B).contiguous()
            weights = weights.to(device=d, dtype=torch.float32).contiguous()
        if noise_job is None:
            if self._noise_pending and _target_raw_normals is None:
                self._noise_pending = False      # online (main.py:151) and target (agent.py:74) noise in one launch
                L.check(self._lib, self._lib.rb_learner_reset_noise(self._h, 2, None, stream))
            else:
                self._flush_noise()
                self._reset_target_noise(_target_raw_normals)                              # agent.py:74
        elif overwrite_target:
            self._reset_target_noise(_target_raw_normals)                                  # agent.py:74 (injected draw)
        if (device_mem and self._fuse_update
                and (getattr(self, "_sink_mem", None) is not mem or self._sink_idx is not idxs)):
            # the learner writes the new priorities into mem's sum-tree itself (one extra workgroup of its backward)
            L.check(self._lib, self._lib.rb_learner_set_priority_sink(self._h, mem._h, idxs.data_ptr()))
            self._sink_mem, self._sink_idx = mem, idxs
            self._watch_sink(mem)
        elif not device_mem and getattr(self, "_sink_mem", None) is not None:
            L.check(self._lib, self._lib.rb_learner_set_priority_sink(self._h, None, None))
            self._sink_mem, self._sink_idx = None, None
        if zero_copy:   # conv1 reads the frames straight out of the HBM ring: no stack gather at all
            frames, windows, wlen = mem.frame_source()
            rc = self._lib.rb_learner_learn_windows(
                self._h, frames, windows, wlen, actions.data_ptr(), returns.data_ptr(), nonterminals.data_ptr(),
                weights.data_ptr(), self._loss.data_ptr(), stream)
            if rc != 0:
                L.check(self._lib, rc)
        else:
            L.check(self._lib, self._lib.rb_learner_learn(
                self._h, states.data_ptr(), next_states.data_ptr(), actions.data_ptr(), returns.data_ptr(),
                nonterminals.data_ptr(), weights.data_ptr(), self._loss.data_ptr(), stream))   # agent.py:66-96
        fused_update = device_mem and bool(self._lib.rb_learner_priority_written(self._h))
        if self._exchange is not None:
            # replicas, factored exchange (the insert point between agent.py:96 and :97): ONE all-gather of every rank's
            # block — the factors of its FC weight gradients (dY and X rows, its noise vectors) and its conv gradients,
            # 1.0 MB — on this stream, then rb_learner_finish_grads forms the replica-mean gradient of the global batch
            # on every replica (rainbow_amd/dist.py)
            self._exchange.run()
        elif self._dist:      # replicas, plain exchange: one RCCL all-reduce of the flat gradient (4*P bytes)
            rdist.average_gradients(self._grads)
            L.check(self._lib, self._lib.rb_learner_grads_modified(self._h))
        if isinstance(self.optimiser, _FlatAdam):
            defer = self._defer_update and device_mem
            self.optimiser.step_direct(float(self.norm_clip), stream, defer=defer)         # agent.py:97-98, one pass
            if defer:
                self._update_pending = True      # (the library decides; flush() is a no-op when it ran at once)
        else:
            L.check(self._lib, self._lib.rb_learner_clip_grad(self._h, float(self.norm_clip), self._norm_buf.data_ptr(),
                                                              stream))                    # agent.py:97
            self.optimiser.step()                                                          # agent.py:98
        if device_mem:
            if not fused_update:
                mem.update_priorities(idxs, self._loss, _immediate=True)                   # agent.py:100, no D2H
        else:
            mem.update_priorities(idxs, self._loss.detach().cpu().numpy())                 # agent.py:100

    def _watch_sink(self, mem):
        """The library caches raw pointers into `mem` (priority sink): drop them when THAT memory goes away before the agent
        does.  One finalizer per memory, and it only acts if that memory is still the current sink (after a switch from
        memory A to memory B, collecting A must not clear B's sink)."""
        key = id(mem)
        if key in self._sink_watched:
            return
        self._sink_watched.add(key)
        me = weakref.ref(self)

        def gone(key=key):
            ag = me()
            if ag is not None:
                ag._sink_watched.discard(key)
                if ag._sink_key == key:
                    ag._clear_sink()
        weakref.finalize(mem, gone)

    @property
    def _sink_key(self):
        m = getattr(self, "_sink_mem", None)
        return id(m) if m is not None else None

    def _clear_sink(self):
        if getattr(self, "_h", None):
            self._lib.rb_learner_set_priority_sink(self._h, None, None)
        self._sink_mem, self._sink_idx = None, None
        self._ts = self._ts_mem = self._ts_out = None      # the one-call path re-arms the sink on its next step

    def update_target_net(self):
        """agent.py:102-103."""
        self._flush_noise()
        L.check(self._lib, self._lib.rb_learner_sync_target(self._h, self._stream()))

    def train(self):
        self.training = True        # agent.py:114-115

    def eval(self):
        self.training = False       # agent.py:117-118 -> mu-only forward (model.py:46)

    # ------------------------------------------------------------------ checkpoints (agent.py:26-36,106-107)
    def state_dict(self):
        """Reference-compatible keys: convs.{0,2,4}.{weight,bias}, fc_*.{weight,bias}_{mu,sigma,epsilon}."""
        self._flush_noise()
        self.flush()
        sd = {}
        for name, _off, _shape in self._layout:
            sd[name] = self._view(self._params, name).clone()
        for layer in _LAYERS:
            e_in = self._noise_view(self.noise, layer + ".eps_in")
            e_out = self._noise_view(self.noise, layer + ".eps_out")
            sd[layer + ".weight_epsilon"] = torch.outer(e_out, e_in)    # model.py:39
            sd[layer + ".bias_epsilon"] = e_out.clone()                 # model.py:40
        # nn.Module.state_dict order: a module's parameters, then its buffers (model.py:15-22)
        order = []
        for name, _o, _s in self._layout:
            order.append(name)
            if name.endswith("bias_sigma"):
                layer = name.split(".")[0]
                order += [layer + ".weight_epsilon", layer + ".bias_epsilon"]
        return {k: sd[k] for k in order}

    def load_state_dict(self, state_dict, strict=True):
        """nn.Module.load_state_dict of the reference's DQN (agent.py:33): STRICT by default — missing or unexpected keys
        raise, as the reference's call does.  The [out, in] weight_epsilon buffers are rank-1 (model.py:39); the
        factorised vectors are recovered EXACTLY (see _recover_eps_in), so save -> load -> save is bit-identical."""
        self.flush()
        sd = dict(state_dict)
        if "conv1.weight" in sd:                                                           # agent.py:29-32
            for old, new in (("conv1.weight", "convs.0.weight"), ("conv1.bias", "convs.0.bias"),
                             ("conv2.weight", "convs.2.weight"), ("conv2.bias", "convs.2.bias"),
                             ("conv3.weight", "convs.4.weight"), ("conv3.bias", "convs.4.bias")):
                sd[new] = sd.pop(old)
        expected = [n for n, _o, _s in self._layout]
        for layer in _LAYERS:
            expected += [layer + ".weight_epsilon", layer + ".bias_epsilon"]
        missing = [k for k in expected if k not in sd]
        unexpected = [k for k in sd if k not in expected]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for DQN: missing key(s) %s; unexpected key(s) %s"
                               % (missing, unexpected))
        self._flush_noise()
        with torch.no_grad():
            for name, _off, shape in self._layout:
                if name not in sd:
                    continue
                t = sd[name]
                if tuple(t.shape) != tuple(shape):
                    raise RuntimeError("size mismatch for %s: %s vs %s" % (name, tuple(t.shape), shape))
                self._view(self._params, name).copy_(t.to(self.device, torch.float32))
            for layer in _LAYERS:
                if layer + ".bias_epsilon" not in sd or layer + ".weight_epsilon" not in sd:
                    continue
                e_out = sd[layer + ".bias_epsilon"].to(self.device, torch.float32)
                e_w = sd[layer + ".weight_epsilon"].to(self.device, torch.float32)
                self._noise_view(self.noise, layer + ".eps_in").copy_(self._recover_eps_in(e_w, e_out))
                self._noise_view(self.noise, layer + ".eps_out").copy_(e_out)

    @staticmethod
    def _recover_eps_in(e_w, e_out):
        """weight_epsilon = fl32(eps_out[i] * eps_in[k]) (model.py:39, torch.ger).  A single division e_w[j] / e_out[j] can
        land one ulp beside the true eps_in[k]; the true value is the neighbour that REPRODUCES the stored products.
        Candidates {q - ulp, q, q + ulp} of the quotient on the row of largest |eps_out| are scored against the 64 rows
        of largest |eps_out|; the best one per column wins (the true vector scores 0 mismatches by construction)."""
        order = torch.argsort(e_out.abs(), descending=True)
        j = int(order[0].item())
        if float(e_out[j]) == 0.0:
            return torch.zeros_like(e_w[0])
        q = e_w[j] / e_out[j]
        inf = torch.full_like(q, float("inf"))
        cands = torch.stack([torch.nextafter(q, -inf), q, torch.nextafter(q, inf)])       # [3, K]
        rows = order[:64]
        prod = e_out[rows][None, :, None] * cands[:, None, :]                              # [3, R, K] float32 products
        miss = (prod != e_w[rows][None, :, :]).sum(dim=1)                                  # [3, K]
        miss[1] = miss[1] * 2 - 1                                                          # ties prefer the quotient itself
        miss[0] *= 2
        miss[2] *= 2
        best = torch.argmin(miss, dim=0)
        return cands.gather(0, best[None, :])[0]

    # ------------------------------------------------------------------ exact resume
    def checkpoint(self, path=None):
        """Everything the learner needs to continue bit-for-bit: online / target parameters, both noise buffers, Adam
        moments + step, the Philox (seed, epoch) of the noise generator and the pending-resample flag.  The reference's
        save() (weights only, agent.py:106-107) stays what main.py:182 calls; this is the superset a resumable run needs
        (main.py has no such thing: it restarts the optimiser).  Returns the dict; writes it with torch.save if `path`."""
        if not isinstance(self.optimiser, _FlatAdam):
            raise NotImplementedError("checkpoint() covers the library's own Adam (RAINBOW_AMD_FUSED_ADAM=1)")
        self.flush()
        self._sync_step()
        seed, epoch = C.c_uint64(0), C.c_uint64(0)
        L.check(self._lib, self._lib.rb_learner_get_rng(self._h, C.byref(seed), C.byref(epoch), self._stream()))
        st = self.optimiser.state[self._params]
        ck = dict(version=1, config=bytes(self._cfg), params=self._params.detach().cpu(), target_params=self.target_params.cpu(),
                  noise=self.noise.cpu(), target_noise=self.target_noise.cpu(), exp_avg=st["exp_avg"].cpu(),
                  exp_avg_sq=st["exp_avg_sq"].cpu(), adam_step=float(st["step"]), rng_seed=int(seed.value),
                  rng_epoch=int(epoch.value), noise_pending=bool(self._noise_pending), training=bool(self.training))
        if path is not None:
            torch.save(ck, path)
        return ck

    def restore(self, ck):
        """Inverse of checkpoint(): `ck` is the dict or a path."""
        if not isinstance(self.optimiser, _FlatAdam):     # before anything is overwritten
            raise NotImplementedError("restore() covers the library's own Adam (RAINBOW_AMD_FUSED_ADAM=1)")
        self.flush()
        if not isinstance(ck, dict):
            ck = torch.load(ck, map_location="cpu")
        if ck.get("version") != 1 or ck["config"] != bytes(self._cfg):
            raise RuntimeError("checkpoint does not match this Agent's configuration")
        with torch.no_grad():
            self._params.copy_(ck["params"])
            self.target_params.copy_(ck["target_params"])
            self.noise.copy_(ck["noise"])
            self.target_noise.copy_(ck["target_noise"])
            st = self.optimiser.state[self._params]
            st["exp_avg"].copy_(ck["exp_avg"])
            st["exp_avg_sq"].copy_(ck["exp_avg_sq"])
            st["step"].fill_(ck["adam_step"])
            if self._step_dev is not None:
                self._step_dev.fill_(int(ck["adam_step"]))
        L.check(self._lib, self._lib.rb_learner_set_rng(self._h, int(ck["rng_seed"]), int(ck["rng_epoch"]), self._stream()))
        self._noise_pending = bool(ck["noise_pending"])
        self.training = bool(ck["training"])
        self._noise_jobs = {}

    def save(self, path, name="model.pth"):
        torch.save(self.state_dict(), os.path.join(path, name))
