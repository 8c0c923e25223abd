"""C-ABI backends for tests/scenarios.py.  The same adapter drives
  - the host-interpreted test build (tests/hipemu/librainbow_emu.so, numpy memory, CPU), and
  - the shipped HIP library (rainbow_amd/librainbow_hip.so, torch cuda memory, -m gpu).
"""
import ctypes as C

import numpy as np

from rainbow_amd import _lib as L


class NumpyMem:
    """'Device' memory for the host interpreter: plain numpy arrays."""
    stream = None

    def empty(self, shape, dtype):
        return np.zeros(shape, dtype=dtype)

    def upload(self, arr):
        return np.ascontiguousarray(arr).copy()

    def download(self, buf):
        return np.array(buf, copy=True)

    def ptr(self, buf):
        return buf.ctypes.data if buf is not None else None

    def view(self, ptr, shape, dtype):
        n = int(np.prod(shape))
        ct = C.cast(ptr, C.POINTER(C.c_uint8))
        return np.ctypeslib.as_array(ct, shape=(n * np.dtype(dtype).itemsize,)).view(dtype).reshape(shape).copy()

    def sync(self):
        pass


class TorchMem:
    """Device memory owned by torch (HBM) for the real library."""

    def __init__(self):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda:0")
        self.stream = torch.cuda.current_stream().cuda_stream

    _map = {np.dtype(np.float32): "float32", np.dtype(np.float64): "float64", np.dtype(np.int64): "int64",
            np.dtype(np.int32): "int32", np.dtype(np.uint8): "uint8"}

    def empty(self, shape, dtype):
        return self.torch.zeros(shape, dtype=getattr(self.torch, self._map[np.dtype(dtype)]), device=self.dev)

    def upload(self, arr):
        return self.torch.from_numpy(np.ascontiguousarray(arr)).to(self.dev)

    def download(self, buf):
        return buf.detach().cpu().numpy()

    def ptr(self, buf):
        return buf.data_ptr() if buf is not None else None

    def view(self, ptr, shape, dtype):
        # copy raw library-owned device memory to the host through the library's own copy helper
        host = np.empty(shape, dtype=dtype)
        lib = L.load()
        L.check(lib, lib.rb_copy_to_host(host.ctypes.data, ptr, host.nbytes, self.stream))
        return host

    def sync(self):
        self.torch.cuda.synchronize()


class CAbiReplayAdapter:
    def __init__(self, lib, mem, capacity, history, n, discount, omega, seed=7):
        self.lib, self.mem = lib, mem
        self.capacity, self.history, self.n = capacity, history, n
        self.h = C.c_void_p()
        L.check(lib, lib.rb_replay_create(C.byref(self.h), capacity, history, n, discount, omega, seed))
        self.t = 0
        self.bufs = L.ReplayBuffers()
        L.check(lib, lib.rb_replay_buffers(self.h, C.byref(self.bufs)))

    def close(self):
        if self.h:
            self.lib.rb_replay_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- protocol ----------------------------------------------------------------
    def append(self, state, action, reward, terminal):
        m = self.mem
        st = m.upload(np.asarray(state, dtype=np.float32))
        L.check(self.lib, self.lib.rb_replay_append(self.h, m.ptr(st), self.t, int(action), float(reward),
                                                    0 if terminal else 1, m.stream))
        m.sync()
        self.t = 0 if terminal else self.t + 1

    def append_batch(self, frames_u8, timesteps, actions, rewards, nonterminals):
        m = self.mem
        bufs = [m.upload(np.asarray(frames_u8, dtype=np.uint8)), m.upload(np.asarray(timesteps, dtype=np.int32)),
                m.upload(np.asarray(actions, dtype=np.int32)), m.upload(np.asarray(rewards, dtype=np.float32)),
                m.upload(np.asarray(nonterminals, dtype=np.uint8))]
        L.check(self.lib, self.lib.rb_replay_append_batch(self.h, *[m.ptr(b) for b in bufs], len(timesteps), m.stream))
        m.sync()

    def sample(self, batch, unit_uniforms, beta):
        m = self.mem
        h = self.history
        uu = m.upload(np.asarray(unit_uniforms, dtype=np.float64)) if unit_uniforms is not None else None
        attempts = int(np.asarray(unit_uniforms).shape[0]) if unit_uniforms is not None else 16
        tree_idx = m.empty((batch,), np.int64)
        states = m.empty((batch, h, 84, 84), np.uint8)
        next_states = m.empty((batch, h, 84, 84), np.uint8)
        actions = m.empty((batch,), np.int64)
        returns = m.empty((batch,), np.float32)
        nonterm = m.empty((batch,), np.float32)
        weights = m.empty((batch,), np.float32)
        L.check(self.lib, self.lib.rb_replay_sample(self.h, batch, float(beta), m.ptr(uu), attempts, m.ptr(tree_idx),
                                                    m.ptr(states), m.ptr(next_states), m.ptr(actions), m.ptr(returns),
                                                    m.ptr(nonterm), m.ptr(weights), m.stream))
        m.sync()
        hdr = self.raw_header()
        assert hdr.last_status == 0, "device sampler gave up after %d attempts" % hdr.last_attempts
        return dict(tree_idxs=m.download(tree_idx), states=m.download(states), next_states=m.download(next_states),
                    actions=m.download(actions), returns=m.download(returns),
                    nonterminals=m.download(nonterm)[:, None], weights=m.download(weights),
                    attempts=hdr.last_attempts)

    def update_sample(self, upd_tree_idxs, upd_losses, batch, unit_uniforms, beta, check_status=True):
        """rb_replay_update_sample: update_priorities(upd_*) + sample(batch) as one call (one launch up to 256 / 256)."""
        m = self.mem
        h = self.history
        ti = m.upload(np.asarray(upd_tree_idxs, dtype=np.int64))
        lo = m.upload(np.asarray(upd_losses, dtype=np.float32))
        uu = m.upload(np.asarray(unit_uniforms, dtype=np.float64)) if unit_uniforms is not None else None
        attempts = int(np.asarray(unit_uniforms).shape[0]) if unit_uniforms is not None else 16
        tree_idx = m.empty((batch,), np.int64)
        states = m.empty((batch, h, 84, 84), np.uint8)
        next_states = m.empty((batch, h, 84, 84), np.uint8)
        actions = m.empty((batch,), np.int64)
        returns = m.empty((batch,), np.float32)
        nonterm = m.empty((batch,), np.float32)
        weights = m.empty((batch,), np.float32)
        L.check(self.lib, self.lib.rb_replay_update_sample(self.h, m.ptr(ti), m.ptr(lo), len(upd_tree_idxs), batch, float(beta), m.ptr(uu),
                                                           attempts, m.ptr(tree_idx), m.ptr(states), m.ptr(next_states), m.ptr(actions),
                                                           m.ptr(returns), m.ptr(nonterm), m.ptr(weights), m.stream))
        m.sync()
        hdr = self.raw_header()
        if check_status:
            assert hdr.last_status == 0, "device sampler gave up after %d attempts" % hdr.last_attempts
        return dict(tree_idxs=m.download(tree_idx), states=m.download(states), next_states=m.download(next_states),
                    actions=m.download(actions), returns=m.download(returns),
                    nonterminals=m.download(nonterm)[:, None], weights=m.download(weights),
                    attempts=hdr.last_attempts)

    def update_priorities(self, tree_idxs, losses):
        m = self.mem
        ti = m.upload(np.asarray(tree_idxs, dtype=np.int64))
        lo = m.upload(np.asarray(losses, dtype=np.float32))
        L.check(self.lib, self.lib.rb_replay_update_priorities(self.h, m.ptr(ti), m.ptr(lo), len(tree_idxs), m.stream))
        m.sync()

    def update_leaves(self, tree_idxs, values):
        m = self.mem
        ti = m.upload(np.asarray(tree_idxs, dtype=np.int64))
        va = m.upload(np.asarray(values, dtype=np.float32))
        L.check(self.lib, self.lib.rb_replay_update_leaves(self.h, m.ptr(ti), m.ptr(va), len(tree_idxs), m.stream))
        m.sync()

    def find(self, values):
        m = self.mem
        n = len(values)
        v = m.upload(np.asarray(values, dtype=np.float64))
        probs, di, ti = m.empty((n,), np.float32), m.empty((n,), np.int64), m.empty((n,), np.int64)
        L.check(self.lib, self.lib.rb_replay_find(self.h, m.ptr(v), n, m.ptr(probs), m.ptr(di), m.ptr(ti), m.stream))
        m.sync()
        return m.download(probs), m.download(di), m.download(ti)

    def tree(self):
        return self.mem.view(self.bufs.sum_tree_dev, (self.bufs.tree_len,), np.float32)

    def raw_header(self):
        hdr = L.ReplayHeader()
        L.check(self.lib, self.lib.rb_replay_header(self.h, C.byref(hdr), self.mem.stream))
        return hdr

    def header(self):
        hdr = self.raw_header()
        return int(hdr.index), bool(hdr.full), np.float32(hdr.max)

    def state_at(self, i):
        m = self.mem
        out = m.empty((self.history, 84, 84), np.float32)
        L.check(self.lib, self.lib.rb_replay_state_at(self.h, int(i), m.ptr(out), m.stream))
        m.sync()
        return m.download(out)


# =============================================================================== learn step
import scenarios  # noqa: E402


def learner_config(c):
    return L.LearnerConfig(batch=c["batch"], atoms=c["atoms"], actions=c["actions"], history=c["history"],
                           hidden=c["hidden"], architecture=0 if c["architecture"] == "canonical" else 1,
                           multi_step=c["multi_step"], v_min=c["v_min"], v_max=c["v_max"], discount=c["discount"])


def query_layout(lib, cfg, fn):
    n = C.c_int32(0)
    L.check(lib, fn(C.byref(cfg), None, C.byref(n)))
    descs = (L.TensorDesc * n.value)()
    L.check(lib, fn(C.byref(cfg), descs, C.byref(n)))
    out = {}
    for d in descs[:n.value]:
        shape = tuple(d.shape[i] for i in range(d.ndim))
        out[d.name.decode()] = (int(d.offset), shape)
    return out


class CAbiLearnAdapter:
    """Drives rb_learner_* exactly like rainbow_amd.agent.Agent does: learn -> fused clip+Adam (default), or
    learn -> clip -> torch Adam (fused_adam = False: the hipGraph path of the Agent and the pre-fusion behaviour)."""
    fused_adam = True
    step_from_device = False   # pass step = 0: the kernel reads the step number from rb_learner_set_step_counter's counter
    learner_flags = 0          # rainbow_amd._lib.LEARNER_* (set on the class or instance BEFORE load())

    def __init__(self, lib, mem, name):
        import torch
        self.torch = torch
        self.lib, self.mem = lib, mem
        self.c = scenarios.LEARN_CONFIGS[name]
        self.hy = scenarios.LEARN_HYPER
        self.cfg = learner_config(self.c)
        n_params, n_noise = C.c_int64(0), C.c_int64(0)
        L.check(lib, lib.rb_learner_sizes(C.byref(self.cfg), C.byref(n_params), C.byref(n_noise)))
        self.n_params, self.n_noise = n_params.value, n_noise.value
        self.layout = query_layout(lib, self.cfg, lib.rb_learner_param_layout)
        self.p_on = mem.empty((self.n_params,), np.float32)
        self.p_tg = mem.empty((self.n_params,), np.float32)
        self.grads = mem.empty((self.n_params,), np.float32)
        self.z_on = mem.empty((self.n_noise,), np.float32)
        self.z_tg = mem.empty((self.n_noise,), np.float32)
        self.h = C.c_void_p()
        L.check(lib, lib.rb_learner_create(C.byref(self.h), C.byref(self.cfg), mem.ptr(self.p_on), mem.ptr(self.p_tg),
                                           mem.ptr(self.grads), mem.ptr(self.z_on), mem.ptr(self.z_tg), 1234))
        self.opt = None

    def close(self):
        if self.h:
            self.lib.rb_learner_destroy(self.h)
            self.h = None

    def _flat(self, params):
        flat = np.zeros(self.n_params, dtype=np.float32)
        for name, (off, shape) in self.layout.items():
            a = np.asarray(params[name], dtype=np.float32)
            assert a.shape == shape, (name, a.shape, shape)
            flat[off:off + a.size] = a.ravel()
        return flat

    def _unflat(self, flat):
        return {name: flat[off:off + int(np.prod(shape))].reshape(shape).copy() for name, (off, shape) in self.layout.items()}

    def _as_torch(self, buf):
        return self.torch.from_numpy(buf) if isinstance(buf, np.ndarray) else buf

    def load(self, online, target):
        m = self.mem
        L.check(self.lib, self.lib.rb_learner_set_flags(self.h, int(self.learner_flags) if self.fused_adam else 0))
        if isinstance(self.p_on, np.ndarray):
            self.p_on[:] = self._flat(online)
            self.p_tg[:] = self._flat(target)
        else:
            self.p_on.copy_(self.torch.from_numpy(self._flat(online)))
            self.p_tg.copy_(self.torch.from_numpy(self._flat(target)))
        p = self._as_torch(self.p_on).requires_grad_()
        p.grad = self._as_torch(self.grads)
        self.param_t = p
        self.opt = self.torch.optim.Adam([p], lr=self.hy["lr"], eps=self.hy["adam_eps"])   # agent.py:46
        self.adam_m = m.upload(np.zeros(self.n_params, dtype=np.float32))
        self.adam_v = m.upload(np.zeros(self.n_params, dtype=np.float32))
        self.adam_t = 0
        m.sync()

    def reset_noise_online(self, raw):
        m = self.mem
        r = m.upload(np.asarray(raw, dtype=np.float32))
        L.check(self.lib, self.lib.rb_learner_reset_noise(self.h, 0, m.ptr(r), m.stream))
        m.sync()

    grad_hook = None   # e.g. rainbow_amd.dist.average_gradients between backward and clip
    exchange = None    # or a rainbow_amd.dist.FactoredExchange over this handle

    def learn_only(self, batch, target_raw):
        """rb_learner_reset_noise(target) + rb_learner_learn: everything of agent.py:61-96 (backward included)."""
        m = self.mem
        B = self.c["batch"]
        r = m.upload(np.asarray(target_raw, dtype=np.float32))
        L.check(self.lib, self.lib.rb_learner_reset_noise(self.h, 1, m.ptr(r), m.stream))       # agent.py:74
        bufs = dict(states=m.upload(batch["states"]), next_states=m.upload(batch["next_states"]),
                    actions=m.upload(batch["actions"].astype(np.int64)), returns=m.upload(batch["returns"]),
                    nonterminals=m.upload(batch["nonterminals"].astype(np.float32).reshape(B)),
                    weights=m.upload(batch["weights"]))
        self._loss = m.empty((B,), np.float32)
        self._bufs = bufs                                    # keep the inputs alive until the stream has consumed them
        L.check(self.lib, self.lib.rb_learner_learn(self.h, m.ptr(bufs["states"]), m.ptr(bufs["next_states"]),
                                                    m.ptr(bufs["actions"]), m.ptr(bufs["returns"]),
                                                    m.ptr(bufs["nonterminals"]), m.ptr(bufs["weights"]), m.ptr(self._loss),
                                                    m.stream))

    def finish_step(self):
        """clip_grad_norm_ + Adam (agent.py:97-98) on whatever the gradient buffer holds now."""
        m = self.mem
        norm = m.empty((1,), np.float32)
        if self.fused_adam:
            self.adam_t += 1
            L.check(self.lib, self.lib.rb_learner_clip_adam(self.h, self.hy["norm_clip"], m.ptr(self.adam_m),
                                                            m.ptr(self.adam_v), self.hy["lr"], 0.9, 0.999,
                                                            self.hy["adam_eps"], 0 if self.step_from_device else self.adam_t,
                                                            m.ptr(norm), m.stream))
            m.sync()
            grads = self._unflat(m.download(self.grads))
        else:
            L.check(self.lib, self.lib.rb_learner_clip_grad(self.h, self.hy["norm_clip"], m.ptr(norm), m.stream))
            m.sync()
            grads = self._unflat(m.download(self.grads))
            self.opt.step()                                                                        # agent.py:98
        m.sync()
        return dict(loss=m.download(self._loss), grad_norm=float(m.download(norm)[0]), grads=grads)

    def learn_step(self, batch, target_raw):
        m = self.mem
        self.learn_only(batch, target_raw)
        if self.exchange is not None:       # rainbow_amd.dist.FactoredExchange: FC gradients from all-gathered factors
            m.sync()
            self.exchange.run(m.stream)
        elif self.grad_hook is not None:
            m.sync()
            self.grad_hook(self._as_torch(self.grads))
            L.check(self.lib, self.lib.rb_learner_grads_modified(self.h))
        return self.finish_step()

    def sync_target(self):
        L.check(self.lib, self.lib.rb_learner_sync_target(self.h, self.mem.stream))
        self.mem.sync()

    def debug(self, what, shape, dtype):
        m = self.mem
        out = m.empty(shape, dtype)
        L.check(self.lib, self.lib.rb_learner_debug_read(self.h, what, m.ptr(out), m.stream))
        m.sync()
        return m.download(out)

    def params(self):
        return self._unflat(self.mem.download(self.p_on.detach() if hasattr(self.p_on, "detach") else self.p_on))

    def act(self, state, noisy):
        m = self.mem
        st = m.upload(np.asarray(state, dtype=np.float32))
        a = m.empty((1,), np.int32)
        q = m.empty((1,), np.float32)
        L.check(self.lib, self.lib.rb_learner_act(self.h, m.ptr(st), 1 if noisy else 0, m.ptr(a), m.ptr(q), m.stream))
        m.sync()
        return int(m.download(a)[0]), float(m.download(q)[0])

    def act_batch(self, states, noisy):
        m = self.mem
        n = len(states)
        st = m.upload(np.asarray(states, dtype=np.float32))
        a = m.empty((n,), np.int32)
        q = m.empty((n,), np.float32)
        L.check(self.lib, self.lib.rb_learner_act_batch(self.h, m.ptr(st), n, 1 if noisy else 0, m.ptr(a), m.ptr(q),
                                                        m.stream))
        m.sync()
        return m.download(a).astype(np.int64), m.download(q)
