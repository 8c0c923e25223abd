"""Drop-in `ReplayMemory` for the reference's main.py / test.py (replaces /root/reference/memory.py).

Same constructor, attributes and methods as the reference class (memory.py:91-180), but the ring
buffer, the float32 sum-tree, the stratified sampler, the frame-stack gather and the priority
update all live in HBM behind librainbow_hip.so (C ABI: include/rainbow_hip.h).  Python only
forwards pointers; there is no CPU path.

Two ways to consume a batch:
  * `sample(batch_size)` — the reference's 7-tuple (memory.py:148-155), states as float32 /255.
    Needs one D2H copy of the tree indices because the reference returns them as a numpy array.
  * `sample_device(batch_size)` — device-resident outputs (uint8 frame stacks, no sync); this is
    what rainbow_amd.agent.Agent.learn uses.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L


class _TransitionsView:
    """Read-only stand-in for the reference's `mem.transitions` SegmentTree attribute."""

    def __init__(self, owner):
        self._o = owner

    def _hdr(self):
        return self._o._header()

    def _pos(self):
        idx, full = C.c_int64(0), C.c_int32(0)
        L.check(self._o._lib, self._o._lib.rb_replay_position(self._o._h, C.byref(idx), C.byref(full)))
        return int(idx.value), bool(full.value)

    @property
    def index(self):
        """memory.py:14 — from the library's host mirror: no device synchronisation (safe to poll in the env loop)."""
        return self._pos()[0]

    @property
    def full(self):
        """memory.py:16 — host mirror, no synchronisation."""
        return self._pos()[1]

    @property
    def max(self):
        """memory.py:20 — lives on the device (updated by priority write-backs): SYNCHRONISES the stream."""
        return float(self._hdr().max)

    def total(self):
        """memory.py:88-89 — tree root: SYNCHRONISES the stream."""
        return float(self._hdr().total)


class ReplayMemory:
    # The reference redraws until the batch is valid (memory.py:128-132).  The device loop is bounded, but generously: a
    # redraw costs ~4 us inside the one sampler launch, and hitting the bound means no valid batch exists at all (buffer
    # too small for the batch).  Then the batch gets zero importance weights (a zero-gradient step) and
    # failed_samples() counts it; Agent.learn raises on the next call.
    MAX_ATTEMPTS = 1024
    # update_priorities() is LAZY (RAINBOW_AMD_LAZY_PRIORITIES=0: immediate): the write-back of learn step k and the draw of
    # step k + 1 are a dependent pair of single-workgroup launches, so the call only records its operands and the next
    # sample_device() issues both as ONE launch (rb_replay_update_sample: bit-identical tree, header and batch).  Anything
    # else that reads or writes the tree — append, the header, state dumps, the raw handle `_h` — applies the pending
    # write-back first.  Operands are copied at the call — numpy arrays (the reference's call site, agent.py:100) by their
    # upload, device tensors into a two-slot staging buffer this object owns (_stage_operands) — so the caller may reuse its
    # loss / index tensors at once, as with the reference's immediate update.
    _pending = None
    _stage = None
    _handle = None
    _lazy = os.environ.get("RAINBOW_AMD_LAZY_PRIORITIES", "1") != "0"

    @property
    def _h(self):
        if self._pending is not None:
            self.flush()
        return self._handle

    @_h.setter
    def _h(self, value):
        self._handle = value

    def flush(self):
        """Apply a pending update_priorities() now (a launch of its own)."""
        pend, self._pending = self._pending, None
        if pend is not None:
            L.check(self._lib, self._lib.rb_replay_update_priorities(self._handle, pend[0].data_ptr(), pend[1].data_ptr(),
                                                                     int(pend[0].numel()), self._stream()))

    def __init__(self, args, capacity, seed=None):
        self.device = torch.device(args.device)
        if self.device.type != "cuda":
            raise RuntimeError("rainbow_amd.ReplayMemory lives in HBM: args.device must be a cuda (ROCm) device, got %s"
                               % self.device)
        self._lib = L.load()
        self.capacity = int(capacity)
        self.history = int(args.history_length)
        self.discount = float(args.discount)
        self.n = int(args.multi_step)
        self.priority_weight = float(args.priority_weight)      # beta, annealed by the caller (main.py:161)
        self.priority_exponent = float(args.priority_exponent)
        self.t = 0                                              # episode timestep counter (memory.py:100)
        self._seed = int(seed if seed is not None else np.random.randint(0, 2 ** 31 - 1))
        self._lazy = os.environ.get("RAINBOW_AMD_LAZY_PRIORITIES", "1") != "0"
        self._pending = None
        self._stage = {}
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self._lib, self._lib.rb_replay_create(C.byref(self._h), self.capacity, self.history, self.n,
                                                          self.discount, self.priority_exponent, self._seed))
        self.transitions = _TransitionsView(self)
        self._out = {}
        self._ptr_cache = {}
        self.current_idx = 0
        self._init_beta_source()

    def _init_beta_source(self):
        # -beta lives in HBM so a captured hipGraph sees main.py:161's annealing (by-value kernel arguments freeze)
        self._neg_beta_dev = torch.full((1,), -float(self.priority_weight), dtype=torch.float32, device=self.device)
        self._neg_beta_val = float(self.priority_weight)
        L.check(self._lib, self._lib.rb_replay_set_beta_source(self._handle, self._neg_beta_dev.data_ptr()))

    def _sync_beta(self):
        if float(self.priority_weight) != self._neg_beta_val:
            self._neg_beta_val = float(self.priority_weight)
            self._neg_beta_dev.fill_(-self._neg_beta_val)     # float32(-beta): NEP-50 weak scalar, memory.py:153

    # ------------------------------------------------------------------ plumbing
    def __del__(self):
        try:
            self._pending = None
            if getattr(self, "_handle", None):
                self._lib.rb_replay_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def _stream(self):
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)      # one C call instead of a Stream object (~4 us) per launch
        if raw is not None:
            return raw(self.device.index if self.device.index is not None else torch.cuda.current_device())
        return torch.cuda.current_stream(self.device).cuda_stream

    def _header(self):
        hdr = L.ReplayHeader()
        L.check(self._lib, self._lib.rb_replay_header(self._h, C.byref(hdr), self._stream()))
        return hdr

    def _buffers(self, batch):
        key = int(batch)
        if key not in self._out:
            self._ptr_cache = {k: v for k, v in self._ptr_cache.items() if k[0] != key}
            d, h = self.device, self.history
            self._out[key] = dict(
                tree_idxs=torch.empty(batch, dtype=torch.int64, device=d),
                states=torch.empty(batch, h, 84, 84, dtype=torch.uint8, device=d),
                next_states=torch.empty(batch, h, 84, 84, dtype=torch.uint8, device=d),
                actions=torch.empty(batch, dtype=torch.int64, device=d),
                returns=torch.empty(batch, dtype=torch.float32, device=d),
                nonterminals=torch.empty(batch, dtype=torch.float32, device=d),
                weights=torch.empty(batch, dtype=torch.float32, device=d))
        return self._out[key]

    # ------------------------------------------------------------------ reference API
    def append(self, state, action, reward, terminal):
        """memory.py:105-108.  `state` float32 [h,84,84] in [0,1] on the device (env.py:52,77)."""
        st = state
        if st.dtype != torch.float32 or st.device != self.device or not st.is_contiguous():     # (env.py hands over exactly this)
            st = state.to(device=self.device, dtype=torch.float32).contiguous()
        rc = self._lib.rb_replay_append(self._h, st.data_ptr(), int(self.t), int(action), float(reward),
                                        0 if terminal else 1, self._stream())
        if rc != 0:
            L.check(self._lib, rc)
        self.t = 0 if terminal else self.t + 1

    def append_batch(self, frames_u8, actions, rewards, terminals):
        """n sequential appends of already-quantised last frames (uint8 [n,84,84], device)."""
        n = int(frames_u8.shape[0])
        terminals = np.asarray(terminals, dtype=bool)
        ts = np.empty(n, dtype=np.int32)
        t = self.t
        for i in range(n):
            ts[i] = t
            t = 0 if terminals[i] else t + 1
        self.t = t
        d = self.device
        fr = frames_u8.to(device=d, dtype=torch.uint8).contiguous()
        ts_d = torch.from_numpy(ts).to(d)
        ac_d = torch.as_tensor(np.asarray(actions, dtype=np.int32)).to(d)
        rw_d = torch.as_tensor(np.asarray(rewards, dtype=np.float32)).to(d)
        nt_d = torch.from_numpy((~terminals).astype(np.uint8)).to(d)
        L.check(self._lib, self._lib.rb_replay_append_batch(self._h, fr.data_ptr(), ts_d.data_ptr(), ac_d.data_ptr(),
                                                            rw_d.data_ptr(), nt_d.data_ptr(), n, self._stream()))
        torch.cuda.current_stream(d).synchronize()   # the temporaries above must outlive the kernels

    def failed_samples(self):
        """Sampler launches completed so far that found no valid batch within MAX_ATTEMPTS (read from pinned host memory
        the kernel writes: no synchronisation)."""
        n = C.c_int64(0)
        L.check(self._lib, self._lib.rb_replay_failed_samples(self._handle, C.byref(n)))
        return int(n.value)

    def dropped_updates(self):
        """Priority write-backs dropped so far because their indices came from a draw that gave up (pinned host word: no
        synchronisation).  Zeroed by reset_failed_samples()."""
        n = C.c_int64(0)
        L.check(self._lib, self._lib.rb_replay_dropped_updates(self._handle, C.byref(n)))
        return int(n.value)

    def expired_waits(self):
        """Cross-stream waits of the early draw (RB_OPTS spec_draw=1, off by default) that hit their ~2 ms bound so far (pinned
        host word: no synchronisation).  Each one failed safe (the waiting launch drew itself / dropped the write-back) and
        switched the early draw off on this replay until reset_failed_samples().  0 on a healthy device."""
        n = C.c_int64(0)
        L.check(self._lib, self._lib.rb_replay_expired_waits(self._handle, C.byref(n)))
        return int(n.value)

    def reset_failed_samples(self):
        """Zero those counters (the failure has been reported to the caller); allows the early draw again."""
        L.check(self._lib, self._lib.rb_replay_reset_failed_samples(self._handle))

    def frame_source(self):
        """(frames_ptr, windows_ptr, window_len): lets the learner read frames straight from the ring (zero-copy)."""
        if not hasattr(self, "_bufs"):
            self._bufs = L.ReplayBuffers()
            L.check(self._lib, self._lib.rb_replay_buffers(self._h, C.byref(self._bufs)))
        return self._bufs.frames_dev, self._bufs.window_dev, int(self._bufs.window_len)

    def sample_device(self, batch_size, unit_uniforms=None, gather=True, noise_job=None, stream=None):
        """Device-resident batch: dict(tree_idxs i64[B], states u8[B,h,84,84], next_states u8, actions i64[B],
        returns f32[B], nonterminals f32[B], weights f32[B]).  Asynchronous.  unit_uniforms (float64 device
        tensor [attempts,B]) injects the sampler's random numbers for parity tests.  gather=False skips the
        frame-stack copies (states/next_states are then stale): the consumer reads the ring via frame_source().
        noise_job (rainbow_amd._lib.NoiseJob from the learner) lets the launch also carry the noise resample."""
        o = self._buffers(batch_size)
        if float(self.priority_weight) != self._neg_beta_val and not torch.cuda.is_current_stream_capturing():
            self._sync_beta()
        uu_ptr, attempts = None, self.MAX_ATTEMPTS
        if unit_uniforms is not None:
            self._uu = unit_uniforms.to(device=self.device, dtype=torch.float64).contiguous()
            uu_ptr, attempts = self._uu.data_ptr(), int(self._uu.shape[0])
        # the output buffers are persistent: their addresses are looked up once per (batch, gather), not per call (the
        # host side of a learn step is about as long as its GPU side at batch 32 — every microsecond here is on the clock)
        key = (int(batch_size), bool(gather))
        ptrs = self._ptr_cache.get(key)
        if ptrs is None:
            ptrs = (o["tree_idxs"].data_ptr(), o["states"].data_ptr() if gather else None,
                    o["next_states"].data_ptr() if gather else None, o["actions"].data_ptr(), o["returns"].data_ptr(),
                    o["nonterminals"].data_ptr(), o["weights"].data_ptr())
            self._ptr_cache[key] = ptrs
        if stream is None:
            stream = self._stream()
        pend = self._pending
        if noise_job is not None:
            if pend is not None:      # (the noise-carrying launch has no write-back slot: apply it first, in stream order)
                self.flush()
            rc = self._lib.rb_replay_sample_fused_noise(self._h, key[0], float(self.priority_weight), uu_ptr, attempts, *ptrs,
                                                        C.byref(noise_job), stream)
        elif pend is not None:        # the pending write-back and this draw as one launch
            self._pending = None
            rc = self._lib.rb_replay_update_sample(self._handle, pend[0].data_ptr(), pend[1].data_ptr(), int(pend[0].numel()), key[0],
                                                   float(self.priority_weight), uu_ptr, attempts, *ptrs, stream)
            self._applied = pend      # (its operands stay alive until the launch after this one)
        else:
            rc = self._lib.rb_replay_sample(self._handle, key[0], float(self.priority_weight), uu_ptr, attempts, *ptrs, stream)
        if rc != 0:
            L.check(self._lib, rc)
        return o

    def sample(self, batch_size):
        """memory.py:148-155 7-tuple: (tree_idxs ndarray, states f32, actions i64, returns f32, next_states f32,
        nonterminals f32[B,1], weights f32)."""
        o = self.sample_device(batch_size)
        states = torch.empty(o["states"].shape, dtype=torch.float32, device=self.device)
        next_states = torch.empty_like(states)
        n = states.numel()
        L.check(self._lib, self._lib.rb_u8_to_unit_f32(o["states"].data_ptr(), states.data_ptr(), n, self._stream()))
        L.check(self._lib, self._lib.rb_u8_to_unit_f32(o["next_states"].data_ptr(), next_states.data_ptr(), n, self._stream()))
        tree_idxs = o["tree_idxs"].cpu().numpy()      # the reference hands indices back as numpy (memory.py:155)
        hdr = self._header()
        if hdr.last_status != 0:
            raise RuntimeError("ReplayMemory.sample: no valid batch in %d attempts (buffer too small for batch?)"
                               % hdr.last_attempts)
        return (tree_idxs, states, o["actions"].clone(), o["returns"].clone(), next_states,
                o["nonterminals"].clone().unsqueeze(1), o["weights"].clone())

    def _stage_operands(self, idxs, priorities, stage_i, stage_p):
        """The operands of a deferred write-back in memory THIS object owns: `.to()` / `.contiguous()` hand a matching device
        tensor back as it is, and the reference's semantics are immediate (memory.py:157-159) — a caller may reuse its loss
        or index tensor right after the call.  Two slots per length, used in turn; an asynchronous device-to-device copy of
        <= 3 KB per operand that needs one (each costs the PER-only loop ~1.7 us per batch: `donate=True` skips them)."""
        n = int(idxs.numel())
        ring = self._stage.get(n)
        if ring is None:
            ring = self._stage[n] = [[torch.empty(n, dtype=torch.int64, device=self.device),
                                      torch.empty(n, dtype=torch.float32, device=self.device)] for _ in range(2)] + [0]
        slot = ring[ring[2]]
        ring[2] ^= 1
        if stage_i:
            slot[0].copy_(idxs.reshape(-1), non_blocking=True)
            idxs = slot[0]
        if stage_p:
            slot[1].copy_(priorities.reshape(-1), non_blocking=True)
            priorities = slot[1]
        return idxs, priorities

    def _is_own_index_buffer(self, t):
        for o in self._out.values():
            if o["tree_idxs"] is t:
                return True
        return False

    def update_priorities(self, idxs, priorities, _immediate=False, donate=False):
        """memory.py:157-159.  Accepts numpy arrays (reference call site agent.py:100) or device tensors.  By default the
        write-back is recorded and rides in the next sample_device() launch (RAINBOW_AMD_LAZY_PRIORITIES); caller-owned device
        operands are copied at the call, so the caller's tensors are free again when this returns, as in the reference.
        donate=True (an extension): the caller will not modify its device operands before the next sample_device() / flush() —
        they are read in place when the write-back runs (no staging copy)."""
        d = self.device
        if not torch.is_tensor(idxs):
            idxs = torch.as_tensor(np.asarray(idxs, dtype=np.int64))
        if not torch.is_tensor(priorities):
            priorities = torch.as_tensor(np.asarray(priorities, dtype=np.float32))
        if self._pending is not None:
            self.flush()
        lazy = self._lazy and not _immediate and not torch.cuda.is_current_stream_capturing()
        if lazy:
            own_i = idxs.device != d or idxs.dtype != torch.int64 or not idxs.is_contiguous()
            own_p = priorities.device != d or priorities.dtype != torch.float32 or not priorities.is_contiguous()
            i_d = idxs.to(device=d, dtype=torch.int64).contiguous()
            p_d = priorities.to(device=d, dtype=torch.float32).contiguous()
            # (the index buffer sample_device() hands out is this object's: the launch that applies the write-back reads it
            # before its sampler part refills it)
            stage_i = not own_i and not donate and not self._is_own_index_buffer(idxs)
            stage_p = not own_p and not donate
            if stage_i or stage_p:          # an operand is still the caller's own device tensor
                i_d, p_d = self._stage_operands(i_d, p_d, stage_i, stage_p)
            self._upd = self._pending = (i_d, p_d)
            return
        self._upd = (idxs.to(device=d, dtype=torch.int64).contiguous(),
                     priorities.to(device=d, dtype=torch.float32).contiguous())
        L.check(self._lib, self._lib.rb_replay_update_priorities(self._handle, self._upd[0].data_ptr(),
                                                                 self._upd[1].data_ptr(), int(self._upd[0].numel()),
                                                                 self._stream()))

    # validation iterator (memory.py:162-180)
    def __iter__(self):
        self.current_idx = 0
        return self

    def __next__(self):
        if self.current_idx == self.capacity:
            raise StopIteration
        out = torch.empty(self.history, 84, 84, dtype=torch.float32, device=self.device)
        L.check(self._lib, self._lib.rb_replay_state_at(self._handle, int(self.current_idx), out.data_ptr(), self._stream()))
        self.current_idx += 1
        return out

    next = __next__

    def states_at(self, indices):
        """[__next__ at index i for i in indices] as ONE launch: float32 [n, h, 84, 84] on the device (memory.py:167-178).
        `indices`: int64 device tensor or anything torch.as_tensor accepts, each in [0, capacity)."""
        idx = torch.as_tensor(indices, dtype=torch.int64).to(self.device).contiguous()
        n = int(idx.numel())
        if n and (int(idx.min()) < 0 or int(idx.max()) >= self.capacity):
            raise IndexError("states_at: index out of range")
        out = torch.empty(n, self.history, 84, 84, dtype=torch.float32, device=self.device)
        self._idx_keep = idx
        L.check(self._lib, self._lib.rb_replay_states_at(self._handle, idx.data_ptr(), n, out.data_ptr(), self._stream()))
        return out

    # ------------------------------------------------------------------ streaming dump / restore
    _COLUMNS = ("tree", "frames", "timestep", "action", "reward", "nonterminal")

    def _column_spec(self):
        b = L.ReplayBuffers()
        L.check(self._lib, self._lib.rb_replay_buffers(self._h, C.byref(b)))
        cap = self.capacity
        return b, {"tree": (b.sum_tree_dev, b.tree_len * 4), "frames": (b.frames_dev, cap * 7056),
                   "timestep": (b.timestep_dev, cap * 4), "action": (b.action_dev, cap * 4),
                   "reward": (b.reward_dev, cap * 4), "nonterminal": (b.nonterminal_dev, cap)}

    def save_to(self, fileobj, chunk_bytes=64 << 20):
        """Streams the replay to a binary file object in `chunk_bytes` pieces (64 MB of host staging instead of the 7 GB
        a pickle of a 1M-capacity memory needs; main.py:94-100 `pickle.dump(memory, bz2 file)` still works through
        __getstate__, this is the scalable alternative: `with bz2.open(path, 'wb') as f: mem.save_to(f)`)."""
        import json
        import struct
        b, spec = self._column_spec()
        hdr = self._header()
        meta = dict(version=1, capacity=self.capacity, history=self.history, n=self.n, discount=self.discount,
                    priority_weight=self.priority_weight, priority_exponent=self.priority_exponent, t=self.t,
                    seed=self._seed, columns={k: spec[k][1] for k in self._COLUMNS}, header=bytes(hdr).hex())
        blob = json.dumps(meta).encode()
        fileobj.write(b"RBRPLY01" + struct.pack("<q", len(blob)) + blob)
        stage = np.empty(chunk_bytes, dtype=np.uint8)
        for k in self._COLUMNS:
            ptr, nbytes = spec[k]
            for lo in range(0, nbytes, chunk_bytes):
                m = min(chunk_bytes, nbytes - lo)
                L.check(self._lib, self._lib.rb_copy_to_host(stage.ctypes.data, ptr + lo, m, self._stream()))
                fileobj.write(stage[:m].tobytes() if m < chunk_bytes else stage.data)

    @classmethod
    def load_from(cls, fileobj, device, chunk_bytes=64 << 20):
        """Inverse of save_to: builds a ReplayMemory on `device` from the stream."""
        import json
        import struct
        import types
        magic = fileobj.read(8)
        if magic != b"RBRPLY01":
            raise ValueError("not a rainbow_amd replay stream")
        (n,) = struct.unpack("<q", fileobj.read(8))
        meta = json.loads(fileobj.read(n).decode())
        args = types.SimpleNamespace(device=device, history_length=meta["history"], discount=meta["discount"],
                                     multi_step=meta["n"], priority_weight=meta["priority_weight"],
                                     priority_exponent=meta["priority_exponent"])
        mem = cls(args, meta["capacity"], seed=meta["seed"])
        mem.t = meta["t"]
        b, spec = mem._column_spec()
        for k in cls._COLUMNS:
            ptr, nbytes = spec[k]
            if nbytes != meta["columns"][k]:
                raise ValueError("replay stream column %s has %d bytes, expected %d" % (k, meta["columns"][k], nbytes))
            for lo in range(0, nbytes, chunk_bytes):
                m = min(chunk_bytes, nbytes - lo)
                buf = fileobj.read(m)
                if len(buf) != m:
                    raise EOFError("replay stream truncated in column %s" % k)
                a = np.frombuffer(buf, dtype=np.uint8)
                L.check(mem._lib, mem._lib.rb_copy_to_device(ptr + lo, a.ctypes.data, m, mem._stream()))
        hdr = np.frombuffer(bytes.fromhex(meta["header"]), dtype=np.uint8).copy()
        L.check(mem._lib, mem._lib.rb_copy_to_device(b.header_dev, hdr.ctypes.data, hdr.nbytes, mem._stream()))
        return mem

    # ------------------------------------------------------------------ pickling (main.py:94-100,118)
    def _grab(self, field, start=0, count=None):
        """Copies rows [start, start+count) of one device column to a numpy array (tests / partial dumps)."""
        b = L.ReplayBuffers()
        L.check(self._lib, self._lib.rb_replay_buffers(self._h, C.byref(b)))
        spec = {"tree": (b.sum_tree_dev, 4, np.float32, b.tree_len), "frames": (b.frames_dev, 7056, np.uint8, self.capacity),
                "timestep": (b.timestep_dev, 4, np.int32, self.capacity), "action": (b.action_dev, 4, np.int32, self.capacity),
                "reward": (b.reward_dev, 4, np.float32, self.capacity), "nonterminal": (b.nonterminal_dev, 1, np.uint8, self.capacity)}
        ptr, row, dtype, rows = spec[field]
        count = rows - start if count is None else count
        host = np.empty(count * row, dtype=np.uint8)
        L.check(self._lib, self._lib.rb_copy_to_host(host.ctypes.data, ptr + start * row, host.nbytes, self._stream()))
        out = host.view(dtype)
        return out.reshape(count, 84, 84) if field == "frames" else out

    def _dump(self):
        b = L.ReplayBuffers()
        L.check(self._lib, self._lib.rb_replay_buffers(self._h, C.byref(b)))
        hdr = self._header()
        cap = self.capacity

        def grab(ptr, nbytes):
            host = np.empty(nbytes, dtype=np.uint8)
            L.check(self._lib, self._lib.rb_copy_to_host(host.ctypes.data, ptr, nbytes, self._stream()))
            return host

        return dict(tree=grab(b.sum_tree_dev, b.tree_len * 4).view(np.float32),
                    frames=grab(b.frames_dev, cap * 7056), timestep=grab(b.timestep_dev, cap * 4).view(np.int32),
                    action=grab(b.action_dev, cap * 4).view(np.int32), reward=grab(b.reward_dev, cap * 4).view(np.float32),
                    nonterminal=grab(b.nonterminal_dev, cap),
                    header=bytes(hdr))

    def __getstate__(self):
        dump = self._dump()       # (applies a pending priority write-back first)
        st = {k: v for k, v in self.__dict__.items()
              if k not in ("_lib", "_h", "_handle", "_pending", "_applied", "_stage", "transitions", "_out", "_uu", "_upd", "_neg_beta_dev",
                           "_neg_beta_val", "_bufs", "_idx_keep")}
        st["device"] = str(self.device)
        st["_dump"] = dump
        return st

    def __setstate__(self, st):
        dump = st.pop("_dump")
        self.__dict__.update(st)
        self.device = torch.device(self.device)
        self._lib = L.load()
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            L.check(self._lib, self._lib.rb_replay_create(C.byref(self._h), self.capacity, self.history, self.n,
                                                          self.discount, self.priority_exponent, self._seed))
        b = L.ReplayBuffers()
        L.check(self._lib, self._lib.rb_replay_buffers(self._h, C.byref(b)))
        for key, ptr in (("tree", b.sum_tree_dev), ("frames", b.frames_dev), ("timestep", b.timestep_dev),
                         ("action", b.action_dev), ("reward", b.reward_dev), ("nonterminal", b.nonterminal_dev)):
            a = np.ascontiguousarray(dump[key])
            L.check(self._lib, self._lib.rb_copy_to_device(ptr, a.ctypes.data, a.nbytes, self._stream()))
        hdr = np.frombuffer(dump["header"], dtype=np.uint8).copy()
        L.check(self._lib, self._lib.rb_copy_to_device(b.header_dev, hdr.ctypes.data, hdr.nbytes, self._stream()))
        self.transitions = _TransitionsView(self)
        self._out = {}
        self._ptr_cache = {}
        self._pending = None
        self._stage = {}
        self._init_beta_source()
        self._header()   # resynchronise the library's host mirror of index/full
