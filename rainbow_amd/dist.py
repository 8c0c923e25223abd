"""Replica data-parallelism for the learn step (BASELINE config 5).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Nothing is sharded:
every replica owns its own env stream, HBM replay and noise, and an identical copy of the
parameters + Adam state.  Between backward (agent.py:96) and clip (agent.py:97) every replica
must end up with the MEAN gradient of the global batch; identical inputs to clip + Adam then
keep the replicas bit-identical.

Two exchanges (RAINBOW_AMD_EXCHANGE):

  factored (default) — xGMI is point-to-point (7 links per GPU, a ring all-reduce of the 27.5 MB
      flat gradient pushes 2 x 7/8 x 27.5 = 48 MB through every GPU's links: longer than the
      whole 0.2 ms step).  But 94 % of those bytes are the noisy-linear weight gradients, rank-B
      products dW = dY^T X.  So the replicas ALL-GATHER ONE BLOCK per rank — the factors (dlogits,
      h, dh, feat rows of their batch), their noise vectors and their conv gradients, 1.0 MB at
      the canonical shape — and every replica computes the mean gradient of the global batch from
      the gathered blocks with the same kernel (rb_learner_finish_grads).  One collective per
      step on the learn call's stream, no event, no side stream, 27x fewer bytes on the wire, no
      re-read of the gradient for the norm (the finishing kernel produces the partials).
  allreduce — one all-reduce (mean) of the flat gradient buffer, then the library re-derives
      the norm (rb_learner_grads_modified).  Kept as the reference point.
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from . import _lib as L


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active():
    """True when the replica exchange must run.  RAINBOW_AMD_FORCE_DIST=1 also runs it in a one-rank group, which is
    how the RCCL plumbing (init, broadcast, collectives, finishing kernels) is exercised on a single-GPU box."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("RAINBOW_AMD_FORCE_DIST") == "1"


def mode():
    m = os.environ.get("RAINBOW_AMD_EXCHANGE", "factored")
    if m not in ("factored", "allreduce"):
        raise ValueError("RAINBOW_AMD_EXCHANGE must be 'factored' or 'allreduce', got %r" % m)
    return m


def _mean_reduce(t):
    if dist.get_backend() == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.AVG)
    else:   # gloo (CPU tests) has no AVG
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t.div_(world_size())
    return t


def _all_gather_blocks(out, local):
    """all_gather_into_tensor; the gloo backend (tests: two ranks on ONE device, where RCCL refuses a second rank per GPU)
    gathers host tensors only, so device blocks are staged through the host there — synchronous, never the product path."""
    if dist.get_backend() == "gloo" and local.is_cuda:
        h_all = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(h_all, local.cpu())
        out.copy_(h_all)
    else:
        dist.all_gather_into_tensor(out, local)


def average_gradients(flat_grads):
    """In-place mean over replicas of the flat gradient buffer (4*P bytes, one collective): the 'allreduce' exchange."""
    if active():
        _mean_reduce(flat_grads)
    return flat_grads


def broadcast_parameters(flat_params, src=0):
    if active():
        with torch.no_grad():
            dist.broadcast(flat_params, src)
    return flat_params


def _create_library_comm(lib, dev):
    """An RCCL communicator owned by the library over the ranks of the default process group (None when librccl cannot be
    loaded: the caller then keeps the collective in torch.distributed)."""
    ident = torch.zeros(128, dtype=torch.uint8, device=dev)
    ok = torch.ones(1, dtype=torch.int32, device=dev)
    # EVERY rank probes librccl (rb_comm_available: dlopen + symbol lookup, nothing created) and the flags are reduced before any
    # rank enters ncclCommInitRank: a rank that cannot load the library must not leave the others blocked inside the collective
    # init.  Only rank 0 creates the bootstrap id (ncclGetUniqueId starts a listener thread per id).
    buf = (C.c_ubyte * 128)()
    if not lib.rb_comm_available():
        ok.zero_()
    elif dist.get_rank() == 0:
        if lib.rb_comm_unique_id(buf) != 0:
            ok.zero_()
        else:
            ident.copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        return None       # on all ranks: the collective stays in torch.distributed
    dist.broadcast(ident, 0)
    raw = (C.c_ubyte * 128).from_buffer_copy(bytes(ident.cpu().numpy().tobytes()))
    comm = C.c_void_p()
    with torch.cuda.device(dev):
        rc = lib.rb_comm_create(C.byref(comm), raw, dist.get_world_size(), dist.get_rank())
    # a rank whose communicator could not be created must not leave the others with one: agree, and fall back TOGETHER to the
    # collective in torch.distributed (the step then costs a torch call more, the results are the same)
    made = torch.tensor([1 if rc == 0 and comm.value else 0], dtype=torch.int32, device=dev)
    dist.all_reduce(made, op=dist.ReduceOp.MIN)
    if int(made.item()) == 0:
        if rc == 0 and comm.value:
            lib.rb_comm_destroy(comm)
        return None
    return comm


class FactoredExchange:
    """The 'factored' exchange of one learner handle (see the module docstring and include/rainbow_hip.h).
    `flat_grads` is the learner's gradient buffer as a torch tensor (CUDA in production; a CPU tensor over the
    host-interpreted test build, where streams do not exist and every call is synchronous)."""

    def __init__(self, lib, handle, flat_grads):
        self.lib, self.h = lib, handle
        f, off, n = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        L.check(lib, lib.rb_learner_exchange_layout(handle, C.byref(f), C.byref(off), C.byref(n)))
        self.world = max(world_size(), 1)
        dev = flat_grads.device
        self.cuda = dev.type == "cuda"
        self.local = torch.zeros(f.value, dtype=torch.float32, device=dev)
        # a one-rank group (RAINBOW_AMD_FORCE_DIST) still runs the deferred-gradient path: the library needs world >= 2 to
        # defer, so the lone rank's block is presented twice and the mean of two equal halves is the gradient itself
        self.lib_world = max(self.world, 2)
        self.all = torch.zeros(self.lib_world * f.value, dtype=torch.float32, device=dev)
        L.check(lib, lib.rb_learner_set_exchange(handle, self.lib_world, self.local.data_ptr(), self.all.data_ptr()))
        self.bytes_per_step = 4 * f.value
        # RCCL inside the library (rb_learner_exchange_rccl: ncclAllGather + the finishing launch as ONE C call, no
        # torch.distributed on the step's path).  The communicator's 128-byte id travels through the process group that
        # already exists; RAINBOW_AMD_EXCHANGE_VIA=torch keeps the collective in torch.distributed (the fallback, and what
        # the gloo test groups use).
        self.comm = None
        if self.cuda and dist.get_backend() == "nccl" and os.environ.get("RAINBOW_AMD_EXCHANGE_VIA", "rccl") == "rccl":
            self.comm = _create_library_comm(lib, dev)

    def __del__(self):
        try:
            if getattr(self, "comm", None):
                self.lib.rb_comm_destroy(self.comm)
                self.comm = None
        except Exception:
            pass

    def close(self):
        if self.h:
            self.lib.rb_learner_set_exchange(self.h, 1, None, None)
            self.h = None
        if getattr(self, "comm", None):
            if getattr(self, "cuda", False):
                torch.cuda.synchronize(self.local.device)      # an all-gather may still be in flight on the learn call's stream
            self.lib.rb_comm_destroy(self.comm)
            self.comm = None

    def run(self, stream_handle=None):
        """Call right after rb_learner_learn*: ONE all-gather of the per-rank blocks on the current stream (it follows the
        learn call's launches in stream order), then the library finishes the gradients of the global batch."""
        if self.cuda:
            stream_handle = torch.cuda.current_stream(self.local.device).cuda_stream
        if self.comm is not None:
            rc = self.lib.rb_learner_exchange_rccl(self.h, self.comm, stream_handle)
            if rc != 0:
                L.check(self.lib, rc)
            return
        if self.lib_world != self.world:      # one-rank plumbing run
            n = self.local.numel()
            _all_gather_blocks(self.all[:n], self.local)
            self.all[n:2 * n].copy_(self.all[:n])
        else:
            _all_gather_blocks(self.all, self.local)
        L.check(self.lib, self.lib.rb_learner_finish_grads(self.h, stream_handle))
