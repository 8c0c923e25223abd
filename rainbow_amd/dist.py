"""Replica data-parallelism for the learn step (BASELINE config 5).

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI).  Nothing is sharded:
every replica owns its own env stream, HBM replay and noise, and an identical copy of the
parameters + Adam state.  The ONLY exchange per step is one all-reduce (mean) of the flat
float32 gradient buffer between backward (agent.py:96) and clip (agent.py:97); identical inputs
to clip + Adam keep the replicas bit-identical.
"""
import os

import torch
import torch.distributed as dist


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def active():
    """True when the replica exchange must run.  RAINBOW_AMD_FORCE_DIST=1 also runs it in a one-rank group, which is
    how the RCCL plumbing (init, broadcast, all-reduce, re-derived norm) is exercised on a single-GPU box."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("RAINBOW_AMD_FORCE_DIST") == "1"


def average_gradients(flat_grads):
    """In-place mean over replicas of the flat gradient buffer (4*P bytes, one collective)."""
    w = world_size()
    if not active():
        return flat_grads
    if dist.get_backend() == "nccl":
        dist.all_reduce(flat_grads, op=dist.ReduceOp.AVG)
    else:   # gloo (CPU tests) has no AVG
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM)
        flat_grads.div_(w)
    return flat_grads


def broadcast_parameters(flat_params, src=0):
    if active():
        with torch.no_grad():
            dist.broadcast(flat_params, src)
    return flat_params
