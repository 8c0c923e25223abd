#!/usr/bin/env python
"""bench.py — Rainbow learn-step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  N > 1 without a launcher: the script re-executes itself under torch.distributed.run with N ranks (one per GPU) and
  fails loudly when fewer than N GPUs are visible; under a launcher (WORLD_SIZE set) it checks WORLD_SIZE == N.

One "step" = one complete gradient step of the hot path on synthetic transitions already
resident in HBM:  online noise resample -> Agent.learn(mem) = PER sample (sum-tree search, frames
read in place) -> 3 forwards -> C51 projection / loss -> backward -> [replica exchange over RCCL]
-> global-norm clip -> Adam -> priority update.          (reference: main.py:151,164; agent.py:61-100)

Workload (SURVEY §8d / BASELINE.md §3): config 2 — canonical network, batch 32, 51 atoms,
6 actions, n=3, 1M-capacity replay filled to capacity with the write head mid-buffer,
non-uniform priorities.  Multi-GPU = independent replicas (own replay, own noise) that exchange
gradient information once per step (rainbow_amd/dist.py); weak scaling (per-GPU work fixed).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline        — the step's dominant kernel by time (chosen by a bracketed probe over every candidate),
                    timed live with HIP events on its launch stream over the timed region (rb_profile_*)
  roofline_others — every other candidate kernel, timed the same way in short passes after the timed region
  roofline_step   — the whole step against SURVEY §8d's per-step algorithmic bytes / FLOPs
  roofline_per    — the second BASELINE metric (PER samples/s: sample + update_priorities, no learner) against HBM
  cpu_baseline    — the CPU oracle (a port of the reference's algorithm: numpy replay + torch-CPU
                    learner) timed on this box's host cores on a bounded sample (rank 0, N=1 only)
  library_source_hash — hash of the sources the loaded librainbow_hip.so was built from; must equal the tree's
--no-profile: no HIP-event bracket anywhere (the run a rocprofv3 kernel trace should see).
Runs shorter than 24 steps bracket 1 launch in ceil(steps / 3) of the dominant kernel (three samples: an event pair idles the stream for
≈ 13 µs — five pairs were 3.4 µs per step of a 20-step run); longer runs 1 in 8.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PROFILE_STRIDE = 8         # the dominant kernel's launches bracketed by HIP events inside the timed region: 1 in 8
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
F32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak

CONFIGS = {
    # BASELINE.json configs[1]
    "pong-canonical-b32": dict(architecture="canonical", hidden_size=512, batch_size=32, actions=6, multi_step=3,
                               capacity=1_000_000),
    # configs[2]
    "breakout-canonical-b256": dict(architecture="canonical", hidden_size=512, batch_size=256, actions=4, multi_step=3,
                                    capacity=1_000_000),
    # configs[3]
    "data-efficient-b32": dict(architecture="data-efficient", hidden_size=256, batch_size=32, actions=6, multi_step=20,
                               capacity=100_000),
}


def make_args(cfg, device):
    return types.SimpleNamespace(
        device=device, history_length=4, discount=0.99, multi_step=cfg["multi_step"], priority_weight=0.4,
        priority_exponent=0.5, atoms=51, V_min=-10.0, V_max=10.0, batch_size=cfg["batch_size"], norm_clip=10.0,
        model=None, learning_rate=6.25e-5, adam_eps=1.5e-4, architecture=cfg["architecture"],
        hidden_size=cfg["hidden_size"], noisy_std=0.1)


def fill_replay(mem, capacity, actions, seed):
    """Synthetic transitions per BASELINE.md §3: uniform u8 frames, uniform actions, rewards {-1,0,+1} with
    P={.05,.90,.05}, terminal p=1/1000; filled past capacity (write head mid-buffer); priorities from
    |N(0,1)|+1e-3 losses (seed 1)."""
    dev = mem.device
    g = torch.Generator(device=dev).manual_seed(seed)
    rs = np.random.RandomState(seed)
    total = capacity + capacity // 2
    chunk = 65536
    done = 0
    while done < total:
        n = min(chunk, total - done)
        frames = torch.randint(0, 256, (n, 84, 84), dtype=torch.uint8, device=dev, generator=g)
        mem.append_batch(frames, rs.randint(0, actions, n), rs.choice([-1.0, 0.0, 1.0], size=n, p=[0.05, 0.9, 0.05]),
                         rs.random_sample(n) < 1e-3)
        done += n
    g1 = torch.Generator(device=dev).manual_seed(1)
    tree_start = 2 ** int(capacity - 1).bit_length() - 1
    for lo in range(0, capacity, 1024):
        n = min(1024, capacity - lo)
        idx = torch.arange(lo, lo + n, device=dev, dtype=torch.int64) + tree_start
        mem.update_priorities(idx, torch.randn(n, device=dev, generator=g1).abs() + 1e-3)
    torch.cuda.synchronize(dev)


CONV_GEOM = {   # (cin, cout, kernel, stride, in, out) per layer, model.py:55-63
    "canonical": [(4, 32, 8, 4, 84, 20), (32, 64, 4, 2, 20, 9), (64, 64, 3, 1, 9, 7)],
    "data-efficient": [(4, 32, 5, 5, 84, 16), (32, 64, 5, 5, 16, 3)],
}


def kernel_table(cfg, n_params, hosted_update=False, implicit_sigma=False):
    """Algorithmic work per launch of the kernels that can dominate a step, with the roofline that bounds
    each: the streamed hidden layer and the optimiser pass are HBM-bound, the conv kernels f32-MFMA-bound.
    Keys are the profiling tags of the launches (RB_LAUNCH_T in csrc/learner.hip)."""
    B, H = cfg["batch_size"], cfg["hidden_size"]
    conv = CONV_GEOM[cfg["architecture"]]
    F = conv[-1][1] * conv[-1][5] ** 2
    wh = 2 * H * F * 4                      # one of mu / sigma of the fused hidden layer, bytes
    fused_dw = os.environ.get("RAINBOW_AMD_FUSED_DW", "0") == "1" and B <= 32   # fc_h weight gradient never stored
    # RB_LEARNER_IMPLICIT_SIGMA (Agent default with the deferred pass; hidden layers from 1 M elements; batch <= 32 or >= 128): the
    # hidden layer's sigma gradient is neither written by the backward nor read by the optimiser pass — wh bytes less in each
    sig = wh if (implicit_sigma and hosted_update and not fused_dw and 2 * H * F >= (1 << 20) and (B <= 32 or B >= 128)) else 0
    def binding(nbytes, flops):
        """The roofline that binds a streamed noisy-linear launch: HBM at batch 32 (weights streamed once for a rank-32
        product), f32 MFMA at batch 256 (the same bytes carry 8x the FLOPs)."""
        if nbytes / (HBM_PEAK_GBS * 1e9) >= flops / (F32_MFMA_PEAK_TF * 1e12):
            return dict(bound="hbm", work=nbytes, unit="GB/s")
        return dict(bound="mfma", work=flops, unit="TFLOP/s")

    t = {
        # clip + Adam over the flat buffers: reads p, g, m, v and writes p, m, v once (fused: no g read for the fc_h weights)
        "clip_adam": dict(bound="hbm", work=7 * 4 * n_params - (2 * wh if fused_dw else 0) - sig, unit="GB/s"),
        # hidden layer forward: streams mu+sigma of BOTH nets once; activations are L2-resident; 3B rows x F x 2H MACs
        "fc_h_fwd": binding(2 * 2 * wh + 3 * B * F * 4 + 3 * B * 2 * H * 4 * 2, 2 * 3 * B * F * 2 * H),
        # hidden layer backward (one launch): streams mu+sigma of the online net once (dX), writes d_mu + d_sigma once (dW);
        # dW and dX are B x F x 2H MACs each
        "fc_h_bwd": binding(2 * wh + (0 if fused_dw else 2 * wh) - sig + B * (F + 2 * H) * 4, 2 * 2 * B * F * 2 * H),
    }
    # the output-layer / head tail (three small launches; HBM-bound by their bytes, latency-bound in fact: DESIGN.md §3)
    NZ = 51 * (cfg["actions"] + 1)
    wz = NZ * H * 4                         # one of mu / sigma of the output layer, bytes
    t["fc_z_fwd"] = dict(bound="hbm", work=2 * 2 * wz + 3 * B * (2 * H + NZ) * 4, unit="GB/s")
    t["head"] = dict(bound="hbm", work=3 * B * NZ * 4 + 2 * B * NZ * 4, unit="GB/s")
    t["fc_z_bwd"] = dict(bound="hbm", work=2 * wz + 2 * wz + B * (NZ + 2 * 2 * H) * 4, unit="GB/s")
    dw = 0
    for i, (cin, cout, ks, _s, _ih, oh) in enumerate(conv):
        K, P = cin * ks * ks, oh * oh
        t["conv%d_fwd" % (i + 1)] = dict(bound="mfma", work=2 * 3 * B * cout * P * K, unit="TFLOP/s")     # 3B images
        if i > 0:
            t["conv%d_dx" % (i + 1)] = dict(bound="mfma", work=2 * B * cout * P * K, unit="TFLOP/s")
        dw += 2 * B * cout * P * (K + 1)
    t["conv_dw_all"] = dict(bound="mfma", work=dw, unit="TFLOP/s")
    # PER sampler: one workgroup, a chain of dependent loads — its useful bytes are tiny (tree nodes on the search paths,
    # window scalars), so its HBM fraction is ~0 by construction: it is listed because at the data-efficient config it is
    # the longest launch of the step (latency-bound)
    L = (cfg["capacity"] - 1).bit_length()
    t["sample"] = dict(bound="hbm", work=B * (4 * L + 4 * (4 + cfg["multi_step"]) + 4 * cfg["multi_step"] + 12), unit="GB/s")
    if hosted_update:
        # RAINBOW_AMD_DEFER_UPDATE (default): the previous step's clip + Adam pass runs as extra workgroups of the sampler
        # launch (include/rainbow_hip.h RB_LEARNER_DEFER_UPDATE) — ONE launch carries both, there is no clip_adam launch
        t["sample"]["work"] += t.pop("clip_adam")["work"]
        t["sample"]["hosts"] = "clip_adam"
    return t


def step_work(cfg, n_params):
    """Per-step algorithmic work: FLOPs = B (5 F - F_conv1), bytes = 4P (2 weight reads + grad write + 7 Adam)
    + frame windows."""
    B, H, A = cfg["batch_size"], cfg["hidden_size"], cfg["actions"]
    conv = CONV_GEOM[cfg["architecture"]]
    fl = [2 * cout * oh * oh * cin * ks * ks for (cin, cout, ks, _s, _ih, oh) in conv]
    F = conv[-1][1] * conv[-1][5] ** 2
    fc = 2 * (2 * H * F + (51 + A * 51) * H)
    per_sample = sum(fl) + fc
    flops = B * (5 * per_sample - fl[0])
    nbytes = 4 * n_params * 10 + B * (4 + cfg["multi_step"]) * 7056
    return flops, nbytes


# rocprofv3's own mean duration of the launch behind a profiling tag, from the committed kernel-stats summary of the same
# config (profiles/round*_final_cfg<N>_kernel_stats.csv, tools/gpu_r6_evidence.sh): a bracket can read SHORTER than the launch
# takes back to back in the step (the event in front of it absorbs part of what the launch pays in situ: k_conv_dw_all at
# batch 256 read 38 us between two events and 64 us in the kernel trace), so every roofline object carries both and prices
# `frac` with the larger.  Tags whose kernel name is shared with another launch of the step (k_nl_fwd3: hidden and output layer)
# have no entry.
CONFIG_TAG = {"pong-canonical-b32": "cfg2", "breakout-canonical-b256": "cfg3", "data-efficient-b32": "cfg4"}
ROCPROF_NAME = {
    "sample": r"k_sample<", "clip_adam": r"k_clip_adam<|k_adam_pending", "head": r"k_head<",
    "conv1_fwd": r"k_conv_fwd\w*<ConvGeom<(8, 4, 84, 20|5, 5, 84, 16)>", "conv2_fwd": r"k_conv_fwd\w*<ConvGeom<(4, 2, 20, 9|5, 5, 16, 3)>",
    "conv3_fwd": r"k_conv_fwd\w*<ConvGeom<3, 1, 9, 7>", "conv2_dx": r"k_conv_dx_lds<ConvGeom<(4, 2, 20, 9|5, 5, 16, 3)>",
    "conv3_dx": r"k_conv_dx_lds<ConvGeom<3, 1, 9, 7>", "conv_dw_all": r"k_conv_dw_all<",
    "fc_h_fwd": r"k_fc_gemm_fwd", "fc_h_bwd": r"k_fc_gemm_bwd|k_nl_bwd<false>", "fc_z_bwd": r"k_nl_bwd<true>",
}


def rocprof_means(config, hosted_update=True):
    """{tag: (mean us, file)} from the newest committed kernel-stats summary of `config`; {} when there is none."""
    import csv
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_final_%s_kernel_stats.csv" % CONFIG_TAG.get(config, "none"))))
    if not files:
        return {}
    rows = list(csv.DictReader(open(files[-1])))
    out = {}
    names = dict(ROCPROF_NAME)
    if CONFIGS[config]["batch_size"] >= 128:    # the hidden layer on the tiled GEMMs: k_nl_bwd is the output layer's backward there
        names["fc_h_bwd"], names["fc_z_bwd"] = r"k_fc_gemm_bwd", r"k_nl_bwd<"
    if hosted_update:                           # the step's sampler launch hosts the optimiser pass: the 1024-thread variant (the
        names["sample"] = r"k_sample<1024"      # PER-only phase of the same run draws with the 256-thread one)
    for tag, pat in names.items():
        hit = [r for r in rows if re.search(pat, r["Name"])]
        if len(hit) == 1:                       # (two variants of a kernel in one run: ambiguous, no entry)
            out[tag] = (float(hit[0]["AverageNs"]) / 1e3, os.path.basename(files[-1]))
    return out


def achieved(k, seconds):
    if k["bound"] == "hbm":
        return k["work"] / seconds / 1e9, HBM_PEAK_GBS
    return k["work"] / seconds / 1e12, F32_MFMA_PEAK_TF


def respawn_under_torchrun(opt):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script (one per GPU) under
    torch.distributed.run on this node and hand the terminal over to it."""
    import socket
    n_dev = torch.cuda.device_count()
    if n_dev < (1 if opt.backend == "gloo" else opt.gpus):
        sys.exit("bench.py: --gpus %d requested but only %d GPU(s) are visible" % (opt.gpus, n_dev))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(opt.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.execv(sys.executable, cmd)


def time_cpu_baseline(cfg, seconds=20.0):
    """CPU port (oracle) of the same step on this host's cores: bounded sample.  The intra-op thread count is the
    best of a short probe over {8, 16, 32, all}: on a 128+-core host the all-cores setting is several times SLOWER
    for these small ops (thread fan-out dominates), and the baseline should be the CPU's best."""
    from oracle import learner_oracle as O
    from oracle.replay_oracle import ReplayOracle
    all_threads = torch.get_num_threads()
    B, A = cfg["batch_size"], cfg["actions"]
    cap = 32768
    ocfg = O.Config(batch=B, atoms=51, actions=A, history=4, hidden=cfg["hidden_size"],
                    architecture=cfg["architecture"], multi_step=cfg["multi_step"])
    mem = ReplayOracle(cap, history=4, discount=0.99, multi_step=cfg["multi_step"])
    rs = np.random.RandomState(0)
    n = cap + cap // 2
    term = rs.random_sample(n) < 1e-3
    ts = np.zeros(n, dtype=np.int32)
    t = 0
    for i in range(n):
        ts[i] = t
        t = 0 if term[i] else t + 1
    for lo in range(0, n, 8192):
        hi = min(n, lo + 8192)
        mem.transitions.bulk_append(ts[lo:hi], rs.randint(0, 256, size=(hi - lo, 84, 84)).astype(np.uint8),
                                    rs.randint(0, A, hi - lo), rs.choice([-1.0, 0.0, 1.0], size=hi - lo, p=[0.05, 0.9, 0.05]).astype(np.float32),
                                    ~term[lo:hi])
    ts_ = mem.transitions
    ts_.set_leaves(np.arange(cap) + ts_.tree_start, np.power(np.abs(np.random.RandomState(1).randn(cap)).astype(np.float32) + 1e-3, 0.5))
    online = O.init_params(ocfg, 0)
    target = {k: v.copy() for k, v in online.items()}
    adam = O.AdamOracle(online, 6.25e-5, 1.5e-4)
    draws = O.noise_draw_count(ocfg)

    def step():
        nonlocal online
        noise_on = O.make_noise(ocfg, rs.randn(draws).astype(np.float32))
        batch = mem.sample_with_uniforms(B, rs.random_sample((64, B)))
        out = O.learn(ocfg, online, target, noise_on, O.make_noise(ocfg, rs.randn(draws).astype(np.float32)), batch)
        _, clipped = O.clip_grads(out["grads"], 10.0)
        online = adam.step(clipped)
        mem.update_priorities(batch["tree_idxs"], out["loss"])

    def run_for(sec):
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < sec:
            step()
            k += 1
        return k, time.perf_counter() - t0

    step()
    best_threads, best_rate = all_threads, 0.0
    for th in sorted({t for t in (8, 16, 32, all_threads) if t <= all_threads}):
        torch.set_num_threads(th)
        step()
        k, dt = run_for(2.0)
        if k / dt > best_rate:
            best_threads, best_rate = th, k / dt
    torch.set_num_threads(best_threads)
    k, dt = run_for(seconds)
    torch.set_num_threads(all_threads)
    return dict(value=k / dt, unit="gradient-steps/s", cores=best_threads, kind="port",
                sample="%d steps of the same config (batch %d) on a %d-capacity numpy replay, %.1f s, best of a "
                       "{8,16,32,%d}-thread probe" % (k, B, cap, dt, all_threads))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="pong-canonical-b32", choices=sorted(CONFIGS))
    ap.add_argument("--roofline-kernel", default="auto",
                    help="profiling tag of the kernel the roofline object reports (default: the dominant one, measured)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true",
                    help="no HIP-event brackets anywhere (no roofline objects): the run a rocprofv3 kernel trace should see — an "
                         "event record is a system-scope barrier packet and shows as ~6 us of idle GPU on both sides of the "
                         "bracketed launch (tools/gpu_trace_gaps.sh)")
    ap.add_argument("--capacity", type=int, default=0, help="override replay capacity (debug)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="debug: gloo = REHEARSAL of the multi-rank control flow on ONE GPU (every rank on cuda:0, RCCL refuses a second "
                         "rank per device): same respawn, barriers, MAX-reduced elapsed and value arithmetic; the replica exchange "
                         "stages its blocks through the host.  Its number says nothing about scaling")
    ap.add_argument("--extra-tags", default="", help="comma-separated extra profiling tags to time (bytes unknown: time only)")
    opt = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(opt)          # does not return
    if world != opt.gpus and not (world == 1 and os.environ.get("RAINBOW_AMD_FORCE_DIST") == "1"):
        sys.exit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (opt.gpus, world))
    import __graft_entry__
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        __graft_entry__.build()
    force_dist = os.environ.get("RAINBOW_AMD_FORCE_DIST") == "1" and "RANK" in os.environ   # one-rank RCCL plumbing test
    rehearsal = opt.backend == "gloo"
    dev_index = 0 if rehearsal else local_rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.device_count() <= dev_index:
            sys.exit("bench.py: rank %d needs GPU %d but only %d are visible" % (rank, dev_index, torch.cuda.device_count()))
        torch.cuda.set_device(dev_index)
        if rehearsal:
            torch.distributed.init_process_group("gloo")
        else:
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        torch.distributed.barrier()
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)

    from rainbow_amd import _lib as L
    from rainbow_amd import dist as rdist
    from rainbow_amd.agent import Agent
    from rainbow_amd.memory import ReplayMemory

    cfg = dict(CONFIGS[opt.config])
    if opt.capacity:
        cfg["capacity"] = opt.capacity
    args = make_args(cfg, dev)
    env = types.SimpleNamespace(action_space=lambda: cfg["actions"])
    np.random.seed(123 + rank)
    torch.manual_seed(123)
    agent = Agent(args, env)
    mem = ReplayMemory(args, cfg["capacity"], seed=1000 + rank)
    fill_replay(mem, cfg["capacity"], cfg["actions"], seed=rank)
    lib = L.load()

    def step():
        agent.reset_noise()      # main.py:151
        agent.learn(mem)         # main.py:164

    def bracketed(tag, n):
        """n steps with every launch tagged `tag` bracketed by two HIP events on its own stream -> mean seconds / launch."""
        lib.rb_profile_select(tag.encode())
        for _ in range(n):
            step()
        drain()
        ms, cnt = C.c_double(0), C.c_int64(0)
        lib.rb_profile_read(C.byref(ms), C.byref(cnt))
        lib.rb_profile_select(None)
        return (ms.value / cnt.value * 1e-3, cnt.value) if cnt.value > 0 else (None, 0)

    def drain():
        """End of a timed region: the optimiser pass the last learn() left pending (it would ride in the NEXT step's sampler
        launch) runs now, inside the region — K steps = K optimiser passes — then the device is drained."""
        agent.flush()
        torch.cuda.synchronize(dev)

    for _ in range(opt.warmup):
        step()
    drain()

    ktab = kernel_table(cfg, int(agent.params.numel()), hosted_update=agent._defer_update,
                        implicit_sigma=getattr(agent, "_implicit_sigma", False))
    # the kernel the roofline object reports = the step's DOMINANT kernel by time, found by a short bracketed pass over
    # every candidate before the timed region (or forced with --roofline-kernel)
    if opt.no_profile:
        kname = None
    elif opt.roofline_kernel in ktab:
        kname = opt.roofline_kernel
    else:
        probe = {k: bracketed(k, 30)[0] for k in ktab}
        kname = max((k for k in probe if probe[k] is not None), key=lambda k: probe[k])
    lib.rb_profile_select(kname.encode() if kname else None)
    # 1 launch in 8 is bracketed inside a long timed region; a short one keeps three bracketed launches (a 20-step run: launches
    # 0, 7, 14).  An event pair idles the stream for ~13 us (round 6: five pairs in 20 steps read 161.75 us per step against 158.37
    # without them), the launch's duration varies by < 1 us from launch to launch, and `frac` is priced with the larger of this
    # mean and rocprofv3's own over 300 steps
    stride = max(1, min(PROFILE_STRIDE, (opt.steps + 2) // 3))  # >= 3 bracketed launches in any run of >= 3 steps
    lib.rb_profile_stride(stride)
    if world > 1 or force_dist:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        step()
    drain()
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    tot_ms, launches = C.c_double(0), C.c_int64(0)
    lib.rb_profile_read(C.byref(tot_ms), C.byref(launches))
    lib.rb_profile_select(None)
    lib.rb_profile_stride(1)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # the same number of steps WITHOUT the event pair around the dominant kernel (what the bracketing itself costs)
    torch.cuda.synchronize(dev)
    t_plain = time.perf_counter()
    for _ in range(min(opt.steps, 500)):
        step()
    drain()
    plain_ms = (time.perf_counter() - t_plain) / min(opt.steps, 500) * 1e3
    # every other candidate, each timed the same way in a short pass of its own AFTER the timed region
    others = {}
    for other in ktab:
        if other != kname and not opt.no_profile:
            t, n = bracketed(other, min(200, opt.steps))
            if t is not None:
                others[other] = (t, n)
    extra = {}
    for tag in [t for t in opt.extra_tags.split(",") if t]:
        t, n = bracketed(tag, min(200, opt.steps))
        if t is not None:
            extra[tag] = t * 1e6
    ev_ms = C.c_double(0)
    lib.rb_profile_overhead(torch.cuda.current_stream(dev).cuda_stream, 256, C.byref(ev_ms))
    ev_us = ev_ms.value * 1e3          # what an EMPTY event pair reads: reported, not subtracted (rocprof's duration of
                                       # the same kernel sits between avg_us and avg_us - event_pair_overhead_us)
    hdr = mem._header()
    assert hdr.last_status == 0 and mem.failed_samples() == 0, "device sampler failed"
    assert bool(torch.isfinite(agent._loss).all()), "non-finite loss"

    # PER-only throughput (sample + priority update, no learner), same replay
    per_iters = 500
    B = cfg["batch_size"]
    fake_loss = torch.rand(B, device=dev) + 0.1
    # (donate=True: this loop never modifies its loss tensor between the call and the next draw, so the write-back reads it in place;
    # the DEFAULT copies caller-owned device operands at the call, like the reference's immediate update — both are timed and
    # reported, `per_samples_per_s` is the default, tools/per_bench.py)
    def per_loop(donate):
        for _ in range(20):
            o = mem.sample_device(B)
            mem.update_priorities(o["tree_idxs"], fake_loss, donate=donate)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        for _ in range(per_iters):
            o = mem.sample_device(B)
            mem.update_priorities(o["tree_idxs"], fake_loss, donate=donate)
        mem.flush()                  # update_priorities is lazy (it rides in the next sampler launch): the last one runs inside the timed region
        torch.cuda.synchronize(dev)
        return per_iters * B / (time.perf_counter() - t1)
    per_rate = per_loop(False)
    per_rate_donate = per_loop(True)
    assert mem.expired_waits() == 0, "a cross-stream wait of the early draw expired during the run"

    if rank == 0:
        ms_per_step = elapsed / opt.steps * 1e3
        par = "single device" if world == 1 else "replicas x%d, exchange=%s" % (world, rdist.mode())
        if world > 1 and rehearsal:
            par += " [REHEARSAL: %d ranks on ONE GPU over gloo — control flow only, not a scaling number]" % world
        out = {
            "metric": "gradient-steps/sec (batch=%d, atoms=51)" % B, "value": world * opt.steps / elapsed,
            "unit": "gradient-steps/s", "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": opt.config, "architecture": cfg["architecture"], "batch_per_gpu": B,
                       "global_batch": B * world, "atoms": 51, "actions": cfg["actions"], "multi_step": cfg["multi_step"],
                       "replay_capacity_per_gpu": cfg["capacity"], "parallelism": par},
            "per_samples_per_s": per_rate * world,
            "early_draw": "spec_draw=1" in os.environ.get("RB_OPTS", ""),     # (opt-in: only an append-free loop like this one arms it)
            "ms_per_step_unbracketed": plain_ms,
            "library_source_hash": L.source_hash(lib),
        }
        assert out["library_source_hash"] == __graft_entry__.source_hash(), "the loaded library was not built from this tree"
        # the second BASELINE metric (memory.py:148-159: sample(B) + update_priorities, no learner), priced against HBM with
        # SURVEY 8(d)'s algorithmic bytes — (h + n) frames read + 2h frames written per sample — and against what actually
        # bounds it at one batch in flight: a chain of DEPENDENT round trips (sampler: tree top in LDS, 2 trips below it,
        # 1 trip for the window's timesteps/rewards, then the frame gather; update: 9 hashed levels + the dense top)
        per_bytes = (4 + cfg["multi_step"] + 2 * 4) * 7056          # history 4 in every BASELINE config
        out["roofline_per"] = {"bound": "hbm", "achieved": per_rate * per_bytes / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": per_rate * per_bytes / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_sample": per_bytes,
                               "batch": B, "us_per_batch": 1e6 * B / per_rate, "donate": False,
                               "samples_per_s_with_donate": per_rate_donate * world,
                               "binding": "latency: one batch in flight = 2 dependent launches (update of batch k + draw of batch k + 1 "
                                          "as one single-workgroup chain, then the frame gather); up to 64 leaves per write-back, "
                                          "else 3 launches; bytes are 0.1% of what HBM moves in that time"}
        # counter traffic: only from a PMC pass of THIS config that is committed under profiles/ (tools/gpu_r6_evidence.sh);
        # null otherwise — never a number measured on another workload
        import glob
        pmc_files = sorted(glob.glob(os.path.join(ROOT, "profiles", "round*_pmc_%s.json" % opt.config)))   # newest round last
        pmc = json.load(open(pmc_files[-1])) if pmc_files else {}

        rp = rocprof_means(opt.config, agent._defer_update)

        def roof(name, seconds, n):
            k = ktab[name]
            bracket = seconds
            rp_us = rp.get(name, (None, None))[0]
            if rp_us is not None and rp_us * 1e-6 > seconds:
                seconds = rp_us * 1e-6          # `achieved` / `frac` are priced with the LARGER of the bracket and rocprof's mean
            ach, peak = achieved(k, seconds)
            # *_net: the same with what an EMPTY event pair reads taken off the bracket — the figure rocprofv3's own
            # duration of the kernel agrees with (profiles/); `achieved` / `frac` stay the raw, conservative bracket
            net = max(seconds - ev_us * 1e-6, 1e-9)
            return {"kernel": name, "bound": k["bound"], "achieved": ach, "peak": peak, "unit": k["unit"], "frac": ach / peak,
                    "traffic": pmc.get(name, {}).get("hbm_bytes_per_launch"), "avg_us": bracket * 1e6, "launches": n,
                    "rocprof_avg_us": rp_us, "rocprof_source": rp.get(name, (None, None))[1],
                    "frac_priced_with": "rocprof_avg_us" if seconds != bracket else "avg_us",
                    "algorithmic_work_per_launch": k["work"], "avg_us_net": net * 1e6, "frac_net": achieved(k, net)[0] / peak,
                    **({"hosts": k["hosts"]} if "hosts" in k else {})}

        if launches.value > 0:
            out["roofline"] = roof(kname, tot_ms.value / launches.value * 1e-3, launches.value)
            out["roofline"]["event_pair_overhead_us"] = ev_us
            out["roofline"]["traffic_source"] = os.path.basename(pmc_files[-1]) if pmc_files else None
            out["roofline"]["bracketed"] = "every %s launch of the timed region" % ({1: "single", 2: "2nd", 3: "3rd"}.get(stride, "%dth" % stride))
            out["roofline"]["selected"] = "forced" if opt.roofline_kernel in ktab else "largest mean launch time of a 30-step probe"
            out["roofline_others"] = [roof(o, t, n) for o, (t, n) in sorted(others.items(), key=lambda kv: -kv[1][0])]
        flops, nbytes = step_work(cfg, int(agent.params.numel()))
        t_hbm, t_mfma = nbytes / (HBM_PEAK_GBS * 1e9), flops / (F32_MFMA_PEAK_TF * 1e12)
        sec = elapsed / opt.steps
        out["roofline_step"] = ({"bound": "hbm", "achieved": nbytes / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": t_hbm / sec} if t_hbm >= t_mfma else
                                {"bound": "mfma", "achieved": flops / sec / 1e12, "peak": F32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                                 "frac": t_mfma / sec})
        out["roofline_step"].update(algorithmic_flops=flops, algorithmic_bytes=nbytes)
        if extra:
            out["extra_kernel_us"] = extra
        if world == 1 and not opt.no_cpu_baseline:
            out["cpu_baseline"] = time_cpu_baseline(cfg)
        print(json.dumps(out))
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
