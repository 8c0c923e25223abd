#!/usr/bin/env python
"""bench.py — Rainbow learn-step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one complete gradient step of the hot path on synthetic transitions already
resident in HBM:  online noise resample -> Agent.learn(mem) = PER sample (sum-tree search +
frame-stack gather) -> 3 forwards -> C51 projection / loss -> backward -> [RCCL all-reduce]
-> global-norm clip -> Adam -> priority update.          (reference: main.py:151,164; agent.py:61-100)

Workload (SURVEY §8d / BASELINE.md §3): config 2 — canonical network, batch 32, 51 atoms,
6 actions, n=3, 1M-capacity replay filled to capacity with the write head mid-buffer,
non-uniform priorities.  Multi-GPU = independent replicas (own replay, own noise) with one
gradient all-reduce per step; weak scaling (per-GPU work fixed).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     — dominant kernel (clip+Adam pass), timed live with HIP events on its own stream (rb_profile_*);
                 roofline_others: the hidden-layer forward / backward streams, timed the same way
  cpu_baseline — the CPU oracle (a port of the reference's algorithm: numpy replay + torch-CPU
                 learner) timed on this box's host cores on a bounded sample (rank 0, N=1 only)
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


def kernel_table(cfg, n_params):
    """Algorithmic work per launch of the candidate dominant kernels."""
    B, A = cfg["batch_size"], cfg["actions"]
    H = cfg["hidden_size"]
    F = 3136 if cfg["architecture"] == "canonical" else 576
    wh = 2 * H * F * 4                      # one of mu / sigma of the fused hidden layer, bytes
    return {
        # clip + Adam over the flat buffers: reads p, g, m, v and writes p, m, v once (the step's largest kernel)
        "clip_adam": dict(bound="hbm", work=7 * 4 * n_params, unit="GB/s"),
        # hidden layer forward: streams mu+sigma of BOTH nets once; activations are L2-resident
        "fc_h_fwd": dict(bound="hbm", work=2 * 2 * wh + 3 * B * F * 4 + 3 * B * 2 * H * 4 * 2, unit="GB/s"),
        # hidden layer weight grads: writes d_mu + d_sigma once
        # hidden layer backward (one launch): streams mu+sigma of the online net once (dX), writes d_mu + d_sigma once (dW)
        "fc_h_bwd": dict(bound="hbm", work=2 * wh + 2 * wh + B * (F + 2 * H) * 4, unit="GB/s"),
    }


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
    ap.add_argument("--roofline-kernel", default="clip_adam")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--capacity", type=int, default=0, help="override replay capacity (debug)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the learn step as one captured hipGraph (no per-kernel HIP-event timing then; measured "
                         "within 2%% of eager on MI355X, profiles/round1_launch_ab.txt)")
    opt = ap.parse_args()

    os.environ["RAINBOW_AMD_GRAPH"] = "1" if opt.graph else "0"
    import __graft_entry__
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if rank == 0:
        __graft_entry__.build()
    force_dist = os.environ.get("RAINBOW_AMD_FORCE_DIST") == "1" and "RANK" in os.environ   # one-rank RCCL plumbing test
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        torch.distributed.barrier()
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from rainbow_amd import _lib as L
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

    for _ in range(opt.warmup):
        step()
    torch.cuda.synchronize(dev)

    ktab = kernel_table(cfg, int(agent.params.numel()))
    kname = opt.roofline_kernel if opt.roofline_kernel in ktab else "clip_adam"
    lib.rb_profile_select(kname.encode())
    if world > 1 or force_dist:
        torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(opt.steps):
        step()
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()
    elapsed = time.perf_counter() - t0
    tot_ms, launches = C.c_double(0), C.c_int64(0)
    lib.rb_profile_read(C.byref(tot_ms), C.byref(launches))
    lib.rb_profile_select(None)
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # the other GB-scale kernels, each timed the same way in a short pass of its own AFTER the timed region
    others = {}
    for other in ktab:
        if other == kname:
            continue
        lib.rb_profile_select(other.encode())
        for _ in range(min(200, opt.steps)):
            step()
        torch.cuda.synchronize(dev)
        o_ms, o_n = C.c_double(0), C.c_int64(0)
        lib.rb_profile_read(C.byref(o_ms), C.byref(o_n))
        lib.rb_profile_select(None)
        if o_n.value > 0:
            others[other] = (o_ms.value / o_n.value * 1e-3, o_n.value)
    ev_ms = C.c_double(0)
    lib.rb_profile_overhead(torch.cuda.current_stream(dev).cuda_stream, 256, C.byref(ev_ms))
    ev_us = ev_ms.value * 1e3          # what an EMPTY event pair reads: reported, not subtracted (rocprof's duration of
                                       # the same kernel sits between avg_us and avg_us - event_pair_overhead_us)
    hdr = mem._header()
    assert hdr.last_status == 0, "device sampler failed"
    assert bool(torch.isfinite(agent._loss).all()), "non-finite loss"

    # PER-only throughput (sample + priority update, no learner), same replay
    per_iters = 500
    B = cfg["batch_size"]
    fake_loss = torch.rand(B, device=dev) + 0.1
    for _ in range(20):
        o = mem.sample_device(B)
        mem.update_priorities(o["tree_idxs"], fake_loss)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    for _ in range(per_iters):
        o = mem.sample_device(B)
        mem.update_priorities(o["tree_idxs"], fake_loss)
    torch.cuda.synchronize(dev)
    per_rate = per_iters * B / (time.perf_counter() - t1)

    if rank == 0:
        ms_per_step = elapsed / opt.steps * 1e3
        out = {
            "metric": "gradient-steps/sec (batch=%d, atoms=51)" % B, "value": world * opt.steps / elapsed,
            "unit": "gradient-steps/s", "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": opt.config, "architecture": cfg["architecture"], "batch_per_gpu": B,
                       "global_batch": B * world, "atoms": 51, "actions": cfg["actions"], "multi_step": cfg["multi_step"],
                       "replay_capacity_per_gpu": cfg["capacity"], "parallelism": "replicas x%d + grad all-reduce" % world},
            "per_samples_per_s": per_rate * world,
        }
        k = ktab[kname]
        if launches.value > 0:
            avg_s = tot_ms.value / launches.value * 1e-3     # raw event-pair time (includes the bracketing, see below)
            if k["bound"] == "hbm":
                achieved, peak = k["work"] / avg_s / 1e9, HBM_PEAK_GBS
            else:
                achieved, peak = k["work"] / avg_s / 1e12, F32_MFMA_PEAK_TF
            pmc = os.path.join(ROOT, "profiles", "round1_pmc.json")   # rocprofv3 --pmc passes (tools/gpu_pmc.sh), per launch
            pmc = json.load(open(pmc)) if os.path.exists(pmc) else {}
            out["roofline"] = {"kernel": kname, "bound": k["bound"], "achieved": achieved, "peak": peak, "unit": k["unit"],
                               "frac": achieved / peak, "traffic": pmc.get(kname, {}).get("hbm_bytes_per_launch"),
                               "avg_us": avg_s * 1e6, "launches": launches.value,
                               "algorithmic_work_per_launch": k["work"], "event_pair_overhead_us": ev_us}
            out["roofline_others"] = [
                {"kernel": o, "bound": "hbm", "achieved": ktab[o]["work"] / t / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": ktab[o]["work"] / t / 1e9 / HBM_PEAK_GBS, "traffic": pmc.get(o, {}).get("hbm_bytes_per_launch"),
                 "avg_us": t * 1e6, "launches": n, "algorithmic_work_per_launch": ktab[o]["work"]}
                for o, (t, n) in others.items()]
        if world == 1 and not opt.no_cpu_baseline:
            out["cpu_baseline"] = time_cpu_baseline(cfg)
        print(json.dumps(out))
    if world > 1 or force_dist:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
