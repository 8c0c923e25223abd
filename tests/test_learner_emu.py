"""Kernel-logic tests of the learn step on the host interpreter (CPU): the same kernel sources as
librainbow_hip.so, against the REAL reference's golden vectors (tests/golden/learn_*.npz)."""
import os

import numpy as np
import pytest

import scenarios
import ts_scenarios
from cabi_adapter import CAbiLearnAdapter, NumpyMem
from helpers import assert_learn_trace_matches, load_golden
from hipemu import loader
from oracle import learner_oracle as O


@pytest.fixture(scope="module")
def emu():
    return loader.load()


@pytest.mark.parametrize("name", ["atoms21", "k10", "dataeff", "canon"])
def test_learn_step_matches_reference_golden(emu, name):
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    trace = scenarios.learn_scenario(ad, name, O)
    assert_learn_trace_matches(trace, load_golden("learn_%s.npz" % name), label="emu/" + name)
    ad.close()


def test_generic_gemm_fallback_still_matches_golden(emu, monkeypatch):
    """RB_OPTS=generic=1 forces every contraction through gemm_core.h (the fallback used when the
    streamed noisy-linear kernels' alignment preconditions do not hold)."""
    monkeypatch.setenv("RB_OPTS", "generic=1")
    name = "atoms21"
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    trace = scenarios.learn_scenario(ad, name, O)
    assert_learn_trace_matches(trace, load_golden("learn_%s.npz" % name), label="emu-generic/" + name)
    ad.close()


def test_row_split_input_gradient_matches_golden(emu, monkeypatch):
    """The hidden layer's input gradient split over 4 row ranges + k_dfeat_finish(splits = 4) — the shape the canonical
    hidden-512 network runs (xs = 2H/256 = 4, learner.hip) — forced on the small canonical fixture with RB_OPTS=xs=4."""
    monkeypatch.setenv("RB_OPTS", "xs=4")
    name = "canon"
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    trace = scenarios.learn_scenario(ad, name, O, steps=1)          # the first golden step is enough to pin the split path
    golden = load_golden("learn_%s.npz" % name)
    assert_learn_trace_matches(trace, {k: v for k, v in golden.items() if k in trace}, label="emu-xs4/" + name)
    assert any("_grad/fc_h" in k for k in trace) and any("_grad/convs" in k for k in trace)
    ad.close()


def test_chunk_fastest_conv_block_order_matches_golden(emu, monkeypatch):
    """RB_OPTS=img_fast=0: the (chunk, tile, image) block order of the conv launches — the fallback when the image count is
    not a multiple of 8; the default image-fastest order is what every other test here runs (3B = 24 images on this fixture)."""
    monkeypatch.setenv("RB_OPTS", "img_fast=0")
    name = "canon"
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    trace = scenarios.learn_scenario(ad, name, O, steps=1)
    golden = load_golden("learn_%s.npz" % name)
    assert_learn_trace_matches(trace, {k: v for k, v in golden.items() if k in trace}, label="emu-chunkfast/" + name)
    assert any("_grad/convs" in k for k in trace)
    ad.close()


@pytest.mark.parametrize("t16", ["0", "6"], ids=["split-k-32x32", "whole-k-16x16-later-layers-only"])
def test_conv_forward_tile_variants_match_golden(emu, monkeypatch, t16):
    """The conv forward's MFMA phase exists twice (conv_lds.h rb_conv_fwd_body): 32x32x2 tiles with the reduction split over 8
    waves + an LDS sum, and (T16) one wave per 16x16 tile over the whole reduction with the epilogue straight from the
    accumulators.  Default: T16 for every layer incl. the u8 first one (RB_OPTS t16=7, what every other test runs); here all layers
    on the split-K body (t16=0) and the first layer alone on it (t16=6)."""
    monkeypatch.setenv("RB_OPTS", "t16=" + t16)
    name = "canon"
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    trace = scenarios.learn_scenario(ad, name, O, steps=1)
    golden = load_golden("learn_%s.npz" % name)
    assert_learn_trace_matches(trace, {k: v for k, v in golden.items() if k in trace}, label="emu-t16=%s/%s" % (t16, name))
    assert any("_grad/convs" in k for k in trace)
    ad.close()


@pytest.mark.parametrize("name,steps,full,t16", [("dataeff", None, "1", 1), ("canon", 1, "1", 1), ("canon", 1, "0", 1), ("canon", 1, "1", 0)])
def test_multi_image_conv_kernels_match_golden(emu, monkeypatch, name, steps, full, t16):
    """Large batches run the conv forward and data-gradient kernels with one weight slab per workgroup and a loop over
    images (k_conv_fwd_multi_t16 / k_conv_dx_t16_multi — whole-K 16x16x4 tiles, round 6 — or, t16 = 0, the split-K k_conv_fwd_multi /
    k_conv_dx_lds<..., MULTI>); RB_OPTS conv_multi / dx_ipb force those paths (ragged: neither divides the batch; the group of 5
    that holds the last online and the first target image re-stages its slab) on the small fixtures, with the last layer's dY
    formed from the row-split partials in the loop."""
    # dx_ipb / conv_multi: ragged image groups in the input-gradient and forward kernels; conv_full: the first layer's
    # whole-image kernel (k_conv_fwd_full) or, 0, the one-image kernel
    # (t16 = 0: the split-K bodies of both the forward and the data gradient; 1: k_conv_fwd_multi_t16 / k_conv_dx_t16_multi)
    # dw_ipb<layer>: images summed per workgroup of the weight-gradient launch, chosen per layer at large batches (ragged here too)
    monkeypatch.setenv("RB_OPTS", "dx_ipb=3,conv_multi=5,conv_full=%s,conv_multi_t16=%d,dx_t16=%d,dw_ipb0=3,dw_ipb1=2,dw_ipb2=5" % (full, t16, t16))
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    trace = scenarios.learn_scenario(ad, name, O, steps=steps)
    golden = load_golden("learn_%s.npz" % name)
    assert_learn_trace_matches(trace, {k: v for k, v in golden.items() if k in trace}, label="emu-dx-multi/" + name)
    assert any("_grad/convs.0" in k for k in trace)
    ad.close()


@pytest.mark.parametrize("name,steps", [("canon", 1), ("dataeff", None), ("atoms21", None)])
def test_tiled_gemm_hidden_layer_matches_golden(emu, monkeypatch, name, steps):
    """From 128 rows per net on (BASELINE config 3: batch 256) the hidden NoisyLinear layer runs as LDS-tiled f32 MFMA GEMMs
    (fc_gemm.h): forward with split-K + last-arriver sum, input gradient into the row-split partial slices, weight gradient
    with the sigma / bias / sum-of-squares epilogue.  RB_OPTS fc_gemm=1 forces them onto the small fixtures: ragged tiles in
    every dimension (M = 16 / 8, N = 128 / 64 / 96, K = 3136 / 576), both streams inside one 128-row tile."""
    monkeypatch.setenv("RB_OPTS", "fc_gemm=1")
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    trace = scenarios.learn_scenario(ad, name, O, steps=steps)
    golden = load_golden("learn_%s.npz" % name)
    assert_learn_trace_matches(trace, {k: v for k, v in golden.items() if k in trace}, label="emu-fc-gemm/" + name)
    assert any("_grad/fc_h" in k for k in trace)
    ad.close()


@pytest.mark.parametrize("name", ["dataeff"])     # (the canonical stack runs the same test on the GPU: 2.5 min on the interpreter)
def test_fused_weight_gradient_in_optimiser_pass_matches_golden(emu, name):
    """RB_LEARNER_FUSE_FC_H_DW (what rainbow_amd.agent.Agent runs): the hidden layer's weight gradient is not stored by the
    backward; the clip + Adam pass recomputes each tile while it streams the parameters.  With WRITE_FUSED_GRADS the pass
    also stores what it computed, so the whole golden trace (all 22 gradients, norms, post-Adam parameters over three
    steps) is checked on the product path; without it the parameters must be bit-identical to that run."""
    from rainbow_amd import _lib as L
    traces = []
    for flags in (L.LEARNER_FUSE_FC_H_DW | L.LEARNER_WRITE_FUSED_GRADS, L.LEARNER_FUSE_FC_H_DW):
        ad = CAbiLearnAdapter(emu, NumpyMem(), name)
        ad.learner_flags = flags
        traces.append(scenarios.learn_scenario(ad, name, O))
        ad.close()
    assert_learn_trace_matches(traces[0], load_golden("learn_%s.npz" % name), label="emu-fused-dw/" + name)
    for k in traces[0]:
        if "_param/" in k or k.endswith("_loss") or k.endswith("grad_norm") or k.startswith(("act_", "q_")):
            assert np.array_equal(np.asarray(traces[0][k]), np.asarray(traces[1][k])), k


def test_device_step_counter_equals_host_step(emu):
    """rb_learner_clip_adam with step = 0 (step number read from the device counter the learn call increments, bias
    corrections formed in the kernel — the hipGraph path) against the by-value step: bit-identical parameters."""
    import ctypes as C
    from rainbow_amd import _lib as L
    name = "atoms21"
    runs = []
    for device_step in (False, True):
        ad = CAbiLearnAdapter(emu, NumpyMem(), name)
        if device_step:
            ctr = np.zeros(1, dtype=np.int64)
            L.check(emu, emu.rb_learner_set_step_counter(ad.h, ctr.ctypes.data))
            ad.step_from_device = True
        runs.append(scenarios.learn_scenario(ad, name, O))
        if device_step:
            assert int(ctr[0]) == scenarios.LEARN_STEPS
        ad.close()
    for k in runs[0]:
        assert np.array_equal(np.asarray(runs[0][k]), np.asarray(runs[1][k])), k


def test_unit_conversion_is_exact():
    """rb_unit's multiply + Newton step equals the correctly rounded x/255 for every byte (memory.py:137)."""
    x = np.arange(256, dtype=np.float32)
    inv = np.float32(1.0 / 255.0)
    q = (x * inv).astype(np.float32)
    r = (np.float64(-255.0) * q.astype(np.float64) + x.astype(np.float64)).astype(np.float32)
    q2 = (r.astype(np.float64) * np.float64(inv) + q.astype(np.float64)).astype(np.float32)
    assert np.array_equal(q2, x / np.float32(255))


@pytest.mark.parametrize("fc_gemm", ["-1", "1"], ids=["streamed-fc", "tiled-gemm-fc"])
def test_zero_copy_windows_equals_gathered_stacks(emu, monkeypatch, fc_gemm):
    """rb_learner_learn_windows (conv1 reads the replay ring through the sampler's window table) must give
    bit-identical loss and gradients to rb_learner_learn on the gathered stacks, incl. blanked frames.  Second half: the
    priority write-back as a tenant of the hidden layer's backward launch — k_nl_bwd (256 threads) and, with RB_OPTS
    fc_gemm=1, k_fc_gemm_bwd (512 threads, block 0 of a launch whose tiles start at block 8)."""
    import ctypes as C
    monkeypatch.setenv("RB_OPTS", "fc_gemm=" + fc_gemm)
    from cabi_adapter import CAbiReplayAdapter
    from rainbow_amd import _lib as L
    name = "dataeff"
    c = scenarios.LEARN_CONFIGS[name]
    B, h, n = c["batch"], c["history"], c["multi_step"]
    mem = NumpyMem()
    rp = CAbiReplayAdapter(emu, mem, 512, h, n, c["discount"], 0.5)
    rs = np.random.RandomState(3)
    for _ in range(600):
        rp.append(scenarios.synth_state(rs, h, 0), int(rs.randint(0, c["actions"])), float(rs.choice([-1.0, 0.0, 1.0])),
                  bool(rs.random_sample() < 0.1))
    out = rp.sample(B, rs.random_sample((32, B)), 0.5)
    assert (out["states"].reshape(B, h, -1).max(axis=2) == 0).any(), "scenario must contain a blanked frame"
    ad = CAbiLearnAdapter(emu, mem, name)
    ad.load(O.init_params(ad_cfg := O.Config(**c), 1), O.init_params(ad_cfg, 2))
    draws = O.noise_draw_count(ad_cfg)
    ad.reset_noise_online(rs.randn(draws).astype(np.float32))
    raw_tg = mem.upload(rs.randn(draws).astype(np.float32))
    L.check(emu, emu.rb_learner_reset_noise(ad.h, 1, mem.ptr(raw_tg), None))
    bufs = {k: mem.upload(v) for k, v in dict(states=out["states"], next_states=out["next_states"], actions=out["actions"],
                                               returns=out["returns"], nonterminals=out["nonterminals"].reshape(B),
                                               weights=out["weights"]).items()}
    loss_a, loss_b = mem.empty((B,), np.float32), mem.empty((B,), np.float32)
    L.check(emu, emu.rb_learner_learn(ad.h, mem.ptr(bufs["states"]), mem.ptr(bufs["next_states"]), mem.ptr(bufs["actions"]),
                                      mem.ptr(bufs["returns"]), mem.ptr(bufs["nonterminals"]), mem.ptr(bufs["weights"]),
                                      mem.ptr(loss_a), None))
    grads_a = ad.grads.copy()
    ad.grads[:] = 0
    L.check(emu, emu.rb_learner_learn_windows(ad.h, rp.bufs.frames_dev, rp.bufs.window_dev, rp.bufs.window_len,
                                              mem.ptr(bufs["actions"]), mem.ptr(bufs["returns"]), mem.ptr(bufs["nonterminals"]),
                                              mem.ptr(bufs["weights"]), mem.ptr(loss_b), None))
    assert np.array_equal(loss_a, loss_b)
    assert np.array_equal(grads_a, ad.grads)
    # fused priority write-back: with a sink set, the same launch also updates the sum-tree — bit-identical to the
    # separate rb_replay_update_priorities call on a twin tree
    tree_before = rp.tree()
    idx_dev = mem.upload(out["tree_idxs"])
    L.check(emu, emu.rb_learner_set_priority_sink(ad.h, rp.h, mem.ptr(idx_dev)))
    L.check(emu, emu.rb_learner_learn_windows(ad.h, rp.bufs.frames_dev, rp.bufs.window_dev, rp.bufs.window_len,
                                              mem.ptr(bufs["actions"]), mem.ptr(bufs["returns"]), mem.ptr(bufs["nonterminals"]),
                                              mem.ptr(bufs["weights"]), mem.ptr(loss_b), None))
    assert emu.rb_learner_priority_written(ad.h) == 1
    tree_fused, hdr_fused = rp.tree(), rp.raw_header()
    assert not np.array_equal(tree_fused, tree_before)
    # rewind the tree and apply the same update through the stand-alone entry point
    leaves = np.arange(rp.bufs.tree_start, rp.bufs.tree_len)
    rp.update_leaves(leaves[:1024][:512], tree_before[rp.bufs.tree_start:][:512])
    assert np.array_equal(rp.tree(), tree_before)
    rp.update_priorities(out["tree_idxs"], loss_b)
    assert np.array_equal(rp.tree(), tree_fused)
    assert rp.raw_header().total == hdr_fused.total
    L.check(emu, emu.rb_learner_set_priority_sink(ad.h, None, None))
    ad.close(); rp.close()


def test_long_history_uses_generic_convs_with_streamed_fc(emu):
    """history > 4 disables the LDS conv kernels but keeps the streamed FC kernels (row-major -> k-blocked copy in
    between); checked against the oracle directly since no golden exists for this shape."""
    import cabi_adapter
    c = dict(architecture="data-efficient", hidden=32, actions=3, atoms=21, batch=4, multi_step=2, discount=0.95,
             history=5, v_min=-5.0, v_max=5.0)
    scenarios.LEARN_CONFIGS["_hist5"] = c
    try:
        cfg = O.Config(**c)
        ad = CAbiLearnAdapter(emu, NumpyMem(), "_hist5")
        online, target = O.init_params(cfg, 41), O.init_params(cfg, 42)
        ad.load(online, target)
        rs = np.random.RandomState(9)
        draws = O.noise_draw_count(cfg)
        raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
        ad.reset_noise_online(raw_on)
        batch = scenarios.make_batch(c, 77)
        out = ad.learn_step(batch, raw_tg)
        want = O.learn(cfg, online, target, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg), batch)
        total, clipped = O.clip_grads(want["grads"], scenarios.LEARN_HYPER["norm_clip"])
        np.testing.assert_allclose(out["loss"], want["loss"], rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(out["grad_norm"], total, rtol=2e-5)
        for k, g in clipped.items():
            np.testing.assert_allclose(out["grads"][k], g, rtol=2e-4, atol=5e-6 * float(np.abs(g).max()) + 1e-12, err_msg=k)
        ad.close()
    finally:
        del scenarios.LEARN_CONFIGS["_hist5"]


@pytest.mark.parametrize("norm_clip", [0.02])   # the idle clip (10.0) is what every golden scenario already runs
def test_fused_clip_adam_equals_clip_then_torch_adam(emu, norm_clip):
    """rb_learner_clip_adam (one pass) against rb_learner_clip_grad followed by torch.optim.Adam (agent.py:97-98),
    with the clip idle (10.0) and biting (0.02: the gradient is rewritten in place, as clip_grad_norm_ leaves it)."""
    name = "atoms21"
    traces = []
    for fused in (False, True):
        ad = CAbiLearnAdapter(emu, NumpyMem(), name)
        ad.fused_adam = fused
        ad.hy = dict(ad.hy, norm_clip=norm_clip)
        traces.append(scenarios.learn_scenario(ad, name, O))
        ad.close()
    a, b = traces
    bites = False
    for k in a:
        if "_param/" in k:        # summaries of the post-step parameters: same tolerance as against the reference
            np.testing.assert_allclose(b[k], a[k], rtol=0, atol=2e-7, err_msg=k)
        elif k.startswith("s0_"):  # before the first optimiser step both runs are the same computation
            assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k
        if k.endswith("_grad_norm") and float(a[k]) > norm_clip:
            bites = True
    assert bites == (norm_clip < 1.0)


@pytest.mark.parametrize("name", ["atoms21", "dataeff"])
def test_act_path_single_equals_batched(emu, name):
    """rb_learner_act (act_path.h: one channel / one weight row per wave) against rb_learner_act_batch (the training
    kernels on n images), noisy and eval mode; the golden act_* entries pin the single path to the reference."""
    c = scenarios.LEARN_CONFIGS[name]
    cfg = O.Config(**c)
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    ad.load(O.init_params(cfg, 77), O.init_params(cfg, 78))
    ad.reset_noise_online(np.random.RandomState(5).randn(O.noise_draw_count(cfg)).astype(np.float32))
    states = [scenarios.synth_state(np.random.RandomState(100 + i), c["history"], 0) for i in range(3)]
    for noisy in (True, False):
        single = [ad.act(s, noisy) for s in states]
        acts, qs = ad.act_batch(states, noisy)
        for i, (a, q) in enumerate(single):
            assert a == acts[i]
            np.testing.assert_allclose(q, qs[i], rtol=2e-5, atol=1e-6)
    ad.close()


def test_projection_known_answers(emu, monkeypatch):
    """Analytic checks of the C51 projection that need no reference run: rows of m sum to 1; a terminal
    transition puts all mass on the two atoms around (R - Vmin)/dz with weights (u - b), (b - l); R = 0 with
    gamma^n = 1 on a non-terminal row returns the target distribution itself (agent.py:79-92)."""
    cfgd = dict(scenarios.LEARN_CONFIGS["atoms21"], discount=1.0)
    monkeypatch.setitem(scenarios.LEARN_CONFIGS, "ka", cfgd)
    cfg = O.Config(**cfgd)
    ad = CAbiLearnAdapter(emu, NumpyMem(), "ka")
    ad.load(O.init_params(cfg, 31), O.init_params(cfg, 32))
    draws = O.noise_draw_count(cfg)
    rs = np.random.RandomState(9)
    ad.reset_noise_online(rs.randn(draws).astype(np.float32))
    batch = scenarios.make_batch(cfgd, 555)
    B, Z = cfgd["batch"], cfgd["atoms"]
    vmin, vmax = cfgd["v_min"], cfgd["v_max"]
    dz = (vmax - vmin) / (Z - 1)
    batch["returns"][:] = np.array([0.0, 0.0, 1.3, -2.75, 0.4][:B], dtype=np.float32)
    batch["nonterminals"][:] = np.array([1, 1, 0, 0, 1][:B], dtype=batch["nonterminals"].dtype).reshape(batch["nonterminals"].shape)
    ad.learn_step(batch, rs.randn(draws).astype(np.float32))
    m = ad.debug(1, (B, Z), np.float32)
    pns = ad.debug(3, (B, Z), np.float32)
    np.testing.assert_allclose(m.sum(1), 1.0, rtol=0, atol=2e-6)
    np.testing.assert_allclose(pns.sum(1), 1.0, rtol=0, atol=2e-6)
    for i in (0, 1):                                   # identity rows
        np.testing.assert_allclose(m[i], pns[i], rtol=0, atol=2e-6)
    for i in (2, 3):                                   # terminal rows
        b = (float(batch["returns"][i]) - vmin) / dz
        lo, hi = int(np.floor(b)), int(np.ceil(b))
        want = np.zeros(Z, dtype=np.float64)
        want[lo] += hi - b
        want[hi] += b - lo
        np.testing.assert_allclose(m[i], want, rtol=0, atol=2e-6)
    ad.close()


def _ts_build(emu, name):
    return ts_scenarios.ts_build(emu, NumpyMem, name)


_ts_snapshot, _ts_args = ts_scenarios.ts_snapshot, ts_scenarios.ts_args


def test_train_step_entry_point_equals_its_three_calls(emu):
    """rb_learner_train_step (what Agent.learn calls: sampler + noise tenant, zero-copy learn, clip + Adam, priority
    write-back through the sink) against the same three entry points issued one by one on a twin replay / twin learner:
    three consecutive steps, device RNG for the sampler and the noise, everything bit-identical — batch, loss, parameters,
    Adam moments, noise buffers and the sum-tree."""
    import ctypes as C
    from rainbow_amd import _lib as L
    name = "dataeff"
    c = scenarios.LEARN_CONFIGS[name]
    B = c["batch"]
    hy = scenarios.LEARN_HYPER
    mem1, rp1, ad1, o1, job1 = _ts_build(emu, name)
    mem2, rp2, ad2, o2, job2 = _ts_build(emu, name)
    for step in range(1, 4):
        beta = 0.4 + 0.1 * step
        ts = _ts_args(name, mem1, rp1, ad1, o1, job1, beta, step)
        L.check(emu, emu.rb_learner_train_step(ad1.h, C.byref(ts), None))
        assert emu.rb_learner_priority_written(ad1.h) == 1
        # twin: the three entry points, one by one
        L.check(emu, emu.rb_replay_sample_fused_noise(rp2.h, B, beta, None, 64, mem2.ptr(o2["tree_idx"]), None, None,
                                                     mem2.ptr(o2["actions"]), mem2.ptr(o2["returns"]), mem2.ptr(o2["nonterm"]),
                                                     mem2.ptr(o2["weights"]), C.byref(job2), None))
        L.check(emu, emu.rb_learner_learn_windows(ad2.h, rp2.bufs.frames_dev, rp2.bufs.window_dev, rp2.bufs.window_len,
                                                  mem2.ptr(o2["actions"]), mem2.ptr(o2["returns"]), mem2.ptr(o2["nonterm"]),
                                                  mem2.ptr(o2["weights"]), mem2.ptr(o2["loss"]), None))
        L.check(emu, emu.rb_learner_clip_adam(ad2.h, hy["norm_clip"], mem2.ptr(ad2.adam_m), mem2.ptr(ad2.adam_v), hy["lr"], 0.9,
                                              0.999, hy["adam_eps"], step, mem2.ptr(o2["norm"]), None))
        a, b = _ts_snapshot(mem1, rp1, ad1, o1), _ts_snapshot(mem2, rp2, ad2, o2)
        for k in a:
            assert np.array_equal(a[k], b[k]), (step, k)
        assert rp1.raw_header().last_status == 0
    assert not np.array_equal(a["params"], ad1._flat(O.init_params(O.Config(**c), 1))), "the online net must have moved"
    ad1.close(); ad2.close(); rp1.close(); rp2.close()


@pytest.mark.parametrize("implicit_sigma,max_norm,gemm", [(False, None, False), (True, None, False), (False, 0.02, False), (True, 0.02, False),
                                                          (True, None, True)],
                         ids=["stored-sigma-grad", "implicit-sigma-grad", "stored-sigma-grad-clip-bites", "implicit-sigma-grad-clip-bites",
                              "implicit-sigma-grad-tiled-gemm"])
def test_deferred_optimiser_pass_is_hosted_by_the_next_sampler_launch(emu, monkeypatch, implicit_sigma, max_norm, gemm):
    """RB_LEARNER_IMPLICIT_SIGMA on top (second and third case): the backward does not store the hidden layer's sigma-weight
    gradient, the hosted pass forms it from g_mu and the noise snapshot while it updates the (mu, sigma) pairs, and whatever runs
    the pass as a launch of its own (act, flush) materialises it first — with a clip that bites (max_norm 0.02) the scaled
    gradients it stores back include sigma's.  Same twin, same bit-identity, the stored gradient included.
    RB_LEARNER_DEFER_UPDATE: rb_learner_train_step leaves clip + Adam (agent.py:97-98) pending and the NEXT train_step's
    sampler launch carries it as extra workgroups (adam_body.h).  Against a twin without the flag, five steps with the
    device-resident step number: after every train_step the deferred handle's parameters are exactly ONE update behind;
    rb_learner_act (any entry point that reads parameters) runs the pending pass first and returns the twin's action; after
    rb_learner_flush everything — parameters, both moments, the stored gradient, the norm, the sum-tree, the sampled
    batches along the way — is bit-identical."""
    import ctypes as C
    from rainbow_amd import _lib as L
    name = "dataeff"
    c = scenarios.LEARN_CONFIGS[name]
    # (the library enables the pairing from 1 M-element layers on; last case: the weight gradient comes from the tiled GEMM of
    # fc_gemm.h — batch 256's path — whose epilogue leaves the sigma gradient out the same way)
    monkeypatch.setenv("RB_OPTS", "implicit_small=1,spec_draw=0" + (",fc_gemm=1" if gemm else ""))    # (the early draw has a twin test of its own)
    h1 = _ts_build(emu, name)
    h2 = _ts_build(emu, name)
    ctrs = []
    for (mem, rp, ad, o, job) in (h1, h2):
        ctr = mem.upload(np.zeros(1, np.int64))
        L.check(emu, emu.rb_learner_set_step_counter(ad.h, mem.ptr(ctr)))
        ctrs.append(ctr)
    L.check(emu, emu.rb_learner_set_flags(h1[2].h, L.LEARNER_DEFER_UPDATE | (L.LEARNER_IMPLICIT_SIGMA if implicit_sigma else 0)))
    state = h1[0].upload(scenarios.synth_state(np.random.RandomState(9), c["history"], 0).astype(np.float32) / 255.0)
    state2 = h2[0].upload(h1[0].download(state).copy())
    prev_twin = None
    for step in range(1, 6):
        beta = 0.4 + 0.05 * step
        snaps = []
        for (mem, rp, ad, o, job) in (h1, h2):
            ts = _ts_args(name, mem, rp, ad, o, job, beta, 0, max_norm)
            L.check(emu, emu.rb_learner_train_step(ad.h, C.byref(ts), None))
            snaps.append(_ts_snapshot(mem, rp, ad, o))
        a, b = snaps
        if implicit_sigma:      # the hidden layer's sigma gradient of THIS step is not in the flat gradient (its mu part is)
            sg = h1[2].layout["fc_h_v.weight_sigma"][0]
            assert not np.array_equal(a["grads"][sg:sg + 512], b["grads"][sg:sg + 512])
        for k in ("idx", "loss", "w", "noise", "tree"):             # the step itself never waits for the pending pass' results
            assert np.array_equal(a[k], b[k]), (step, k)            # ... because it has run by then (same launch as the sampler)
        if prev_twin is not None:                                   # one update behind, exactly
            for k in ("params", "m", "v"):
                assert np.array_equal(a[k], prev_twin[k]), (step, k)
        assert not np.array_equal(a["params"], b["params"])
        if step == 3:     # a reader of the parameters in between: the pending pass runs first, as a launch of its own
            outs = []
            for (mem, rp, ad, o, job), st in ((h1, state), (h2, state2)):
                act, q = mem.empty((1,), np.int32), mem.empty((1,), np.float32)
                L.check(emu, emu.rb_learner_act(ad.h, mem.ptr(st), 1, mem.ptr(act), mem.ptr(q), None))
                outs.append((int(mem.download(act)[0]), float(mem.download(q)[0])))
            assert outs[0] == outs[1]
            if implicit_sigma:  # act ran the pending pass (pairing included) as a launch of its own; a READER of grads_dev flushes,
                L.check(emu, emu.rb_learner_flush(h1[2].h, None))   # which forms the sigma gradient the backward left out
            a = _ts_snapshot(*h1[:4])
            for k in a:
                assert np.array_equal(a[k], b[k]), (step, k)
        prev_twin = b
    L.check(emu, emu.rb_learner_flush(h1[2].h, None))
    L.check(emu, emu.rb_learner_flush(h1[2].h, None))               # idempotent
    a, b = _ts_snapshot(*h1[:4]), _ts_snapshot(*h2[:4])
    for k in a:
        assert np.array_equal(a[k], b[k]), ("final", k)
    assert int(h1[0].download(ctrs[0])[0]) == 5 and int(h2[0].download(ctrs[1])[0]) == 5
    for (mem, rp, ad, o, job) in (h1, h2):
        ad.close(); rp.close()


def test_deferred_pass_and_batches_the_sampler_gives_up_on(emu):
    """The pending optimiser pass shares a launch with the NEXT call's sampler, which overwrites the replay header's
    last_status — the pass must obey the status of ITS OWN batch (k_head's copy).  With max_attempts = 1 on a 512-slot ring
    roughly a third of the draws are rejected (zero weights, no write-back, no optimiser update, no step number): eight
    steps on a deferring handle and on its twin must see the same mix of good and failed batches and end bit-identical."""
    import ctypes as C
    from rainbow_amd import _lib as L
    name = "dataeff"
    h1 = _ts_build(emu, name)
    h2 = _ts_build(emu, name)
    ctrs = []
    for (mem, rp, ad, o, job) in (h1, h2):
        ctr = mem.upload(np.zeros(1, np.int64))
        L.check(emu, emu.rb_learner_set_step_counter(ad.h, mem.ptr(ctr)))
        ctrs.append(ctr)
    L.check(emu, emu.rb_learner_set_flags(h1[2].h, L.LEARNER_DEFER_UPDATE))
    status = []
    for step in range(8):
        row = []
        for (mem, rp, ad, o, job) in (h1, h2):
            ts = _ts_args(name, mem, rp, ad, o, job, 0.5, 0)
            ts.max_attempts = 1
            L.check(emu, emu.rb_learner_train_step(ad.h, C.byref(ts), None))
            row.append(int(rp.raw_header().last_status))
        assert row[0] == row[1], step
        status.append(row[0])
    assert 0 in status and 1 in status, status                      # both kinds of step occurred
    assert any(a == 0 and b == 1 for a, b in zip(status, status[1:])), status    # a good step's pass hosted by a failing sampler
    L.check(emu, emu.rb_learner_flush(h1[2].h, None))
    a, b = _ts_snapshot(*h1[:4]), _ts_snapshot(*h2[:4])
    for k in a:
        assert np.array_equal(a[k], b[k]), ("final", k)
    good = status.count(0)
    assert int(h1[0].download(ctrs[0])[0]) == good and int(h2[0].download(ctrs[1])[0]) == good
    assert not np.array_equal(a["params"], h1[2]._flat(O.init_params(O.Config(**scenarios.LEARN_CONFIGS[name]), 1)))
    for (mem, rp, ad, o, job) in (h1, h2):
        ad.close(); rp.close()


def test_sync_target_copies_parameters_and_noise(emu):
    """Agent.update_target_net (agent.py:102-103) is load_state_dict(online.state_dict()): the registered epsilon BUFFERS
    travel with the parameters (model.py:19,22).  rb_learner_sync_target must therefore leave target == online for both
    the parameter buffer and the (factorised) noise buffer — and leave the online side alone."""
    from rainbow_amd import _lib as L
    name = "atoms21"
    c = scenarios.LEARN_CONFIGS[name]
    cfg = O.Config(**c)
    ad = CAbiLearnAdapter(emu, NumpyMem(), name)
    ad.load(O.init_params(cfg, 3), O.init_params(cfg, 4))
    rs = np.random.RandomState(8)
    draws = O.noise_draw_count(cfg)
    ad.reset_noise_online(rs.randn(draws).astype(np.float32))
    r = ad.mem.upload(rs.randn(draws).astype(np.float32))
    L.check(emu, emu.rb_learner_reset_noise(ad.h, 1, ad.mem.ptr(r), None))          # a DIFFERENT target noise first
    p_on, z_on = ad.p_on.copy(), ad.z_on.copy()
    assert not np.array_equal(ad.z_tg, z_on) and not np.array_equal(ad.p_tg, p_on)
    assert np.abs(z_on).sum() > 0
    L.check(emu, emu.rb_learner_sync_target(ad.h, None))
    assert np.array_equal(ad.p_tg, p_on) and np.array_equal(ad.z_tg, z_on)
    assert np.array_equal(ad.p_on, p_on) and np.array_equal(ad.z_on, z_on)
    ad.close()


def test_trajectory_tracks_reference_for_30_steps(emu):
    """north_star: 'loss curves matching reference within tolerance'.  The first 30 steps of the 200-step reference
    trajectory tests/golden/learn_k200.npz (real agent.py:61-100 with injected noise, a fresh batch per step; the GPU test
    runs all 200 steps incl. the update_target_net at step 100): per-step loss and gradient norm, gradients and
    parameters at step 29, with the one-step tolerances of helpers.assert_learn_trace_matches — the drift of this path
    against the reference does not grow beyond them (measured: parameters within 3e-8 of the reference after 200 steps)."""
    ad = CAbiLearnAdapter(emu, NumpyMem(), "k200")
    trace = scenarios.learn_scenario(ad, "k200", O, steps=30)
    golden = load_golden("learn_k200.npz")
    assert any(k.startswith("s29_param/") for k in trace)
    assert_learn_trace_matches(trace, {k: golden[k] for k in trace}, label="emu/k200[:30]")
    ad.close()


def test_large_batch_fc_backward_matches_oracle(emu, monkeypatch):
    """Batch 64 on the data-efficient stack with hidden 64: the batch > 32 bodies of the noisy-linear backward (weight
    gradient without the pipelined one-pass body, the transposed-dh operand of the input gradient, image groups in the conv
    kernels) against the oracle: loss and every gradient (the GPU runs the same check at batch 256)."""
    nbatch = 64
    cfgd = dict(scenarios.LEARN_CONFIGS["dataeff"], batch=nbatch, multi_step=3, hidden=64)
    monkeypatch.setitem(scenarios.LEARN_CONFIGS, "wide64", cfgd)
    cfg = O.Config(**cfgd)
    ad = CAbiLearnAdapter(emu, NumpyMem(), "wide64")
    online, target = O.init_params(cfg, 277), O.init_params(cfg, 278)
    ad.load(online, target)
    rs = np.random.RandomState(25)
    draws = O.noise_draw_count(cfg)
    raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
    ad.reset_noise_online(raw_on)
    batch = scenarios.make_batch(cfgd, 421)
    got = ad.learn_step(batch, raw_tg)
    want = O.learn(cfg, online, target, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg), batch)
    total, clipped = O.clip_grads(want["grads"], scenarios.LEARN_HYPER["norm_clip"])
    np.testing.assert_allclose(got["loss"], want["loss"], rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(got["grad_norm"], total, rtol=5e-5)
    for k, g in clipped.items():
        scale = float(np.max(np.abs(g))) if g.size else 0.0
        np.testing.assert_allclose(got["grads"][k], g, rtol=2e-4, atol=5e-6 * scale + 1e-9, err_msg=k)
    ad.close()


@pytest.mark.parametrize("world", [3, 8])
def test_replica_exchange_kernels_fold_more_than_two_blocks(emu, world):
    """The replica exchange of SURVEY 8(e) (insert point agent.py:96-97) with MORE than two replicas on the host interpreter:
    `world` learner handles with their own batch and noise, k_pack_factors in every learn call, a host concatenation for the
    all-gather, k_finish_grads folding `world` blocks in rank order (M = world x B gathered rows, scale 1 / world), then
    clip + Adam: every handle bit-identical, gradients and parameters == the oracle fed with the mean of the `world`
    gradients.  (World 8 — BASELINE config 5 — runs at the canonical H = 512 / B = 32 shape on the GPU, tests/test_exchange_gpu.py.)"""
    import ctypes as C
    from rainbow_amd import _lib as L
    name = "dataeff"
    cfgd = scenarios.LEARN_CONFIGS[name]
    cfg = O.Config(**cfgd)
    hy = scenarios.LEARN_HYPER
    ads = [CAbiLearnAdapter(emu, NumpyMem(), name) for _ in range(world)]
    online, target = O.init_params(cfg, 811), O.init_params(cfg, 812)
    for ad in ads:
        ad.load(online, target)
    f = C.c_int64(0)
    L.check(emu, emu.rb_learner_exchange_layout(ads[0].h, C.byref(f), None, None))
    local = [np.zeros(f.value, dtype=np.float32) for _ in ads]
    gathered = [np.zeros(world * f.value, dtype=np.float32) for _ in ads]
    for ad, lo, al in zip(ads, local, gathered):
        L.check(emu, emu.rb_learner_set_exchange(ad.h, world, lo.ctypes.data, al.ctypes.data))
    adam = O.AdamOracle(online, hy["lr"], hy["adam_eps"])
    draws = O.noise_draw_count(cfg)
    got_t, want_t = {}, {}
    for k in range(2):
        per_rank = []
        for r, ad in enumerate(ads):
            rs = np.random.RandomState(500 + 10 * k + r)
            raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
            batch = scenarios.make_batch(cfgd, 600 + 10 * k + r)
            ad.reset_noise_online(raw_on)
            ad.learn_only(batch, raw_tg)
            per_rank.append(O.learn(cfg, online, target, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg), batch))
        cat = np.concatenate(local)
        for ad, al in zip(ads, gathered):
            al[:] = cat
            L.check(emu, emu.rb_learner_finish_grads(ad.h, None))
        outs = [ad.finish_step() for ad in ads]
        for r in range(1, world):
            assert np.array_equal(ads[0].grads, ads[r].grads), "step %d: gradients of replica %d differ" % (k, r)
            assert np.array_equal(ads[0].params()["fc_h_v.weight_mu"], ads[r].params()["fc_h_v.weight_mu"])
            assert outs[0]["grad_norm"] == outs[r]["grad_norm"]
        gmean = {n: sum(pr["grads"][n].astype(np.float64) for pr in per_rank).astype(np.float32) / np.float32(world)
                 for n in per_rank[0]["grads"]}
        total, clipped = O.clip_grads(gmean, hy["norm_clip"])
        online = adam.step(clipped)
        got_t["s%d_grad_norm" % k], want_t["s%d_grad_norm" % k] = np.float32(outs[0]["grad_norm"]), np.float32(total)
        for r in range(world):
            got_t["s%d_r%d_loss" % (k, r)], want_t["s%d_r%d_loss" % (k, r)] = outs[r]["loss"], per_rank[r]["loss"]
        for n in clipped:
            got_t["s%d_grad/%s" % (k, n)], want_t["s%d_grad/%s" % (k, n)] = outs[0]["grads"][n], clipped[n]
        for n, p in ads[0].params().items():
            got_t["s%d_param/%s" % (k, n)], want_t["s%d_param/%s" % (k, n)] = p, online[n]
    assert_learn_trace_matches(got_t, want_t, label="exchange-emu/world%d" % world)
    for ad in ads:
        emu.rb_learner_set_exchange(ad.h, 1, None, None)
        ad.close()


def test_early_draw_on_the_replays_stream_is_bit_identical(emu, monkeypatch):
    """RB_OPTS spec_draw=1 (replay_internal.h rb_replay_spec_launch): from the third back-to-back rb_learner_train_step on, the
    priority write-back of call k (agent.py:100) and the draw of call k + 1 (agent.py:63) run behind call k's head kernel on the
    replay's own stream; call k + 1's sampler launch ACCEPTS the draw.  Twin without the option, ten calls with device RNG:
    per-sample loss, parameters, Adam moments, norm, noise, the sum-tree and the replay header (Philox counter, attempts, status)
    bit-identical after every call — through accepted draws (calls 3, 4, 7), a REJECTED one (beta changes before call 5: the
    tentative draw waits in the sampler launch, is discarded and redrawn), a CANCELLED one (an append before call 8 joins the
    stream) and the streak building up again.  With an early draw accepted, the twin's batch of call k + 1 is what the
    speculating handle's buffers already held after call k."""
    import ctypes as C
    from rainbow_amd import _lib as L
    name = "dataeff"
    c = scenarios.LEARN_CONFIGS[name]
    monkeypatch.setenv("RB_OPTS", "spec_draw=1")
    h1 = _ts_build(emu, name)
    monkeypatch.setenv("RB_OPTS", "spec_draw=0")
    h2 = _ts_build(emu, name)
    rs = np.random.RandomState(77)
    extra = [(scenarios.synth_state(rs, c["history"], 0), int(rs.randint(0, c["actions"])), 0.0, False) for _ in range(3)]
    idx_after = {}
    accepted = 0
    for step in range(1, 11):
        beta = 0.4 if step < 5 else 0.55
        if step == 8:                          # something else touches the replay between two calls: the early draw is cancelled
            for (mem, rp, ad, o, job) in (h1, h2):
                for tr in extra:
                    rp.append(*tr)
        snaps, hdrs = [], []
        for (mem, rp, ad, o, job) in (h1, h2):
            ts = _ts_args(name, mem, rp, ad, o, job, beta, step)
            L.check(emu, emu.rb_learner_train_step(ad.h, C.byref(ts), None))
            assert emu.rb_learner_priority_written(ad.h) == 1
            snaps.append(_ts_snapshot(mem, rp, ad, o))
            h = rp.raw_header()
            hdrs.append((h.index, h.full, h.max, h.total, h.last_attempts, h.last_status, h.rng_counter))
        a, b = snaps
        for k in ("loss", "params", "m", "v", "noise", "tree", "norm", "grads"):
            assert np.array_equal(a[k], b[k]), (step, k)
        assert hdrs[0][:4] == hdrs[1][:4], step
        idx_after[step] = a["idx"]
        if np.array_equal(idx_after.get(step - 1), b["idx"]) and step >= 3:
            accepted += 1                       # the speculating handle held call `step`'s batch one call early
            assert hdrs[0][4:6] == hdrs[1][4:6]
        elif not np.array_equal(a["idx"], b["idx"]):
            pass                                # (an early draw for the NEXT call is in the buffers: compared one call later)
    assert accepted >= 3, accepted
    # the Philox counter: the twin's is final; the speculating handle's last tentative draw is not committed until accepted
    for (mem, rp, ad, o, job) in (h1, h2):
        L.check(emu, emu.rb_replay_sample(rp.h, c["batch"], 0.55, None, 64, mem.ptr(o["tree_idx"]), None, None, mem.ptr(o["actions"]),
                                          mem.ptr(o["returns"]), mem.ptr(o["nonterm"]), mem.ptr(o["weights"]), None))
    assert np.array_equal(h1[0].download(h1[3]["tree_idx"]), h2[0].download(h2[3]["tree_idx"]))
    ha, hb = h1[1].raw_header(), h2[1].raw_header()
    assert (ha.rng_counter, ha.last_attempts, ha.last_status) == (hb.rng_counter, hb.last_attempts, hb.last_status)
    for (mem, rp, ad, o, job) in (h1, h2):
        ad.close(); rp.close()



@pytest.mark.parametrize("preceding", [4])      # (the tentative draw then sits in table 1; 3 and 4 both run on the GPU)
def test_public_sample_after_an_early_draw_reads_table_zero(emu, monkeypatch, preceding):
    ts_scenarios.public_sample_after_early_draw_check(emu, NumpyMem, monkeypatch, preceding)


def test_expired_gate_of_the_early_draw_fails_safe(emu, monkeypatch):
    ts_scenarios.early_draw_expiry_check(emu, NumpyMem, monkeypatch)
