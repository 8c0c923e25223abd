"""rb_learner_train_step twins shared by the host-interpreter tests (NumpyMem) and the GPU tests (TorchMem): a filled replay, a
learner with a priority sink into it, the buffers of one call — and the scenarios around the early draw (RB_OPTS spec_draw=1)."""
import ctypes as C

import numpy as np

import scenarios
from cabi_adapter import CAbiLearnAdapter, CAbiReplayAdapter
from oracle import learner_oracle as O
from rainbow_amd import _lib as L


def ts_build(lib, Mem, name):
    """A filled replay, a learner with a priority sink into it and the buffers of one rb_learner_train_step call."""
    c = scenarios.LEARN_CONFIGS[name]
    B, h, n = c["batch"], c["history"], c["multi_step"]
    mem = Mem()
    rp = CAbiReplayAdapter(lib, mem, 512, h, n, c["discount"], 0.5)
    rs = np.random.RandomState(5)
    for _ in range(600):
        rp.append(scenarios.synth_state(rs, h, 0), int(rs.randint(0, c["actions"])), float(rs.choice([-1.0, 0.0, 1.0])),
                  bool(rs.random_sample() < 0.05))
    ad = CAbiLearnAdapter(lib, mem, name)
    cfg = O.Config(**c)
    ad.load(O.init_params(cfg, 1), O.init_params(cfg, 2))
    ad.reset_noise_online(rs.randn(O.noise_draw_count(cfg)).astype(np.float32))
    out = dict(tree_idx=mem.empty((B,), np.int64), actions=mem.empty((B,), np.int64), returns=mem.empty((B,), np.float32),
               nonterm=mem.empty((B,), np.float32), weights=mem.empty((B,), np.float32), loss=mem.empty((B,), np.float32),
               norm=mem.empty((1,), np.float32))
    L.check(lib, lib.rb_learner_set_priority_sink(ad.h, rp.h, mem.ptr(out["tree_idx"])))
    job = L.NoiseJob()
    L.check(lib, lib.rb_learner_noise_job(ad.h, 1, C.byref(job)))
    return mem, rp, ad, out, job


def ts_snapshot(mem, rp, ad, out):
    mem.sync()          # (the whole device: an early pair may still be running on the replay's own stream)
    return dict(idx=mem.download(out["tree_idx"]).copy(), loss=mem.download(out["loss"]).copy(),
                w=mem.download(out["weights"]).copy(), params=mem.download(ad.p_on).copy(),
                m=mem.download(ad.adam_m).copy(), v=mem.download(ad.adam_v).copy(), noise=mem.download(ad.z_tg).copy(),
                tree=rp.tree().copy(), norm=mem.download(out["norm"]).copy(), grads=mem.download(ad.grads).copy())


def ts_args(name, mem, rp, ad, o, job, beta, step, max_norm=None):
    hy = scenarios.LEARN_HYPER
    return L.TrainStep(replay=rp.h, batch=scenarios.LEARN_CONFIGS[name]["batch"], max_attempts=64, window_len=rp.bufs.window_len,
                       priority_weight=beta, tree_idx_dev=mem.ptr(o["tree_idx"]), actions_dev=mem.ptr(o["actions"]),
                       returns_dev=mem.ptr(o["returns"]), nonterminals_dev=mem.ptr(o["nonterm"]), weights_dev=mem.ptr(o["weights"]),
                       noise_job=C.addressof(job), frames_dev=rp.bufs.frames_dev, windows_dev=rp.bufs.window_dev,
                       loss_dev=mem.ptr(o["loss"]), exp_avg_dev=mem.ptr(ad.adam_m), exp_avg_sq_dev=mem.ptr(ad.adam_v),
                       norm_dev=mem.ptr(o["norm"]), lr=hy["lr"], beta1=0.9, beta2=0.999, eps=hy["adam_eps"], step=step,
                       max_norm=hy["norm_clip"] if max_norm is None else max_norm)




def early_draw_expiry_check(lib, Mem, monkeypatch):
    """VERDICT r5 item 2 / ADVICE r5: a cross-stream wait of the early draw that EXPIRES must fail safe.  RB_OPTS spec_stall=1
    (test hook) makes the launch behind the head kernel skip the store of the go flag, so the gate in front of the early pair
    (k_spec_gate) runs into its ~2 ms bound in call 2.  Required: the pair gives up — NO write-back from losses that might not be
    final (counted: rb_replay_dropped_updates == 1), no tentative draw; call 3's sampler launch sees the aborted record and draws
    itself; the expiry is counted (rb_replay_expired_waits == 1) and switches the early draw off on the handle, so calls 3 and 4
    keep write-back and draw in their own launches.  Twin: spec_draw=0, with the priority sink taken away for call 2 only (the one
    write-back the stalled pair dropped).  Everything bit-identical after every call: batch, loss, parameters, moments, noise, norm,
    the sum-tree, the header.  rb_replay_reset_failed_samples clears the count and allows the early draw again."""
    name = "dataeff"
    monkeypatch.setenv("RB_OPTS", "spec_draw=1,spec_stall=1")
    h1 = ts_build(lib, Mem, name)
    monkeypatch.setenv("RB_OPTS", "spec_draw=0")
    h2 = ts_build(lib, Mem, name)

    def counts(rp):
        e, d = C.c_int64(-1), C.c_int64(-1)
        L.check(lib, lib.rb_replay_expired_waits(rp.h, C.byref(e)))
        L.check(lib, lib.rb_replay_dropped_updates(rp.h, C.byref(d)))
        return int(e.value), int(d.value)

    for step in range(1, 5):
        for which, (mem, rp, ad, o, job) in enumerate((h1, h2)):
            if which == 1 and step == 2:       # the twin of the dropped write-back: no sink for this one call
                L.check(lib, lib.rb_learner_set_priority_sink(ad.h, None, None))
            ts = ts_args(name, mem, rp, ad, o, job, 0.4, step)
            L.check(lib, lib.rb_learner_train_step(ad.h, C.byref(ts), mem.stream))
            if which == 1 and step == 2:
                L.check(lib, lib.rb_learner_set_priority_sink(ad.h, rp.h, mem.ptr(o["tree_idx"])))
        a, b = ts_snapshot(*h1[:4]), ts_snapshot(*h2[:4])
        if step == 2:
            # (the stalled pair never drew: the speculating handle's buffers still hold call 2's own batch, like the twin's)
            assert counts(h1[1]) == (1, 1), counts(h1[1])
        for k in a:
            assert np.array_equal(a[k], b[k]), (step, k)
        ha, hb = h1[1].raw_header(), h2[1].raw_header()
        assert (ha.index, ha.full, ha.max, ha.total, ha.last_attempts, ha.last_status, ha.rng_counter) == \
               (hb.index, hb.full, hb.max, hb.total, hb.last_attempts, hb.last_status, hb.rng_counter), step
    assert counts(h1[1]) == (1, 1) and counts(h2[1]) == (0, 0)      # disabled after the first expiry: no second one
    L.check(lib, lib.rb_replay_reset_failed_samples(h1[1].h))
    assert counts(h1[1]) == (0, 0)
    # allowed again: the next pair stalls again (the hook is still on) and is counted again
    for step in range(5, 7):
        mem, rp, ad, o, job = h1
        ts = ts_args(name, mem, rp, ad, o, job, 0.4, step)
        L.check(lib, lib.rb_learner_train_step(ad.h, C.byref(ts), mem.stream))
    h1[0].sync()
    e, d = counts(h1[1])       # (on the device the host runs ahead: a second pair may have been launched before the first expiry was seen)
    assert e >= 1 and d >= 1, (e, d)
    for (mem, rp, ad, o, job) in (h1, h2):
        ad.close(); rp.close()


def public_sample_after_early_draw_check(lib, Mem, monkeypatch, preceding):
    """with RB_OPTS spec_draw=1, `preceding` back-to-back rb_learner_train_step calls at constant beta leave a
    tentative draw in flight in the replay's OTHER window table.  A caller that then issues the step's entry points one by one —
    rb_replay_sample_fused_noise + rb_learner_learn_windows(rb_replay_buffers_t.window_dev) + rb_learner_clip_adam, what
    Agent._learn_eager does with the cached frame_source() pointer — must get the windows of ITS draw in table 0: the public
    sample entry points never accept the tentative draw (only rb_learner_train_step does, which resolves the current table
    itself).  Twin with spec_draw=0; an odd and an even number of preceding steps (the live table alternates), everything
    bit-identical, and one more train_step afterwards."""
    name = "dataeff"
    c = scenarios.LEARN_CONFIGS[name]
    B = c["batch"]
    hy = scenarios.LEARN_HYPER
    monkeypatch.setenv("RB_OPTS", "spec_draw=1")
    h1 = ts_build(lib, Mem, name)
    monkeypatch.setenv("RB_OPTS", "spec_draw=0")
    h2 = ts_build(lib, Mem, name)
    step = 0
    for _ in range(preceding):
        step += 1
        for (mem, rp, ad, o, job) in (h1, h2):
            ts = ts_args(name, mem, rp, ad, o, job, 0.4, step)
            L.check(lib, lib.rb_learner_train_step(ad.h, C.byref(ts), mem.stream))
    # (a tentative draw IS in flight: the speculating handle's sample buffers already hold the NEXT call's batch)
    h1[0].sync()
    assert not np.array_equal(h1[0].download(h1[3]["tree_idx"]), h2[0].download(h2[3]["tree_idx"])), "the scenario needs a tentative draw in flight"
    step += 1
    for (mem, rp, ad, o, job) in (h1, h2):
        L.check(lib, lib.rb_replay_sample_fused_noise(rp.h, B, 0.4, None, 64, mem.ptr(o["tree_idx"]), None, None, mem.ptr(o["actions"]),
                                                     mem.ptr(o["returns"]), mem.ptr(o["nonterm"]), mem.ptr(o["weights"]), C.byref(job), mem.stream))
        L.check(lib, lib.rb_learner_learn_windows(ad.h, rp.bufs.frames_dev, rp.bufs.window_dev, rp.bufs.window_len, mem.ptr(o["actions"]),
                                                  mem.ptr(o["returns"]), mem.ptr(o["nonterm"]), mem.ptr(o["weights"]), mem.ptr(o["loss"]), mem.stream))
        L.check(lib, lib.rb_learner_clip_adam(ad.h, hy["norm_clip"], mem.ptr(ad.adam_m), mem.ptr(ad.adam_v), hy["lr"], 0.9, 0.999,
                                              hy["adam_eps"], step, mem.ptr(o["norm"]), mem.stream))
    a, b = ts_snapshot(*h1[:4]), ts_snapshot(*h2[:4])
    for k in a:
        assert np.array_equal(a[k], b[k]), ("three calls after %d train_steps" % preceding, k)
    step += 1
    for (mem, rp, ad, o, job) in (h1, h2):
        ts = ts_args(name, mem, rp, ad, o, job, 0.4, step)
        L.check(lib, lib.rb_learner_train_step(ad.h, C.byref(ts), mem.stream))
    a, b = ts_snapshot(*h1[:4]), ts_snapshot(*h2[:4])
    for k in a:
        assert np.array_equal(a[k], b[k]), ("train_step after the three calls", k)
    for (mem, rp, ad, o, job) in (h1, h2):
        ad.close(); rp.close()


def captured_train_step_check(lib, Mem, monkeypatch):
    """rb_learner_train_step captured into a hipGraph with RB_OPTS spec_draw=1.  Two eager calls arm the early
    draw (streak >= 1); the THIRD call is captured: under stream capture the library must neither launch the early pair (it would run
    at once, outside the graph, waiting for a flag the captured kernels only store on replay) nor bake an accepted tentative draw
    into the captured sampler launch (replays would skip both the draw and the priority write-back).  The graph is replayed four
    times — beta from the device word (rb_replay_set_beta_source), the step number from the device counter — against an eager twin
    with spec_draw=0: batch, loss, parameters, moments, noise, norm, the sum-tree and the header bit-identical after every replay."""
    import torch
    name = "dataeff"
    monkeypatch.setenv("RB_OPTS", "spec_draw=1")
    h1 = ts_build(lib, Mem, name)
    monkeypatch.setenv("RB_OPTS", "spec_draw=0")
    h2 = ts_build(lib, Mem, name)
    keep = []
    for (mem, rp, ad, o, job) in (h1, h2):
        ctr = mem.upload(np.zeros(1, np.int64))
        nb = mem.upload(np.array([-0.4], np.float32))
        L.check(lib, lib.rb_learner_set_step_counter(ad.h, mem.ptr(ctr)))
        L.check(lib, lib.rb_replay_set_beta_source(rp.h, mem.ptr(nb)))
        keep.append((ctr, nb))

    def call(h, stream):
        mem, rp, ad, o, job = h
        ts = ts_args(name, mem, rp, ad, o, job, 0.4, 0)
        L.check(lib, lib.rb_learner_train_step(ad.h, C.byref(ts), stream))

    for _ in range(2):
        call(h1, h1[0].stream)
        call(h2, h2[0].stream)
    a, b = ts_snapshot(*h1[:4]), ts_snapshot(*h2[:4])
    assert np.array_equal(a["params"], b["params"]) and np.array_equal(a["tree"], b["tree"])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        call(h1, torch.cuda.current_stream().cuda_stream)
    e = C.c_int64(-1)
    for rep in range(4):
        g.replay()
        call(h2, h2[0].stream)
        a, b = ts_snapshot(*h1[:4]), ts_snapshot(*h2[:4])
        for k in a:
            assert np.array_equal(a[k], b[k]), ("replay %d" % rep, k)
        ha, hb = h1[1].raw_header(), h2[1].raw_header()
        assert (ha.index, ha.full, ha.max, ha.total, ha.last_attempts, ha.last_status, ha.rng_counter) == \
               (hb.index, hb.full, hb.max, hb.total, hb.last_attempts, hb.last_status, hb.rng_counter), rep
    L.check(lib, lib.rb_replay_expired_waits(h1[1].h, C.byref(e)))
    assert e.value == 0
    del g
    for (mem, rp, ad, o, job) in (h1, h2):
        ad.close(); rp.close()
