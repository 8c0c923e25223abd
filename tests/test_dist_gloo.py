"""world_size-2 test of the replica path on CPU (gloo): two processes run the learn step on different
batches (host-interpreted kernels), average the flat gradient with rainbow_amd.dist, clip and step Adam.
Replicas must stay bit-identical and match the oracle fed with the averaged gradient."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# the factored exchange runs on the streamed noisy-linear kernels (F and H multiples of 32): the data-efficient fixture
_NAME = {"factored": "dataeff", "allreduce": "atoms21"}


def _worker(rank, world, port, outdir, mode="allreduce"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import scenarios
    from cabi_adapter import CAbiLearnAdapter, NumpyMem
    from hipemu import loader
    from oracle import learner_oracle as O
    from rainbow_amd import dist as rdist
    name = _NAME[mode]
    c = scenarios.LEARN_CONFIGS[name]
    cfg = O.Config(**c)
    ad = CAbiLearnAdapter(loader.load(), NumpyMem(), name)
    if mode == "factored":     # all-gather of the FC gradient factors + all-reduce of the conv gradients
        ad.exchange = rdist.FactoredExchange(ad.lib, ad.h, torch.from_numpy(ad.grads))
    else:                      # one all-reduce of the flat gradient
        ad.grad_hook = rdist.average_gradients
    online, target = O.init_params(cfg, 5), O.init_params(cfg, 6)
    if rank != 0:   # replicas start different on purpose; broadcast must fix it
        online = {k: v + 1.0 for k, v in online.items()}
    ad.load(online, target)
    rdist.broadcast_parameters(ad.param_t.detach(), 0)
    draws = O.noise_draw_count(cfg)
    results = []
    for k in range(2):
        rs = np.random.RandomState(100 + 10 * k + rank)     # per-replica noise and data
        raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
        ad.reset_noise_online(raw_on)
        out = ad.learn_step(scenarios.make_batch(c, 200 + 10 * k + rank), raw_tg)
        results.append(out["grad_norm"])
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), params=ad.mem.download(ad.p_on), norms=np.array(results),
             grads=ad.mem.download(ad.grads))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["factored", "allreduce"])
def test_two_replicas_stay_identical(tmp_path, mode):
    world = 2
    import socket
    with socket.socket() as sk:                      # a free port (the two modes may run in parallel pytest-xdist workers)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path), mode), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert np.array_equal(r0["params"], r1["params"]), "replicas diverged"
    assert np.array_equal(r0["grads"], r1["grads"])
    assert np.array_equal(r0["norms"], r1["norms"])

    # oracle: same two steps with the mean of the two replicas' gradients
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import scenarios
    from hipemu import loader
    from cabi_adapter import CAbiLearnAdapter, NumpyMem
    from oracle import learner_oracle as O
    name = _NAME[mode]
    c = scenarios.LEARN_CONFIGS[name]
    cfg = O.Config(**c)
    online, target = O.init_params(cfg, 5), O.init_params(cfg, 6)
    adam = O.AdamOracle(online, scenarios.LEARN_HYPER["lr"], scenarios.LEARN_HYPER["adam_eps"])
    draws = O.noise_draw_count(cfg)
    for k in range(2):
        gs = []
        for rank in range(2):
            rs = np.random.RandomState(100 + 10 * k + rank)
            raw_on, raw_tg = rs.randn(draws).astype(np.float32), rs.randn(draws).astype(np.float32)
            out = O.learn(cfg, online, target, O.make_noise(cfg, raw_on), O.make_noise(cfg, raw_tg),
                          scenarios.make_batch(c, 200 + 10 * k + rank))
            gs.append(out["grads"])
        mean = {n: (gs[0][n] + gs[1][n]) / np.float32(2) for n in gs[0]}
        total, clipped = O.clip_grads(mean, scenarios.LEARN_HYPER["norm_clip"])
        online = adam.step(clipped)
        np.testing.assert_allclose(r0["norms"][k], total, rtol=2e-5)
    ad = CAbiLearnAdapter(loader.load(), NumpyMem(), name)
    got = ad._unflat(r0["params"])
    for n in online:
        np.testing.assert_allclose(got[n], online[n], rtol=0, atol=3e-7, err_msg=n)
