"""the frame preprocessing of env.py:27-29,57-69 (84x84 bilinear resize of the raw grayscale screen,
max over the last two frames, / 255).  PARITY UNPINNED against cv2 itself (absent here); pinned are
  - the oracle restatement of OpenCV's fixed-point INTER_LINEAR by properties (below), and
  - the device kernel bit-for-bit against that restatement (host interpreter on CPU, HIP on the GPU)."""
import ctypes as C

import numpy as np
import pytest

from oracle import frame_oracle as F


def _screens(seed, n=1):
    rs = np.random.RandomState(seed)
    a = rs.randint(0, 256, size=(n, 210, 160)).astype(np.uint8)
    a[:, 50:120, 30:90] = rs.randint(0, 256, size=(n, 1, 1)).astype(np.uint8)     # flat regions like a game screen
    return a


def test_oracle_resize_properties():
    rs = np.random.RandomState(1)
    for v in (0, 1, 137, 255):                       # constants are preserved (coefficients sum to 2048 exactly)
        assert (F.resize_linear_u8(np.full((210, 160), v, np.uint8)) == v).all()
    img = rs.randint(0, 256, size=(84, 84)).astype(np.uint8)
    assert np.array_equal(F.resize_linear_u8(img), img)                            # scale 1: identity
    a = _screens(2)[0]
    r = F.resize_linear_u8(a)
    assert r.shape == (84, 84) and r.dtype == np.uint8
    # a bilinear sample lies between the extremes of its 2 x 2 neighbourhood (up to the 2^-2 final rounding)
    sy, sx = (np.floor((np.arange(84) + 0.5) * 2.5 - 0.5)).astype(int), (np.floor((np.arange(84) + 0.5) * (160 / 84) - 0.5)).astype(int)
    for dy in (0, 1, 40, 83):
        for dx in (0, 1, 41, 83):
            ys = np.clip([sy[dy], sy[dy] + 1], 0, 209)
            xs = np.clip([sx[dx], sx[dx] + 1], 0, 159)
            nb = a[np.ix_(ys, xs)].astype(int)
            assert nb.min() - 1 <= int(r[dy, dx]) <= nb.max() + 1
    # against float bilinear interpolation at the same sample points: the fixed-point result is within 1 grey level
    fy = (np.arange(84) + 0.5) * 2.5 - 0.5
    fx = (np.arange(84) + 0.5) * (160 / 84) - 0.5
    y0 = np.floor(fy).astype(int); wy = fy - y0
    x0 = np.floor(fx).astype(int); wx = fx - x0
    y0c, y1c = np.clip(y0, 0, 209), np.clip(y0 + 1, 0, 209)
    x0c, x1c = np.clip(x0, 0, 159), np.clip(x0 + 1, 0, 159)
    A = a.astype(np.float64)
    ref = ((A[np.ix_(y0c, x0c)] * (1 - wx) + A[np.ix_(y0c, x1c)] * wx) * (1 - wy)[:, None]
           + (A[np.ix_(y1c, x0c)] * (1 - wx) + A[np.ix_(y1c, x1c)] * wx) * wy[:, None])
    assert np.abs(r.astype(np.float64) - ref).max() <= 1.0
    # the observation: max of two states == state of the max, values are k / 255
    b = _screens(3)[0]
    o = F.observe(a, b)
    assert np.array_equal(o, np.maximum(F.get_state(a), F.get_state(b)))
    assert np.array_equal((o * 255).round().astype(np.uint8).astype(np.float32) / np.float32(255), o)


def _device_observe(lib, mem, a, b):
    from rainbow_amd import _lib as L
    n, H, W = a.shape
    da = mem.upload(a)
    db = mem.upload(b) if b is not None else None
    out = mem.empty((n, 84, 84), np.float32)
    L.check(lib, lib.rb_frame_preprocess(mem.ptr(da), mem.ptr(db), H, W, n, mem.ptr(out), mem.stream))
    mem.sync()
    return mem.download(out)


def _check_backend(lib, mem):
    a, b = _screens(10, 3), _screens(11, 3)
    got = _device_observe(lib, mem, a, b)
    want = np.stack([F.observe(a[i], b[i]) for i in range(3)])
    assert np.array_equal(got, want)
    got1 = _device_observe(lib, mem, a[:1], None)                                  # env.py:50 (reset): a single frame
    assert np.array_equal(got1[0], F.get_state(a[0]))
    odd = np.random.RandomState(5).randint(0, 256, size=(1, 97, 131)).astype(np.uint8)   # another screen geometry
    assert np.array_equal(_device_observe(lib, mem, odd, None)[0],
                          (F.resize_linear_u8(odd[0]).astype(np.float32) / np.float32(255)))
    from rainbow_amd import _lib as L
    with pytest.raises(L.RainbowError):
        L.check(lib, lib.rb_frame_preprocess(None, None, 210, 160, 1, None, None))


def test_host_interpreted_kernel_matches_oracle():
    from cabi_adapter import NumpyMem
    from hipemu import loader
    _check_backend(loader.load(), NumpyMem())


@pytest.mark.gpu
def test_hip_kernel_matches_oracle_and_feeds_append():
    import torch
    from cabi_adapter import TorchMem
    from rainbow_amd import _lib
    from rainbow_amd.frames import FramePreprocessor
    _check_backend(_lib.load(), TorchMem())
    # the drop-in helper: raw screens -> observation -> state deque -> ReplayMemory.append stores the resized bytes
    import types
    from rainbow_amd.memory import ReplayMemory
    dev = torch.device("cuda:0")
    pre = FramePreprocessor(dev)
    a, b = _screens(20)[0], _screens(21)[0]
    obs = pre.observe(a, b)
    assert obs.shape == (84, 84) and np.array_equal(obs.cpu().numpy(), F.observe(a, b))
    args = types.SimpleNamespace(device=dev, history_length=4, discount=0.99, multi_step=3, priority_weight=0.4, priority_exponent=0.5)
    mem = ReplayMemory(args, 64)
    state = torch.stack([torch.zeros(84, 84, device=dev)] * 3 + [obs])
    mem.append(state, 1, 0.0, False)
    torch.cuda.synchronize()
    stored = mem._grab("frames")[0].reshape(84, 84)
    want = np.maximum(F.resize_linear_u8(a), F.resize_linear_u8(b))
    x = (want.astype(np.float32) / np.float32(255)) * np.float32(255)              # memory.py:106: mul(255) truncating
    assert np.array_equal(stored, x.astype(np.uint8))
    batch = pre.observe(_screens(30, 5), _screens(31, 5))
    assert batch.shape == (5, 84, 84)
