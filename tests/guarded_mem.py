"""Caller-owned 'device' buffers with canaries: every allocation the test hands to the C ABI sits between two 4 KiB bands
of a fixed byte; check() names any buffer whose bands a kernel wrote into.  Drop-in for cabi_adapter.NumpyMem / TorchMem."""
import numpy as np

from cabi_adapter import NumpyMem, TorchMem

GUARD = 4096
FILL = 0xC5


class GuardedNumpyMem(NumpyMem):
    def __init__(self):
        self.blocks = []

    def empty(self, shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        raw = np.full(nbytes + 2 * GUARD, FILL, dtype=np.uint8)
        user = raw[GUARD:GUARD + nbytes]
        user[:] = 0
        self.blocks.append((raw, nbytes, "%s%s" % (np.dtype(dtype).name, tuple(shape))))
        return user.view(dtype).reshape(shape)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        out = self.empty(arr.shape, arr.dtype)
        out[...] = arr
        return out

    def check(self):
        bad = [name for raw, n, name in self.blocks if not ((raw[:GUARD] == FILL).all() and (raw[GUARD + n:] == FILL).all())]
        return len(self.blocks), bad


class GuardedTorchMem(TorchMem):
    def __init__(self):
        super().__init__()
        self.blocks = []

    def empty(self, shape, dtype):
        t = self.torch
        shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        raw = t.full((nbytes + 2 * GUARD,), FILL, dtype=t.uint8, device=self.dev)
        user = raw[GUARD:GUARD + nbytes]
        user.zero_()
        self.blocks.append((raw, nbytes, "%s%s" % (np.dtype(dtype).name, shape)))
        return user.view(getattr(t, self._map[np.dtype(dtype)])).reshape(shape)

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        out = self.empty(arr.shape, arr.dtype)
        out.copy_(self.torch.from_numpy(arr))
        return out

    def check(self):
        self.sync()
        bad = []
        for raw, n, name in self.blocks:
            if not (bool((raw[:GUARD] == FILL).all()) and bool((raw[GUARD + n:] == FILL).all())):
                bad.append(name)
        return len(self.blocks), bad
