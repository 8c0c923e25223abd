"""Frame preprocessing of the reference's environment wrapper on the device.

env.py:27-29   `_get_state`: cv2.resize(ale.getScreenGrayscale(), (84, 84), INTER_LINEAR) -> float32 / 255
env.py:57-69   `step`: the observation is the element-wise max of the states after frames 3 and 4 of the action repeat

`FramePreprocessor.observe(frame_a, frame_b)` takes the raw u8 grayscale screens (torch uint8 tensors on the device, or
numpy arrays / CPU tensors, which are uploaded as 33 KB of bytes instead of 28 KB of float32 per state after a host-side
resize) and returns the float32 [84, 84] observation on the device, ready for the state deque of env.py:25,70 and for
`ReplayMemory.append`.  The resize is OpenCV's fixed-point 8-bit INTER_LINEAR restated (include/rainbow_hip.h
rb_frame_preprocess): parity with cv2 itself is UNPINNED because cv2 is absent from the build container."""
import numpy as np
import torch

from . import _lib as L


class FramePreprocessor:
    def __init__(self, device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("rainbow_amd.frames runs on MI355X: device must be a cuda (ROCm) device, got %s" % self.device)
        self._lib = L.load()

    def _dev_u8(self, f):
        if f is None:
            return None
        t = torch.from_numpy(np.ascontiguousarray(f)) if isinstance(f, np.ndarray) else f
        if t.dtype != torch.uint8:
            raise TypeError("raw frames are uint8 grayscale screens (ale.getScreenGrayscale()), got %s" % t.dtype)
        return t.to(self.device).contiguous()

    def observe(self, frame_a, frame_b=None):
        """[H, W] u8 (+ optional second frame) -> float32 [84, 84];  [n, H, W] batches -> [n, 84, 84] (vectorised actors)."""
        a, b = self._dev_u8(frame_a), self._dev_u8(frame_b)
        if b is not None and b.shape != a.shape:
            raise ValueError("the two frames of a max-pool pair must have the same shape")
        batched = a.dim() == 3
        if a.dim() not in (2, 3):
            raise ValueError("frames are [H, W] or [n, H, W]")
        n = int(a.shape[0]) if batched else 1
        H, W = int(a.shape[-2]), int(a.shape[-1])
        out = torch.empty((n, 84, 84), dtype=torch.float32, device=self.device)
        L.check(self._lib, self._lib.rb_frame_preprocess(a.data_ptr(), b.data_ptr() if b is not None else None, H, W, n,
                                                         out.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))
        self._keep = (a, b)            # inputs stay alive until the stream has consumed them
        return out if batched else out[0]
