"""CPU restatement of the reference's frame preprocessing (TEST INFRASTRUCTURE: only tests/, smoke and bench.py's
cpu_baseline leg may import anything under oracle/).

Reference: env.py:27-29  _get_state = cv2.resize(ale.getScreenGrayscale() [210,160] u8, (84, 84), INTER_LINEAR) -> f32 / 255
           env.py:54-69  step(): the observation is the element-wise max of the states after frames 3 and 4 of the repeat.

PARITY UNPINNED: cv2 (OpenCV) is a third-party dependency that is absent from this container and from /root/reference
(requirements.txt:1-5 pins no version), so no golden vector of `cv2.resize` can be generated here.  What follows restates
the PUBLISHED fixed-point algorithm of OpenCV's 8-bit INTER_LINEAR path (modules/imgproc/src/resize.cpp, 3.x / 4.x:
resizeGeneric_ + HResizeLinear<uchar,int,short,2048> + VResizeLinear<uchar,...>, INTER_RESIZE_COEF_BITS = 11):

    scale = src / dst;  f = float((d + 0.5) * scale - 0.5);  s = floor(f);  f -= s
    x only: s < 0 -> (s, f) = (0, 0);  s >= W - 1 -> (s, f) = (W - 1, 0)           [taps clamped to the last column]
    coefficients: short(rint((1 - f) * 2048)), short(rint(f * 2048))                [saturate_cast<short>(float)]
    horizontal:  H[y][dx] = S[y][sx] * a0 + S[y][sx + 1] * a1                        (int, scaled by 2^11)
    vertical:    rows sy, sy + 1 clamped to [0, H - 1];
                 D[dy][dx] = (((b0 * (H[sy0][dx] >> 4)) >> 16) + ((b1 * (H[sy1][dx] >> 4)) >> 16) + 2) >> 2

(the shape of the reference's call — 210x160 -> 84x84, scales 2.5 and 1.904... — takes none of OpenCV's special paths:
INTER_AREA substitution needs both scales == 2, the exact-bit variant is INTER_LINEAR_EXACT).  The device kernel
(rb_frame_preprocess) is tested bit-for-bit against THIS restatement; the restatement itself is pinned only by the
properties in tests/test_frames.py."""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def _taps(dst, src, clamp_f):
    scale = np.float64(src) / np.float64(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)                 # float((dx + 0.5) * scale_x - 0.5)
    s = np.floor(f).astype(np.int64)                                 # cvFloor
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_f:                                                      # the x loop of resize(): border taps collapse
        lo = s < 0
        f[lo], s[lo] = 0.0, 0
        hi = s >= src - 1
        f[hi], s[hi] = 0.0, src - 1
    a0 = np.rint((np.float32(1.0) - f).astype(np.float32) * np.float32(COEF_SCALE)).astype(np.int64)    # saturate_cast<short>
    a1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int64)
    return s, a0, a1


def resize_linear_u8(src, dst_h=84, dst_w=84):
    """cv2.resize(src u8 [H, W], (dst_w, dst_h), interpolation=cv2.INTER_LINEAR) -> u8 [dst_h, dst_w]."""
    src = np.asarray(src)
    assert src.dtype == np.uint8 and src.ndim == 2
    H, W = src.shape
    sx, a0, a1 = _taps(dst_w, W, True)
    sy, b0, b1 = _taps(dst_h, H, False)
    s = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, W - 1)                                  # (a1 == 0 wherever sx + 1 would leave the row)
    hrow = s[:, sx] * a0[None, :] + s[:, sx1] * a1[None, :]          # [H, dst_w], scaled by 2^11
    y0 = np.clip(sy, 0, H - 1)
    y1 = np.clip(sy + 1, 0, H - 1)
    v = ((b0[:, None] * (hrow[y0] >> 4)) >> 16) + ((b1[:, None] * (hrow[y1] >> 4)) >> 16)
    return ((v + 2) >> 2).astype(np.uint8)


def get_state(screen_u8):
    """env.py:27-29: resize, then float32 / 255."""
    return (resize_linear_u8(screen_u8).astype(np.float32) / np.float32(255)).astype(np.float32)


def observe(frame_a, frame_b=None):
    """env.py:52 (reset: one frame) / env.py:57-69 (step: max over the last two frames of the action repeat)."""
    st = get_state(frame_a)
    return st if frame_b is None else np.maximum(st, get_state(frame_b))
