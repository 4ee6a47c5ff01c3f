"""CPU restatement of the inference tool's image plumbing (test_real_refinement.py:119-155) -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.  cv2 is not installed
here, so `resize_cubic` restates OpenCV's INTER_CUBIC from its published definition (resize.cpp: interpolateCubic with
A = -0.75, source coordinate (d + 0.5) * scale - 0.5, replicated border) and `rgbe_encode` restates Ward's RGBE pixel
conversion as cv2.imwrite('.hdr') applies it: parity unpinned (no cv2-written fixture exists in the reference).
"""
import numpy as np


def _cubic_weights(n_in, n_out):
    scale = np.float32(np.float64(n_in) / n_out)
    f = (np.arange(n_out, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
    i0 = np.floor(f).astype(np.int64)
    t = (f - i0.astype(np.float32)).astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1.0)
    c0 = ((A * (t + one) - np.float32(5) * A) * (t + one) + np.float32(8) * A) * (t + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * t - (A + np.float32(3))) * t * t + one
    c2 = ((A + np.float32(2)) * (one - t) - (A + np.float32(3))) * (one - t) * (one - t) + one
    c3 = one - c0 - c1 - c2
    idx = np.clip(i0[:, None] + np.arange(-1, 3)[None, :], 0, n_in - 1)
    return idx, np.stack([c0, c1, c2, c3], axis=1).astype(np.float32)


def resize_cubic(x, out_hw):
    """x [N,H,W,C] float32 -> [N,Ho,Wo,C]; rows of 4 horizontal taps first, then the 4 vertical taps (kernel order)"""
    x = np.asarray(x, dtype=np.float32)
    n, h, w, c = x.shape
    ho, wo = out_hw
    iy, cy = _cubic_weights(h, ho)
    ix, cx = _cubic_weights(w, wo)
    out = np.zeros((n, ho, wo, c), dtype=np.float32)
    for i in range(4):
        rows = x[:, iy[:, i]]                                    # [n, ho, w, c]
        acc = np.zeros((n, ho, wo, c), dtype=np.float32)
        for j in range(4):
            acc += cx[None, None, :, j, None] * rows[:, :, ix[:, j]]
        out += cy[None, :, i, None, None] * acc
    return out


def pad_symmetric(x, pad):
    return np.pad(x, ((0, 0), (pad, pad), (pad, pad), (0, 0)), "symmetric")


def rgbe_encode(rgb):
    rgb = np.maximum(np.asarray(rgb, dtype=np.float32), 0)
    v = rgb.max(axis=-1)
    m, e = np.frexp(v)
    with np.errstate(divide="ignore", invalid="ignore"):
        s = (m.astype(np.float32) * np.float32(256.0) / v).astype(np.float32)
    out = np.zeros(rgb.shape[:-1] + (4,), dtype=np.uint8)
    ok = v >= 1e-32
    out[..., :3] = np.where(ok[..., None], (rgb * np.where(ok, s, 0)[..., None]).astype(np.uint8), 0)
    out[..., 3] = np.where(ok, e + 128, 0).astype(np.uint8)
    return out
