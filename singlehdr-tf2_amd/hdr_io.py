"""Image I/O around the inference path: the file loop of test_real_refinement.py:114-155.

    recon = HdrReconstructor(pipeline.Inference(deq, lin, hal, ref))
    recon.reconstruct_file("scene.jpg", "scene.hdr")          # or recon.reconstruct_dir(in_dir, out_dir)

Per image the reference does: cv2.imread (BGR) -> flip to RGB, /255 -> cubic resize up to a multiple of 64 ->
symmetric pad by 32 -> `bgr2rgb` (a second flip: the networks see BGR-ordered data) -> inference ->
flip, crop the pad, cubic resize back -> cv2.imwrite('.hdr') of the channel-reversed result.  Here only the JPEG
decode (PIL) and the file write are host work; the uint8 image goes to the device once and RGBE bytes (4 B/pixel
instead of 12) come back -- every step in between is a libshdr kernel (csrc/imageio.hip).
cv2 is not installed in this image, so cv2's behaviour is restated: INTER_CUBIC = a -0.75 bicubic
with replicated borders, '.hdr' = Radiance RGBE with adaptive scanline RLE and the `-Y h +X w` orientation.
"""
import ctypes
import glob
import os
import re
import time

import numpy as np
import torch

try:
    from . import _lib
    from . import _ops as K
except ImportError:
    import _lib
    import _ops as K

PADDING = 32          # test_real_refinement.py:135
MULTIPLE = 64         # :129-133


def read_ldr(path):
    """8-bit image file -> uint8 RGB [H, W, 3] (cv2.imread drops alpha and converts grey to 3 channels as well).  Like
    cv2.imread with its default flags (test_real_refinement.py:124) the EXIF Orientation tag is APPLIED: a camera JPEG stored
    rotated comes back upright."""
    from PIL import Image, ImageOps
    with Image.open(path) as im:
        return np.array(ImageOps.exif_transpose(im).convert("RGB"), dtype=np.uint8)          # a writable copy


def rle_encode(rgbe):
    """uint8 [H, W, 4] -> scanline-RLE bytes (libshdr host routine)"""
    rgbe = np.ascontiguousarray(rgbe, dtype=np.uint8)
    h, w, _ = rgbe.shape
    cap = h * (4 + 4 * (w + w // 127 + 2))
    out = np.empty(cap, dtype=np.uint8)
    n = _lib.load().shdr_rgbe_rle_encode(ctypes.c_void_p(rgbe.ctypes.data), w, h, ctypes.c_void_p(out.ctypes.data), cap)
    if n < 0:
        raise RuntimeError("shdr_rgbe_rle_encode: %s" % _lib.load().shdr_last_error().decode())
    return out[:n].tobytes()


def write_hdr(path, rgbe):
    """Radiance picture file from RGBE bytes [H, W, 4]"""
    rgbe = np.asarray(rgbe)
    if rgbe.dtype != np.uint8 or rgbe.ndim != 3 or rgbe.shape[2] != 4:
        raise ValueError("write_hdr: expected uint8 [H, W, 4] RGBE (see _ops.rgbe_encode)")
    h, w, _ = rgbe.shape
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w))
        f.write(rle_encode(rgbe))


def read_hdr(path):
    """Radiance picture file -> float32 RGB [H, W, 3] (flat and RLE scanlines, -Y +X orientation)"""
    with open(path, "rb") as f:
        data = f.read()
    if not data.startswith(b"#?"):
        raise ValueError("%s: not a Radiance file" % path)
    end = data.index(b"\n\n")
    if b"32-bit_rle_rgbe" not in data[:end]:
        raise ValueError("%s: only FORMAT=32-bit_rle_rgbe is supported" % path)
    nl = data.index(b"\n", end + 2)
    m = re.match(rb"-Y (\d+) \+X (\d+)", data[end + 2:nl])
    if not m:
        raise ValueError("%s: unsupported resolution line %r" % (path, data[end + 2:nl]))
    h, w = int(m.group(1)), int(m.group(2))
    buf = np.frombuffer(data, dtype=np.uint8, offset=nl + 1)
    rgbe = np.empty((h, w, 4), dtype=np.uint8)
    pos = 0
    for y in range(h):
        if 8 <= w <= 32767 and buf[pos] == 2 and buf[pos + 1] == 2 and (int(buf[pos + 2]) << 8 | int(buf[pos + 3])) == w:
            pos += 4
            for c in range(4):
                x = 0
                while x < w:
                    n = int(buf[pos])
                    if n > 128:
                        rgbe[y, x:x + n - 128, c] = buf[pos + 1]
                        x += n - 128
                        pos += 2
                    else:
                        rgbe[y, x:x + n, c] = buf[pos + 1:pos + 1 + n]
                        x += n
                        pos += 1 + n
                if x != w:
                    raise ValueError("%s: corrupt scanline %d" % (path, y))
        else:
            rgbe[y] = buf[pos:pos + 4 * w].reshape(w, 4)
            pos += 4 * w
    return rgbe_decode(rgbe)


def rgbe_decode(rgbe):
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e == 0, 0.0, np.ldexp(1.0, e - (128 + 8))).astype(np.float32)
    return rgbe[..., :3].astype(np.float32) * scale[..., None]


class HdrReconstructor:
    """LDR file -> HDR file with the reference tool's geometry (resize to 64x, 32-pixel symmetric pad, crop, resize back)."""

    def __init__(self, inference, padding=PADDING, multiple=MULTIPLE):
        self.inference, self.padding, self.multiple = inference, padding, multiple

    def reconstruct(self, rgb_u8):
        """uint8 RGB [H, W, 3] (host) -> RGBE bytes uint8 [H, W, 4] (host) of the HDR estimate, RGB order"""
        rgb_u8 = np.array(rgb_u8, dtype=np.uint8)               # contiguous, writable (torch.from_numpy)
        h, w, _ = rgb_u8.shape
        dev = torch.device("cuda", torch.cuda.current_device())
        x = K.u8_to_unit(torch.from_numpy(rgb_u8).to(dev, non_blocking=True), False)[None]        # RGB in [0,1]  (:125)
        m = self.multiple
        rh, rw = -(-h // m) * m, -(-w // m) * m
        if (rh, rw) != (h, w):
            x = K.resize_cubic(x, (rh, rw))                                                       # :129-133
        x = K.pad_symmetric(x, self.padding)                                                      # :135-136
        x = K.reverse3(x)                                                                         # tf_utils.bgr2rgb (:141)
        with torch.no_grad():
            y = self.inference(x)                                                                 # :142
        p = self.padding
        y = y[:, p:y.shape[1] - p, p:y.shape[2] - p, :].contiguous()                              # :145 (flip folded below)
        if (rh, rw) != (h, w):
            y = K.resize_cubic(y, (h, w))                                                         # :146-147
        # :144 flips the channels, :150 flips them back and cv2 stores its BGR argument as RGB: the net's channel 0 is the
        # file's blue, i.e. the network output is read as BGR
        return K.rgbe_encode(y[0], reverse_channels=True).cpu().numpy()

    def reconstruct_file(self, ldr_path, hdr_path):
        write_hdr(hdr_path, self.reconstruct(read_ldr(ldr_path)))

    def reconstruct_dir(self, dataset_dir, output_dir, pattern="*.jpg", verbose=True):
        """the `for ldr_img_path in ldr_imgs` loop (:119-151); returns the written paths"""
        os.makedirs(output_dir, exist_ok=True)
        written = []
        for path in sorted(glob.glob(os.path.join(dataset_dir, pattern))):
            start = time.perf_counter()
            out = os.path.join(output_dir, os.path.split(path)[-1].split(".")[0] + ".hdr")          # :148-149
            self.reconstruct_file(path, out)
            written.append(out)
            if verbose:
                print("Spends time : %s seconds" % (time.perf_counter() - start))
        return written
