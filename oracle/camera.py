"""CPU restatement of the camera-pipeline simulator of joint_training.py:26-69 -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

`jpeg_round_trip` restates what tf.image.adjust_jpeg_quality(img, q) does to an 8-bit RGB image -- libjpeg baseline
encode (JFIF YCbCr, 4:2:0 chroma, Annex-K tables scaled by the IJG quality rule, "islow" integer DCT) followed by
decode (islow IDCT, fancy chroma upsampling) -- in libjpeg's own integer arithmetic.  It is PINNED against a real
libjpeg: Pillow (libjpeg-turbo) is installed in this image, and tests/test_camera.py checks the restatement BIT FOR BIT
against Image.save(quality=q, subsampling=2) -> Image.open on the same pixels.  (TensorFlow links the same library;
tf.image.adjust_jpeg_quality = encode_jpeg(quality, chroma_downsampling=True) + decode_jpeg(fancy_upscaling=True).)
"""
import numpy as np

# ITU T.81 Annex K.1 / K.2 (the tables libjpeg's jpeg_set_quality scales)
LUMA_Q = np.array([16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                   14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                   49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99], dtype=np.int32).reshape(8, 8)
CHROMA_Q = np.array([17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                     47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32, dtype=np.int32).reshape(8, 8)


def quant_tables(quality):
    """jpeg_quality_scaling + jpeg_add_quant_table(force_baseline=TRUE)"""
    q = min(max(int(quality), 1), 100)
    scale = 5000 // q if q < 50 else 200 - 2 * q
    return tuple(np.clip((t * scale + 50) // 100, 1, 255).astype(np.int32) for t in (LUMA_Q, CHROMA_Q))


# ---- libjpeg's "islow" integer DCT pair (jfdctint.c / jidctint.c: Loeffler-Ligtenberg-Moschytz, CONST_BITS = 13,
#      PASS1_BITS = 2), restated on int64 arrays; all shifts are arithmetic ----------------------------------------------
CONST_BITS, PASS1_BITS = 13, 2
F_0_298631336, F_0_390180644, F_0_541196100, F_0_765366865 = 2446, 3196, 4433, 6270
F_0_899976223, F_1_175875602, F_1_501321110, F_1_847759065 = 7373, 9633, 12299, 15137
F_1_961570560, F_2_053119869, F_2_562915447, F_3_072711026 = 16069, 16819, 20995, 25172


def _descale(x, n):
    return (x + (1 << (n - 1))) >> n


def _fdct_pass(d, first):
    """one 1-D pass along the LAST axis of d [..., 8] (int64)"""
    t0, t7 = d[..., 0] + d[..., 7], d[..., 0] - d[..., 7]
    t1, t6 = d[..., 1] + d[..., 6], d[..., 1] - d[..., 6]
    t2, t5 = d[..., 2] + d[..., 5], d[..., 2] - d[..., 5]
    t3, t4 = d[..., 3] + d[..., 4], d[..., 3] - d[..., 4]
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    out = [None] * 8
    if first:
        out[0], out[4] = (t10 + t11) << PASS1_BITS, (t10 - t11) << PASS1_BITS
        sh = CONST_BITS - PASS1_BITS
    else:
        out[0], out[4] = _descale(t10 + t11, PASS1_BITS), _descale(t10 - t11, PASS1_BITS)
        sh = CONST_BITS + PASS1_BITS
    z1 = (t12 + t13) * F_0_541196100
    out[2] = _descale(z1 + t13 * F_0_765366865, sh)
    out[6] = _descale(z1 - t12 * F_1_847759065, sh)
    z1, z2, z3, z4 = t4 + t7, t5 + t6, t4 + t6, t5 + t7
    z5 = (z3 + z4) * F_1_175875602
    t4, t5, t6, t7 = t4 * F_0_298631336, t5 * F_2_053119869, t6 * F_3_072711026, t7 * F_1_501321110
    z1, z2 = -z1 * F_0_899976223, -z2 * F_2_562915447
    z3, z4 = -z3 * F_1_961570560 + z5, -z4 * F_0_390180644 + z5
    out[7], out[5] = _descale(t4 + z1 + z3, sh), _descale(t5 + z2 + z4, sh)
    out[3], out[1] = _descale(t6 + z2 + z3, sh), _descale(t7 + z1 + z4, sh)
    return np.stack(out, axis=-1)


def _idct_pass(c, first):
    """one 1-D inverse pass along the LAST axis of c [..., 8] (int64)"""
    z2, z3 = c[..., 2], c[..., 6]
    z1 = (z2 + z3) * F_0_541196100
    t2, t3 = z1 - z3 * F_1_847759065, z1 + z2 * F_0_765366865
    t0, t1 = (c[..., 0] + c[..., 4]) << CONST_BITS, (c[..., 0] - c[..., 4]) << CONST_BITS
    t10, t13, t11, t12 = t0 + t3, t0 - t3, t1 + t2, t1 - t2
    t0, t1, t2, t3 = c[..., 7], c[..., 5], c[..., 3], c[..., 1]
    z1, z2, z3, z4 = t0 + t3, t1 + t2, t0 + t2, t1 + t3
    z5 = (z3 + z4) * F_1_175875602
    t0, t1, t2, t3 = t0 * F_0_298631336, t1 * F_2_053119869, t2 * F_3_072711026, t3 * F_1_501321110
    z1, z2 = -z1 * F_0_899976223, -z2 * F_2_562915447
    z3, z4 = -z3 * F_1_961570560 + z5, -z4 * F_0_390180644 + z5
    t0, t1, t2, t3 = t0 + z1 + z3, t1 + z2 + z4, t2 + z2 + z3, t3 + z1 + z4
    sh = CONST_BITS - PASS1_BITS if first else CONST_BITS + PASS1_BITS + 3
    out = [t10 + t3, t11 + t2, t12 + t1, t13 + t0, t13 - t0, t12 - t1, t11 - t2, t10 - t3]
    return np.stack([_descale(o, sh) for o in out], axis=-1)


def _blocks(plane):           # [H, W] -> [H/8, W/8, 8 (row), 8 (col)]
    h, w = plane.shape
    return plane.reshape(h // 8, 8, w // 8, 8).transpose(0, 2, 1, 3)


def _unblocks(b):
    nh, nw = b.shape[:2]
    return b.transpose(0, 2, 1, 3).reshape(nh * 8, nw * 8)


def _code_plane(plane, q):
    """level shift, forward DCT (rows then columns; output scaled by 8), quantise as jcdctmgr.c does
    (|c| + 4q) // 8q with the sign restored), dequantise, inverse DCT (columns then rows), +128, range limit"""
    b = _blocks(plane.astype(np.int64) - 128)
    coef = _fdct_pass(b, True)                                          # along columns index = within a row
    coef = _fdct_pass(coef.swapaxes(-1, -2), False).swapaxes(-1, -2)    # along rows index = down a column
    q8 = (q.astype(np.int64) * 8)[None, None]
    quant = np.sign(coef) * ((np.abs(coef) + (q8 >> 1)) // q8)
    deq = quant * q.astype(np.int64)[None, None]
    ws = _idct_pass(deq.swapaxes(-1, -2), True).swapaxes(-1, -2)        # pass 1 works down the columns
    rec = _idct_pass(ws, False)                                         # pass 2 along the rows
    return np.clip(_unblocks(rec) + 128, 0, 255)


def _fix16(x):
    return int(x * 65536 + 0.5)


def jpeg_round_trip(rgb_u8, quality):
    """uint8 [H, W, 3] with H % 16 == W % 16 == 0 -> uint8 [H, W, 3] after a quality-`quality` baseline JPEG round trip
    (4:2:0 chroma, islow DCT, fancy upsampling): bit-exact w.r.t. libjpeg / libjpeg-turbo"""
    rgb = np.asarray(rgb_u8).astype(np.int64)
    h, w, _ = rgb.shape
    assert h % 16 == 0 and w % 16 == 0, "whole 16x16 MCUs only (the training crops are 256x256)"
    r, g, b = rgb[..., 0], rgb[..., 1], rgb[..., 2]
    half, off = 1 << 15, 128 << 16
    # jccolor.c rgb_ycc_convert
    y = (_fix16(0.29900) * r + _fix16(0.58700) * g + _fix16(0.11400) * b + half) >> 16
    cb = (-_fix16(0.16874) * r - _fix16(0.33126) * g + _fix16(0.50000) * b + off + half - 1) >> 16
    cr = (_fix16(0.50000) * r - _fix16(0.41869) * g - _fix16(0.08131) * b + off + half - 1) >> 16

    def down(p):              # jcsample.c h2v2_downsample: (a+b+c+d + bias) >> 2, bias alternating 1, 2 along a row
        s = p[0::2, 0::2] + p[0::2, 1::2] + p[1::2, 0::2] + p[1::2, 1::2]
        return (s + (1 + (np.arange(w // 2) & 1))[None, :]) >> 2

    ql, qc = quant_tables(quality)
    y2 = _code_plane(y, ql)
    cb2 = _code_plane(down(cb), qc)
    cr2 = _code_plane(down(cr), qc)

    def up(p):                # jdsample.c h2v2_fancy_upsample: triangle filter 9:3:3:1, biases 8 / 7, >> 4
        ph, pw = p.shape
        pad = np.pad(p, 1, mode="edge")
        out = np.empty((2 * ph, 2 * pw), dtype=np.int64)
        for dy in (0, 1):
            near = pad[1:-1]
            far = pad[0:-2] if dy == 0 else pad[2:]
            col = 3 * near + far                                   # [ph, pw + 2] column sums ("thiscolsum")
            out[dy::2, 0::2] = (3 * col[:, 1:-1] + col[:, 0:-2] + 8) >> 4
            out[dy::2, 1::2] = (3 * col[:, 1:-1] + col[:, 2:] + 7) >> 4
        return out

    cbu, cru = up(cb2) - 128, up(cr2) - 128
    # jdcolor.c build_ycc_rgb_table / ycc_rgb_convert
    rr = y2 + ((_fix16(1.40200) * cru + half) >> 16)
    gg = y2 + ((-_fix16(0.34414) * cbu + half - _fix16(0.71414) * cru) >> 16)
    bb = y2 + ((_fix16(1.77200) * cbu + half) >> 16)
    return np.clip(np.stack([rr, gg, bb], axis=-1), 0, 255).astype(np.uint8)


def rgb_to_gray_u8(rgb_u8):
    """tf.image.rgb_to_grayscale on uint8: float in [0,1], weights (0.2989, 0.5870, 0.1140), back to uint8 by
    saturate_cast(x * 255.5) (convert_image_dtype)"""
    f = np.asarray(rgb_u8).astype(np.float32) * (np.float32(1.0) / np.float32(255.0))    # convert_image_dtype multiplies by 1/max
    g = f[..., 0] * np.float32(0.2989) + f[..., 1] * np.float32(0.5870) + f[..., 2] * np.float32(0.1140)
    return np.clip(np.floor(g * np.float32(255.5)), 0, 255).astype(np.uint8)


def loss_mask(jpeg_u8):
    """joint_training.py:54-63: 0 for a sample where more than half of 256*256 pixels are >= 249 or <= 6 grey levels"""
    gray = rgb_to_gray_u8(jpeg_u8)                                 # [b, h, w]
    over = (gray >= 249).sum(axis=(1, 2)) > 256.0 * 256.0 * 0.5
    under = (gray <= 6).sum(axis=(1, 2)) > 256.0 * 256.0 * 0.5
    return (~(over | under)).astype(np.float32).reshape(-1, 1, 1, 1)


def jpeg_quality_of_sample(i, batch_size):
    """joint_training.py:48: int(round(i / (BATCH_SIZE - 1) * 10 + 90)) -- Python's round (half to even)"""
    return int(round(float(i) / float(batch_size - 1) * 10.0 + 90.0)) if batch_size > 1 else 90


# ---- exposure + noise (joint_training.py:30-43) with the SAME counter-based random stream as csrc/camera.hip ----------
def philox4x32_10(counter, key):
    """counter [..., 4] uint32, key (k0, k1) -> [..., 4] uint32   (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3")"""
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    m32 = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & m32, p1 & m32, ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & m32, p0 & m32]
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return np.stack(c, axis=-1).astype(np.uint32)


def camera_expose(hdr, t, seed):
    """(hdr_t, clipped) in float32, op for op as the kernel evaluates them"""
    hdr = np.asarray(hdr, dtype=np.float32)
    n, h, w, _ = hdr.shape
    key = (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    f32 = np.float32
    e = np.arange(hdr.size, dtype=np.uint64)
    ctr = np.stack([e & np.uint64(0xFFFFFFFF), e >> np.uint64(32), np.zeros_like(e), np.zeros_like(e)], axis=-1)
    r = philox4x32_10(ctr, key)
    u1 = (r[:, 0] >> 8).astype(f32) * f32(2.0 ** -24) + f32(2.0 ** -25)
    u2 = (r[:, 1] >> 8).astype(f32) * f32(2.0 ** -24) + f32(2.0 ** -25)
    rad = np.sqrt(f32(-2.0) * np.log(u1, dtype=f32), dtype=f32)
    ang = f32(6.283185307179586) * u2
    z0 = (rad * np.cos(ang, dtype=f32)).reshape(hdr.shape)
    z1 = (rad * np.sin(ang, dtype=f32)).reshape(hdr.shape)
    sc = np.arange(n * 3, dtype=np.uint64)
    s = philox4x32_10(np.stack([sc, np.zeros_like(sc), np.zeros_like(sc), np.ones_like(sc)], axis=-1), key)
    sigma_s = (f32(0.08 / 6.0) * ((s[:, 0] >> 8).astype(f32) * f32(2.0 ** -24))).reshape(n, 1, 1, 3)
    sigma_c = (f32(0.005) * ((s[:, 1] >> 8).astype(f32) * f32(2.0 ** -24))).reshape(n, 1, 1, 3)
    x = hdr * np.asarray(t, dtype=f32).reshape(n, 1, 1, 1)
    v = x + z0 * (sigma_s * x)
    v = v + sigma_c * z1
    v = np.maximum(v, f32(0))
    return v, np.minimum(v, f32(1))
