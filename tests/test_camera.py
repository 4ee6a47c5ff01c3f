"""Camera-pipeline simulator (joint_training.py:26-69): the JPEG restatement against a REAL libjpeg (Pillow), the
counter-based generator against its published vectors, and the host-side helpers.  CPU only."""
import ctypes
import importlib
import io

import numpy as np
import pytest

from oracle import camera as O

pkg = importlib.import_module("singlehdr-tf2_amd")


def pil_round_trip(img, q):
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, format="JPEG", quality=q, subsampling=2)          # 4:2:0 like tf.io.encode_jpeg's default
    return np.array(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))


def images():
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:128, 0:128]
    yield "blocks+noise", np.clip(rng.random((8, 8, 3)).repeat(16, 0).repeat(16, 1) * 255 + rng.normal(size=(128, 128, 3)) * 12,
                                  0, 255).astype(np.uint8)
    yield "smooth", np.stack([(np.sin(xx / 9.0) * 0.5 + 0.5) * 255, (np.cos(yy / 13.0) * 0.5 + 0.5) * 255,
                              (xx + yy) % 256], -1).astype(np.uint8)
    yield "white noise", rng.integers(0, 256, size=(64, 96, 3), dtype=np.uint8)
    yield "saturated", np.where(rng.random((48, 32, 3)) > 0.5, 255, 0).astype(np.uint8)
    yield "training crop", np.clip(np.cumsum(rng.normal(size=(256, 256, 3)), axis=1) * 6 + 128, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("q", [90, 91, 93, 95, 97, 99, 100, 75, 30])
def test_jpeg_restatement_is_bit_exact_against_libjpeg(q):
    for name, img in images():
        if name == "training crop" and q not in (90, 95, 100):
            continue
        assert np.array_equal(O.jpeg_round_trip(img, q), pil_round_trip(img, q)), (name, q)


def test_quality_scaling_and_sample_qualities():
    ql, qc = O.quant_tables(90)
    assert ql[0].tolist() == [3, 2, 2, 3, 5, 8, 10, 12] and qc[0].tolist() == [3, 4, 5, 9, 20, 20, 20, 20]
    assert np.all(O.quant_tables(100)[0] == 1) and np.array_equal(O.quant_tables(50)[0], O.LUMA_Q)
    assert pkg.camera.jpeg_qualities(16) == [90, 91, 91, 92, 93, 93, 94, 95, 95, 96, 97, 97, 98, 99, 99, 100]
    assert [O.jpeg_quality_of_sample(i, 16) for i in range(16)] == pkg.camera.jpeg_qualities(16)
    assert pkg.camera.jpeg_qualities(1) == [90] and pkg.camera.jpeg_qualities(2) == [90, 100]


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32 with 10 rounds"""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    lib = pkg._lib.load()
    for ctr, key, want in kat:
        assert tuple(int(v) for v in O.philox4x32_10(np.array([ctr], dtype=np.uint32), key)[0]) == want
        c, k, out = (ctypes.c_uint32 * 4)(*ctr), (ctypes.c_uint32 * 2)(*key), (ctypes.c_uint32 * 4)()
        assert lib.shdr_philox4x32_10(c, k, out) == 0 and tuple(out) == want


def test_noise_restatement_statistics():
    hdr = np.full((4, 64, 64, 3), 0.5, dtype=np.float32)
    t = np.array([1.0, 2.0, 0.5, 1.0], dtype=np.float32)
    a, clipped = O.camera_expose(hdr, t, seed=7)
    b, _ = O.camera_expose(hdr, t, seed=7)
    c, _ = O.camera_expose(hdr, t, seed=8)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert a.min() >= 0 and np.array_equal(clipped, np.minimum(a, 1))
    x = 0.5 * t.reshape(4, 1, 1, 1)
    resid = (a - x).reshape(4, -1, 3)
    assert np.all(np.abs(resid.mean(axis=1)) < 4e-4)                     # zero-mean noise
    sd = resid.std(axis=1)                                               # per (sample, channel): sqrt((sigma_s x)^2 + sigma_c^2)
    assert np.all(sd <= np.sqrt((0.08 / 6 * x.reshape(4, 1)) ** 2 + 0.005 ** 2) * 1.05) and sd.max() > 1e-3
    assert len(np.unique(np.round(sd, 6))) == 12                         # a different level for every sample and channel


def test_loss_mask_rule():
    img = np.zeros((3, 256, 256, 3), dtype=np.uint8)
    img[0] = 255                                                         # over-exposed everywhere
    img[1, :100] = 255                                                   # 39 % bright, rest dark (<= 6): under-exposed > half
    img[2, :, :] = 128
    img[2, :120] = 3                                                     # 47 % dark: kept
    assert O.loss_mask(img).reshape(-1).tolist() == [0.0, 0.0, 1.0]
    g = O.rgb_to_gray_u8(np.array([[[255, 255, 255], [249, 249, 249], [6, 6, 6], [255, 0, 0]]], dtype=np.uint8))
    assert g.tolist() == [[255, 249, 6, 76]]
