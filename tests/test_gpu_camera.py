"""GPU tests of the camera-pipeline simulator (csrc/camera.hip) against oracle/camera.py and a real libjpeg (Pillow)."""
import io

import numpy as np
import pytest
import torch

from oracle import camera as O
from oracle import ops as oracle_ops

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def test_expose_matches_restatement_and_is_launch_independent(shdr):
    K = shdr._ops
    rng = np.random.default_rng(0)
    hdr = (rng.random((3, 40, 56, 3)) ** 3 * 4).astype(np.float32)
    t = np.array([0.5, 1.0, 3.0], dtype=np.float32)
    want_t, want_c = O.camera_expose(hdr, t, seed=0x1234567890)
    got_t, got_c = K.camera_expose(dev(hdr), dev(t), 0x1234567890)
    assert np.abs(got_t.cpu().numpy() - want_t).max() <= 2e-6 * want_t.max()          # logf / cosf differ in the last ulp
    assert torch.equal(got_c, got_t.clamp(max=1.0)) and float(got_t.min()) >= 0.0
    again, _ = K.camera_expose(dev(hdr), dev(t), 0x1234567890)
    other, _ = K.camera_expose(dev(hdr), dev(t), 0x1234567891)
    assert torch.equal(again, got_t) and not torch.equal(other, got_t)


@pytest.mark.parametrize("size", [(256, 256), (64, 96), (16, 16)])
def test_jpeg_round_trip_is_bit_exact(shdr, size):
    from PIL import Image
    K = shdr._ops
    h, w = size
    rng = np.random.default_rng(h + w)
    b = 6
    base = np.clip(np.cumsum(rng.normal(size=(b, h, w, 3)), axis=2) * 6 + 128, 0, 255)
    base[1] = rng.integers(0, 256, size=(h, w, 3))                       # white noise
    base[2, : h // 2] = 255                                              # saturated half
    u8 = base.astype(np.uint8)
    ldr = (u8.astype(np.float32) / np.float32(255.0))
    q = [90, 93, 95, 98, 100, 91]
    jpeg, mask = K.jpeg_round_trip(dev(ldr), q)
    got = torch.round(jpeg * 255.0).to(torch.uint8).cpu().numpy()
    assert torch.equal(jpeg, torch.from_numpy(got.astype(np.float32) / np.float32(255.0)).cuda())       # exactly u8 / 255
    for i in range(b):
        assert np.array_equal(got[i], O.jpeg_round_trip(u8[i], q[i])), ("oracle", i)
        buf = io.BytesIO()
        Image.fromarray(u8[i]).save(buf, format="JPEG", quality=q[i], subsampling=2)
        assert np.array_equal(got[i], np.array(Image.open(io.BytesIO(buf.getvalue())).convert("RGB"))), ("libjpeg", i)
    assert torch.equal(mask.cpu(), torch.from_numpy(O.loss_mask(got)))
    with pytest.raises(RuntimeError, match="MCU"):
        K.jpeg_round_trip(dev(ldr[:, : h - 8] if h > 16 else ldr[:, :8]), q)


def test_loss_mask_excludes_extreme_exposures(shdr):
    K = shdr._ops
    ldr = np.full((3, 256, 256, 3), 0.5, dtype=np.float32)
    ldr[0] = 1.0
    ldr[1] = 0.0
    ldr[2, :120] = 0.0
    jpeg, mask = K.jpeg_round_trip(dev(ldr), [95, 95, 95])
    assert mask.reshape(-1).tolist() == [0.0, 0.0, 1.0]
    got = torch.round(jpeg * 255.0).to(torch.uint8).cpu().numpy()
    assert torch.equal(mask.cpu(), torch.from_numpy(O.loss_mask(got)))


def test_camera_pipeline_returns_the_five_tensors(shdr, emor_table):
    rng = np.random.default_rng(3)
    b = 4
    hdr = (rng.random((b, 64, 64, 3)) ** 2 * 2).astype(np.float32)
    crf = np.sort(rng.random((b, 1024)).astype(np.float32), axis=1)
    crf = (crf - crf[:, :1]) / (crf[:, -1:] - crf[:, :1])
    t = np.array([0.7, 1.0, 1.5, 2.0], dtype=np.float32)
    cam = shdr.camera.CameraPipeline(seed=11)
    ldr, jpeg, clipped, hdr_t, mask = cam(dev(hdr), dev(crf), dev(t))
    want_t, want_c = O.camera_expose(hdr, t, seed=11)
    assert np.abs(hdr_t.cpu().numpy() - want_t).max() <= 2e-6 * want_t.max()
    c = clipped.cpu().numpy()
    assert np.abs(ldr.cpu().numpy() - oracle_ops.apply_rf(c, crf)).max() <= 1e-6
    u8 = np.round(ldr.cpu().numpy() * 255.0).astype(np.uint8)
    want = np.stack([O.jpeg_round_trip(u8[i], q) for i, q in enumerate(shdr.camera.jpeg_qualities(b))])
    assert np.array_equal(torch.round(jpeg * 255.0).to(torch.uint8).cpu().numpy(), want)
    assert mask.shape == (b, 1, 1, 1) and jpeg.shape == ldr.shape == clipped.shape == hdr_t.shape == (b, 64, 64, 3)
    ldr2 = cam(dev(hdr), dev(crf), dev(t))[0]
    assert not torch.equal(ldr2, ldr)                                    # a fresh noise field on every call
