"""GPU tests of the inference tool's image plumbing (csrc/imageio.hip) and of the file loop that uses it."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import imageio as O
from oracle import nets

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def test_u8_to_unit_and_symmetric_pad_exact(shdr):
    K = shdr._ops
    rng = np.random.default_rng(0)
    u8 = rng.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    want = u8.astype(np.float32) / np.float32(255.0)
    assert np.array_equal(K.u8_to_unit(dev(u8)).cpu().numpy(), want)
    assert np.array_equal(K.u8_to_unit(dev(u8), True).cpu().numpy(), want[..., ::-1])
    x = rng.random((2, 9, 7, 3)).astype(np.float32)
    for pad in (0, 1, 5, 7):
        assert np.array_equal(K.pad_symmetric(dev(x), pad).cpu().numpy(), O.pad_symmetric(x, pad))
    with pytest.raises(RuntimeError, match="pad"):
        K.pad_symmetric(dev(x), 8)                          # numpy's single-reflection rule needs pad <= size


@pytest.mark.parametrize("shape", [((1, 100, 90, 3), (128, 128)), ((2, 64, 64, 3), (33, 47)), ((1, 17, 200, 3), (64, 256)),
                                   ((1, 128, 128, 3), (100, 90)), ((1, 5, 6, 4), (5, 6))])
def test_resize_cubic_matches_restatement(shdr, shape):
    (n, h, w, c), out_hw = shape
    x = np.random.default_rng(h * w).random((n, h, w, c)).astype(np.float32)
    got = shdr._ops.resize_cubic(dev(x), out_hw).cpu().numpy()
    assert got.shape == (n,) + out_hw + (c,)
    assert rel_err(got, O.resize_cubic(x, out_hw)) <= 2e-6


def test_rgbe_encode_bytes_match_restatement(shdr):
    rng = np.random.default_rng(3)
    x = (rng.random((64, 50, 3)) * np.exp(rng.normal(size=(64, 50, 1)) * 4)).astype(np.float32)
    x[0, 0] = 0
    x[0, 1] = [1.0, 0.5, 0.25]
    x[0, 2] = [-1.0, 2.0, 1e-35]
    x[0, 3] = [1e-33, 0, 0]
    got = shdr._ops.rgbe_encode(dev(x)).cpu().numpy()
    assert got.dtype == np.uint8 and got[0, 1].tolist() == [128, 64, 32, 129]
    assert np.array_equal(got, O.rgbe_encode(x))
    assert np.array_equal(shdr._ops.rgbe_encode(dev(x), True).cpu().numpy(), O.rgbe_encode(x[..., ::-1]))


def test_file_loop_matches_the_oracle_chain(shdr, emor_table, tmp_path):
    """JPEG -> .hdr through HdrReconstructor vs the same geometry restated on the CPU oracle (networks included)."""
    from PIL import Image
    rng = np.random.default_rng(5)
    P = {k: nets.init_params(getattr(nets, k + "_spec")(), 40 + i) for i, k in enumerate(("deq", "lin", "hal", "ref"))}
    mods = dict(deq="dequantization_net", lin="linearization_net", hal="hallucination_net", ref="refinement_net")
    ms = {k: getattr(shdr, mods[k]).model().load_numpy(P[k]) for k in mods}
    recon = shdr.hdr_io.HdrReconstructor(shdr.pipeline.Inference(ms["deq"], ms["lin"], ms["hal"], ms["ref"]))
    base = np.clip(rng.random((8, 7, 3)).repeat(10, axis=0).repeat(10, axis=1) + rng.normal(size=(80, 70, 3)) * 0.03, 0, 1)
    base[:20, :20] = 1.0                                    # a saturated region: alpha = 1 there
    src = tmp_path / "in"
    src.mkdir()
    Image.fromarray((base * 255).astype(np.uint8)).save(str(src / "scene.01.jpg"), quality=95)
    Image.fromarray((base[:64, :64] * 255).astype(np.uint8)).save(str(src / "aligned.jpg"), quality=95)
    written = recon.reconstruct_dir(str(src), str(tmp_path / "out"), verbose=False)
    assert [p.split("/")[-1] for p in written] == ["aligned.hdr", "scene.hdr"]          # name up to the FIRST dot (:148-149)
    for name, path in (("scene.01.jpg", written[1]), ("aligned.jpg", written[0])):
        rgb = shdr.hdr_io.read_ldr(str(src / name))
        h, w, _ = rgb.shape
        x = (rgb.astype(np.float32) / np.float32(255.0))[None]
        rh, rw = -(-h // 64) * 64, -(-w // 64) * 64
        if (rh, rw) != (h, w):
            x = O.resize_cubic(x, (rh, rw))
        x = O.pad_symmetric(x, 32)[..., ::-1]
        y = nets.inference(P, np.ascontiguousarray(x), emor_table)["hdr"][:, 32:-32, 32:-32]
        if (rh, rw) != (h, w):
            y = O.resize_cubic(y.astype(np.float32), (h, w))
        want = np.maximum(y[0][..., ::-1], 0)               # the file stores the network's channel 2 as red
        got = shdr.hdr_io.read_hdr(path)
        assert got.shape == (h, w, 3)
        step = want.max(axis=-1, keepdims=True) / 128 + 1e-6                          # one RGBE mantissa unit per pixel
        assert np.all(np.abs(got - want) <= 1.01 * step + 1e-4 * want.max())
