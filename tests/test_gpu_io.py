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


def test_flip_rot90_matches_tf_semantics(shdr):
    K = shdr._ops
    rng = np.random.default_rng(9)
    x = rng.random((10, 12, 12, 3)).astype(np.float32) * 255
    flip = np.array([0, 1, 0, 1, 0, 1, 0, 1, 0, 1], dtype=np.int32)
    rot = np.array([0, 0, 1, 1, 2, 2, 3, 3, 4, 4], dtype=np.int32)       # k = int(u*4 + 0.5) reaches 4 (= 0)
    got = K.flip_rot90(dev(x), dev(flip), dev(rot), 255.0).cpu().numpy()
    for i in range(10):
        img = x[i][:, ::-1] if flip[i] else x[i]                          # tf.image.flip_left_right
        want = np.rot90(img, int(rot[i]), axes=(0, 1)) / np.float32(255.0)   # tf.image.rot90: counter-clockwise
        assert np.array_equal(got[i], want), i


def test_hdr_real_dataset_iterates_device_batches(shdr, tmp_path):
    """records written in the layout of convert_to_tf_record.py:62-66, read back as finetune_real_dataset.py:34-78 does"""
    T = shdr.tfrecord
    rng = np.random.default_rng(10)
    patches = [(rng.random((256, 256, 3)).astype(np.float32) * 255, (rng.random((256, 256, 3)) ** 3 * 40).astype(np.float32))
               for _ in range(6)]
    for f in range(2):
        T.write_records(str(tmp_path / ("train_64_%04d.tfrecords" % f)),
                        [T.make_example({"ref_HDR": h.tobytes(), "ref_LDR": l.tobytes()}) for l, h in patches[3 * f:3 * f + 3]])
    plain = list(T.HdrRealDataset(str(tmp_path), batch_size=4, augment=False, shuffle_buffer=0))
    assert [tuple(b[0].shape) for b in plain] == [(4, 256, 256, 3), (2, 256, 256, 3)]             # drop_remainder=False
    ldr = torch.cat([b[0] for b in plain]).cpu().numpy()
    hdr = torch.cat([b[1] for b in plain]).cpu().numpy()
    for i, (l, h) in enumerate(patches):
        assert np.array_equal(ldr[i], l / np.float32(255.0))
        want = h / (np.float32(1e-6) + h.mean(dtype=np.float64).astype(np.float32)) * np.float32(0.5)
        assert np.abs(hdr[i] - want).max() <= 2e-6 * want.max()
    aug = list(T.HdrRealDataset(str(tmp_path), batch_size=4, seed=3))
    assert sum(b[0].shape[0] for b in aug) == 6
    a_ldr = torch.cat([b[0] for b in aug]).cpu().numpy()
    # every augmented patch is one of the 8 flips / rotations of exactly one source patch
    for img in a_ldr:
        hits = 0
        for l, _ in patches:
            base = l / np.float32(255.0)
            hits += any(np.array_equal(img, np.rot90(b, k, axes=(0, 1))) for b in (base, base[:, ::-1]) for k in range(4))
        assert hits == 1
