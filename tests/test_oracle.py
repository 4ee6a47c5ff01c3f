"""CPU tests that pin the oracle (no GPU, no product code).

Pins, per SURVEY.md section 8c: (i) the reference's only known-answer example
(figure/lin2.png), (ii) the invemor.txt check-sums, (iii) hand-computable index
tests of the TF padding / pooling / sobel conventions, (iv) torch-CPU
cross-checks where torch semantics coincide with TF's, plus the committed
golden vectors (tests/golden, made by tools/make_golden.py).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, quantised_image
from oracle import nets, ops


# ---- (i) the reference's worked example: figure/lin2.png ---------------------
def test_soft_histogram_known_answer_lin2():
    x = np.array([0.63, 0.65, 0.32, 0.84, 0.15]).reshape(1, 1, 5, 1)
    h = ops.histogram_layer(x, 5)[0, 0]          # [5 pixels, 5 bins]
    expect = np.array([[0, 0, .35, .65, 0], [0, 0, .25, .75, 0], [0, .9, .1, 0, 0],
                       [0, 0, 0, .3, .7], [.75, .25, 0, 0, 0]])
    np.testing.assert_allclose(h, expect, atol=1e-12)


@pytest.mark.parametrize("B", [4, 8, 16, 32])
def test_soft_histogram_properties(B):
    rng = np.random.default_rng(B)
    x = rng.random((2, 9, 7, 3)).astype(np.float32)
    h = ops.histogram_layer(x, B)
    assert h.shape == (2, 9, 7, 3 * B) and h.dtype == np.float32
    hb = h.reshape(2, 9, 7, B, 3)
    assert ((hb > 0).sum(axis=3) <= 2).all()                     # at most two bins fire
    inner = (x >= 0.5 / B) & (x <= 1 - 0.5 / B)
    np.testing.assert_allclose(hb.sum(axis=3)[inner], 1.0, atol=1e-6)  # partition of unity
    # channel order is [bin1.RGB, bin2.RGB, ...]
    c = 1
    centre = np.float32(2 * 3 - 1) / np.float32(2 * B)
    d = np.abs(x[..., c] - centre)
    np.testing.assert_array_equal(h[..., 2 * 3 + c], np.where(d < np.float32(1.0 / B), np.float32(1) - d * np.float32(B), 0))


# ---- (ii) EMoR table check-sums -----------------------------
def test_invemor_table_checksums(emor_table):
    t = emor_table.astype(np.float64)
    assert emor_table.shape == (1024, 12) and emor_table.dtype == np.float32
    assert t[0, 0] == 0.0 and t[-1, 0] == 1.0 and (np.diff(t[:, 0]) >= 0).all()
    assert abs(t[:, 0].sum() - 322.249649) < 1e-4
    assert (t[0, 1:] == 0).all() and (t[-1, 1:] == 0).all()
    assert abs(t[:, 1:].sum() - (-56.117486)) < 1e-4


# ---- (iii) index conventions ------------------------------------------------
def test_same_padding_rule():
    assert ops.same_pad(512, 7, 2) == (256, 2, 3)   # asymmetric: extra on bottom/right
    assert ops.same_pad(128, 1, 2) == (64, 0, 0)
    assert ops.same_pad(256, 3, 2) == (128, 0, 1)
    assert ops.same_pad(64, 3, 1) == (64, 1, 1)
    assert ops.same_pad(64, 5, 1) == (64, 2, 2)
    assert ops.same_pad(7, 3, 2) == (4, 1, 1)


def test_conv_stride2_samples_even_indices():
    x = np.arange(36, dtype=np.float64).reshape(1, 6, 6, 1)
    y = ops.conv2d(x, np.ones((1, 1, 1, 1)), stride=2)
    np.testing.assert_array_equal(y[0, :, :, 0], x[0, ::2, ::2, 0])


def test_conv_7x7_stride2_pad_placement():
    # a delta at the last input row/col is seen by the last output through tap index
    # 2*127... use a small case: H=8, k=7, s=2 -> out 4, pad (2,3)
    x = np.zeros((1, 8, 8, 1))
    x[0, 0, 0, 0] = 1.0
    w = np.arange(49, dtype=np.float64).reshape(7, 7, 1, 1)
    y = ops.conv2d(x, w, stride=2)
    assert y.shape == (1, 4, 4, 1)
    assert y[0, 0, 0, 0] == w[2, 2, 0, 0]    # output 0 window starts at -2
    assert y[0, 1, 1, 0] == w[0, 0, 0, 0]    # output 1 window starts at 0


def test_maxpool3s2_pad_bottom_right_only():
    x = np.arange(16, dtype=np.float64).reshape(1, 4, 4, 1)
    y = ops.max_pool(x, 3, 2)[0, :, :, 0]
    np.testing.assert_array_equal(y, [[10, 11], [14, 15]])
    xn = -x - 1
    yn = ops.max_pool(xn, 3, 2)[0, :, :, 0]     # padded cells must never win
    np.testing.assert_array_equal(yn, [[-1, -3], [-9, -11]])


def test_sobel_reflect_and_channel_order():
    h, w = 5, 6
    ramp_y = np.tile(np.arange(h, dtype=np.float64)[:, None], (1, w))
    x = np.stack([ramp_y, ramp_y.T[:w, :h].T * 0 + np.arange(w)[None, :], np.zeros((h, w))], -1)[None]
    e = ops.sobel_edges(x)
    assert e.shape == (1, h, w, 6)
    np.testing.assert_allclose(e[0, 1:-1, :, 0], 8.0)   # R: vertical ramp -> dy = 8, dx = 0
    np.testing.assert_allclose(e[0, :, :, 1], 0.0)
    np.testing.assert_allclose(e[0, :, 1:-1, 3], 8.0)   # G: horizontal ramp -> dx = 8
    np.testing.assert_allclose(e[0, :, :, 2], 0.0)
    np.testing.assert_allclose(e[0, 0, :, 0], 0.0)      # REFLECT: rows -1 and 1 coincide
    np.testing.assert_allclose(e[0, :, :, 4:], 0.0)


def test_tv_loss_symmetric_pad_adds_zero_difference():
    y = np.random.default_rng(0).random((2, 4, 5, 3))
    ref = np.abs(np.diff(y, axis=1)).sum() / (2 * 4 * 5 * 3) + np.abs(np.diff(y, axis=2)).sum() / (2 * 4 * 5 * 3)
    assert abs(ops.tv_loss(y) - ref) < 1e-12


def test_increase_properties():
    rng = np.random.default_rng(1)
    rf = np.cumsum(rng.normal(0.001, 0.01, (3, 1024)), axis=1)
    out = ops.increase(rf)
    assert out.shape == (3, 1024)
    assert (out[:, 0] == 0).all() and np.allclose(out[:, -1], 1.0)
    assert (np.diff(out, axis=1) >= -1e-15).all()


def test_apply_rf_identity_and_clipping():
    x = np.random.default_rng(2).random((2, 5, 4, 3))
    x[0, 0, 0, 0] = 1.0
    x[1, 0, 0, 0] = 0.0
    rf = np.tile(np.linspace(0, 1, 1024)[None], (2, 1))
    np.testing.assert_allclose(ops.apply_rf(x, rf), x, atol=1e-12)


# ---- (iv) torch-CPU cross-checks --------------------------------------------
def _t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)


@pytest.mark.parametrize("k,stride,h", [(3, 1, 9), (5, 1, 8), (7, 2, 12), (1, 2, 8), (7, 1, 7), (3, 2, 7)])
def test_conv_matches_torch_with_explicit_tf_padding(k, stride, h):
    rng = np.random.default_rng(k * 10 + stride)
    x = rng.normal(size=(2, h, h + 1, 5))
    w = rng.normal(size=(k, k, 5, 4))
    b = rng.normal(size=(4,))
    y = ops.conv2d(x, w, b, stride)
    _, pt, pb = ops.same_pad(h, k, stride)
    _, pl, pr = ops.same_pad(h + 1, k, stride)
    xt = F.pad(_t(x), (pl, pr, pt, pb))
    yt = F.conv2d(xt, torch.from_numpy(w).permute(3, 2, 0, 1), torch.from_numpy(b), stride=stride)
    np.testing.assert_allclose(y, yt.permute(0, 2, 3, 1).numpy(), atol=1e-10)


def test_resize_matches_torch_half_pixel():
    x = np.random.default_rng(3).normal(size=(2, 5, 7, 3))
    yt = F.interpolate(_t(x), scale_factor=2, mode="bilinear", align_corners=False)
    np.testing.assert_allclose(ops.resize_bilinear_2x(x), yt.permute(0, 2, 3, 1).numpy(), atol=1e-12)


def test_pools_match_torch():
    x = np.random.default_rng(4).normal(size=(2, 8, 6, 3))
    np.testing.assert_allclose(ops.avg_pool2(x), F.avg_pool2d(_t(x), 2).permute(0, 2, 3, 1).numpy(), atol=1e-12)
    np.testing.assert_array_equal(ops.max_pool(x, 2, 2), F.max_pool2d(_t(x), 2).permute(0, 2, 3, 1).numpy())
    xp = F.pad(_t(x), (0, 1, 0, 1), value=float("-inf"))
    np.testing.assert_array_equal(ops.max_pool(x, 3, 2), F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1).numpy())


def test_batchnorm_matches_torch():
    rng = np.random.default_rng(5)
    x = rng.normal(size=(3, 4, 5, 6))
    g, b, m = rng.normal(size=6), rng.normal(size=6), rng.normal(size=6)
    v = rng.uniform(0.5, 1.5, 6)
    yt = F.batch_norm(_t(x), torch.from_numpy(m), torch.from_numpy(v), torch.from_numpy(g), torch.from_numpy(b),
                      False, 0.0, 1e-3)
    np.testing.assert_allclose(ops.batch_norm_infer(x, g, b, m, v), yt.permute(0, 2, 3, 1).numpy(), atol=1e-12)
    y, mean, var = ops.batch_norm_train(x, g, b)
    yt = F.batch_norm(_t(x), None, None, torch.from_numpy(g), torch.from_numpy(b), True, 0.0, 1e-3)
    np.testing.assert_allclose(y, yt.permute(0, 2, 3, 1).numpy(), atol=1e-12)


# ---- network-level structure --------------------------------------------------
def test_parameter_counts_match_survey():
    assert nets.count_trainable(nets.deq_spec()) == 1999779
    assert nets.count_trainable(nets.lin_spec()) == 1172747
    assert nets.count_trainable(nets.hal_spec()) == 24569118
    assert nets.count_trainable(nets.ref_spec()) == 1266947
    assert nets.count_trainable(nets.vgg_spec()) == 1735488


def test_frontend_has_93_channels_in_reference_order():
    x = np.random.default_rng(6).random((1, 6, 5, 3))
    f = ops.lin_frontend(x)
    assert f.shape == (1, 6, 5, 93)
    np.testing.assert_array_equal(f[..., :3], x)
    np.testing.assert_array_equal(f[..., 9:21], ops.histogram_layer(x, 4))
    np.testing.assert_array_equal(f[..., 45:93], ops.histogram_layer(x, 16))


# ---- committed golden vectors ---------------------------------------------------
def test_oracle_reproduces_golden_inference(emor_table):
    g = np.load(os.path.join(GOLDEN, "inference_64.npz"))
    params = {k: nets.init_params(getattr(nets, k + "_spec")(), int(g["seed_" + k])) for k in ("deq", "lin", "hal", "ref")}
    out = nets.inference(params, g["ldr"].astype(np.float64), emor_table)
    for key in ("C_pred", "invcrf", "B_pred", "hal", "A_pred", "hdr"):
        np.testing.assert_allclose(out[key], g[key], rtol=2e-6, atol=2e-6, err_msg=key)


def test_torch_cpu_baseline_equals_numpy_oracle(emor_table):
    """oracle/torch_cpu.py (bench.py's `cpu_baseline`: the fp32 channels-last torch-CPU restatement of BASELINE.md section 4)
    computes the function of the NumPy oracle: same seeded weights and input, deq+lin+hal inference, <= 1e-4 relative"""
    from oracle import nets, torch_cpu
    rng = np.random.default_rng(9)
    P = {k: nets.init_params(getattr(nets, k + "_spec")(), 200 + i) for i, k in enumerate(("deq", "lin", "hal"))}
    ldr = np.round(rng.random((1, 64, 96, 3)) * 255.0) / 255.0
    want = nets.inference(P, ldr, emor_table, with_refinement=False)["A_pred"]
    got = torch_cpu.inference({k: torch_cpu.Net(v) for k, v in P.items()}, ldr, emor_table)
    assert got.shape == want.shape
    assert float(np.abs(got - want).max() / np.abs(want).max()) <= 1e-4
