"""GPU parity tests, per op: HIP kernel (through the C ABI) vs the float64 oracle on
the same seeded inputs.  Tolerances: <= 1e-5 tensor-relative per layer for
fp32-accumulated ops (north-star bar is 1e-4 end to end); the soft histogram
is bit-exact against the float32 oracle."""
import os
import zlib

import numpy as np
import pytest
import torch

from conftest import GOLDEN, quantised_image, rel_err
from oracle import ops

pytestmark = pytest.mark.gpu

TOL = 1e-5


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def f32(x):
    return np.asarray(x, dtype=np.float32)


def _act(v, act):
    return {0: lambda t: t, 1: ops.relu, 2: ops.leaky_relu, 3: np.tanh}[act](v)


def oracle_conv(x, w, bias=None, stride=1, x2=None, x2_scale=1.0, act1=0, scale=None, shift=None,
                residual=None, act2=0):
    xin = x if x2 is None else np.concatenate([x, x2 * x2_scale], axis=-1)
    v = ops.conv2d(xin.astype(np.float64), w.astype(np.float64), None if bias is None else bias.astype(np.float64), stride)
    v = _act(v, act1)
    if scale is not None:
        v = v * scale + shift
    if residual is not None:
        v = v + residual[..., :v.shape[-1]]
    return _act(v, act2)


CONV_CASES = [
    # name, N, H, W, C1, C2, Cout, k, stride, extras
    ("mfma128_3x3", 2, 16, 16, 64, 0, 128, 3, 1, dict(bias=True, act1=1)),
    ("mfma64_3x3_ragged", 1, 20, 13, 32, 0, 64, 3, 1, dict(bias=True, act1=2)),
    ("mfma32_5x5", 1, 16, 24, 32, 0, 32, 5, 1, dict(bias=True, act1=2)),
    ("mfma16_7x7_natural", 1, 24, 16, 16, 0, 16, 7, 1, dict(bias=True, act1=2)),
    ("mfma32_5x5_natural_cin16", 1, 16, 16, 16, 0, 32, 5, 1, dict(bias=True)),
    ("mfma16_cin12_natural", 1, 16, 16, 12, 0, 16, 7, 1, dict(bias=True, act1=2)),
    ("mfma_concat_16_16", 1, 16, 16, 16, 16, 16, 3, 1, dict(bias=True, act1=2)),
    ("mfma_concat_scale_1x1", 2, 8, 8, 64, 64, 64, 1, 1, dict(bias=True, x2_scale=1.0 / 255)),
    ("mfma_7x7_s2_cin96", 1, 32, 32, 96, 0, 64, 7, 2, dict(bias=True, affine=True, act2=1)),
    ("mfma_1x1_s2", 1, 16, 16, 64, 0, 128, 1, 2, dict(affine=True)),
    ("mfma_1x1_res_relu", 1, 16, 16, 64, 0, 256, 1, 1, dict(affine=True, residual=True, act2=1)),
    ("mfma_relu_bn_relu", 1, 16, 16, 128, 0, 64, 3, 1, dict(bias=True, act1=1, affine=True, act2=1)),
    ("mfma_small_8x8", 2, 8, 8, 128, 0, 128, 3, 1, dict(bias=True, act1=1)),
    ("mfma_tiny_4x4", 1, 4, 4, 256, 0, 256, 3, 1, dict(bias=True, act1=1)),
    ("mfma_cin3pad4_7x7", 2, 16, 16, 3, 0, 16, 7, 1, dict(bias=True, act1=2, pad_cin=4)),
    ("mfma_cin3pad4_3x3_to64", 1, 16, 16, 3, 0, 64, 3, 1, dict(bias=True, act1=1, pad_cin=4)),
    ("mfma_cin9pad12_7x7", 1, 16, 16, 9, 0, 16, 7, 1, dict(bias=True, act1=2, pad_cin=12)),
    ("mfma_cout3pad16_tanh_res", 1, 16, 16, 16, 0, 3, 3, 1, dict(bias=True, act1=3, residual=True, pad_cout=16)),
    ("mfma_cout3pad16_res12_relu", 1, 16, 24, 16, 0, 3, 3, 1, dict(bias=True, residual=True, res_c=12, act2=1, pad_cout=16)),
    ("mfma_cout3pad16_1x1_bn_relu", 2, 16, 16, 64, 0, 3, 1, 1, dict(bias=True, affine=True, act2=1, pad_cout=16)),
    ("mfma_cout20pad32", 1, 8, 8, 32, 0, 20, 3, 1, dict(bias=True, act1=1, pad_cout=32)),
    ("direct_3to16_7x7", 2, 16, 16, 3, 0, 16, 7, 1, dict(bias=True, act1=2)),
    ("direct_3to64_3x3", 1, 16, 16, 3, 0, 64, 3, 1, dict(bias=True, act1=1)),
    ("direct_9to16_7x7", 1, 16, 16, 9, 0, 16, 7, 1, dict(bias=True, act1=2)),
    ("direct_16to3_tanh_res", 1, 16, 16, 16, 0, 3, 3, 1, dict(bias=True, act1=3, residual=True)),
    ("direct_16to3_res9_relu", 1, 16, 16, 16, 0, 3, 3, 1, dict(bias=True, residual=True, res_c=9, act2=1)),
    ("direct_64to3_bn_relu", 1, 16, 16, 64, 0, 3, 1, 1, dict(bias=True, affine=True, act2=1)),
    ("direct_3p3to3_scale_relu", 1, 16, 16, 3, 3, 3, 1, 1, dict(bias=True, x2_scale=1.0 / 255, act1=1)),
]


@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv2d_parity(shdr, case):
    name, n, h, w, c1, c2, cout, k, stride, ex = case
    rng = np.random.default_rng(zlib.crc32(name.encode()))
    x = f32(rng.normal(size=(n, h, w, c1)))
    x2 = f32(rng.normal(size=(n, h, w, c2)) * (255.0 if "x2_scale" in ex else 1.0)) if c2 else None
    wt = f32(rng.normal(size=(k, k, c1 + c2, cout)) / np.sqrt(k * k * (c1 + c2)))
    bias = f32(rng.normal(size=cout)) if ex.get("bias") else None
    scale = f32(rng.uniform(0.5, 1.5, cout)) if ex.get("affine") else None
    shift = f32(rng.normal(size=cout)) if ex.get("affine") else None
    ho, wo = -(-h // stride), -(-w // stride)
    res = f32(rng.normal(size=(n, ho, wo, ex.get("res_c", cout)))) if ex.get("residual") else None
    kw = dict(stride=stride, x2_scale=ex.get("x2_scale", 1.0), act1=ex.get("act1", 0), act2=ex.get("act2", 0))
    ref = oracle_conv(x, wt, bias, x2=x2, scale=scale, shift=shift, residual=res, **kw)
    K = shdr._ops
    if "pad_cin" in ex:      # zero-padded input channels + zero filter taps: same result, MFMA tile
        pc = ex["pad_cin"] - c1
        x = np.concatenate([x, np.zeros((n, h, w, pc), np.float32)], -1)
        wt = np.concatenate([wt, np.zeros((k, k, pc, cout), np.float32)], 2)
    if "pad_cout" in ex:     # zero-padded filter columns, only `cout` channels stored
        wt = np.concatenate([wt, np.zeros(wt.shape[:3] + (ex["pad_cout"] - cout,), np.float32)], 3)
        kw["cout_valid"] = cout
        kw["algo"] = K.ALGO_MFMA
    if "pad_cin" in ex:
        kw["algo"] = K.ALGO_MFMA
    y = K.conv2d(dev(x), dev(wt), None if bias is None else dev(bias), x2=None if x2 is None else dev(x2),
                 scale=None if scale is None else dev(scale), shift=None if shift is None else dev(shift),
                 residual=None if res is None else dev(res), **kw)
    assert tuple(y.shape) == ref.shape
    assert rel_err(host(y), ref) <= TOL


@pytest.mark.parametrize("shape", [(2, 24, 20, 64, 0, 128, 3, 1), (1, 16, 16, 32, 32, 64, 1, 1), (1, 20, 20, 16, 0, 32, 5, 1),
                                   (1, 33, 17, 96, 0, 64, 7, 2), (1, 16, 16, 16, 16, 16, 3, 1), (2, 9, 9, 128, 0, 16, 3, 1),
                                   (1, 16, 16, 12, 0, 16, 7, 1), (1, 8, 8, 4, 0, 64, 3, 1)])
def test_conv2d_dma_and_register_kernels_bit_identical(shdr, shape):
    """the LDS-DMA and the register-staged MFMA kernels run the same k-ordered fmaf chains"""
    n, h, w, c1, c2, cout, k, stride = shape
    rng = np.random.default_rng(sum(shape))
    K = shdr._ops
    x = dev(f32(rng.normal(size=(n, h, w, c1))))
    x2 = dev(f32(rng.normal(size=(n, h, w, c2)))) if c2 else None
    wt = dev(f32(rng.normal(size=(k, k, c1 + c2, cout)) / np.sqrt(k * k * (c1 + c2))))
    b = dev(f32(rng.normal(size=cout)))
    y_dma = K.conv2d(x, wt, b, stride=stride, x2=x2, act1=K.ACT_RELU, algo=K.ALGO_MFMA)
    y_reg = K.conv2d(x, wt, b, stride=stride, x2=x2, act1=K.ACT_RELU, algo=K.ALGO_MFMA_REG)
    assert rel_err(host(y_dma), host(y_reg)) <= 1e-6


def test_conv2d_mfma_and_direct_agree(shdr):
    """the two kernels are interchangeable where both apply"""
    rng = np.random.default_rng(7)
    x, wt, b = f32(rng.normal(size=(1, 12, 12, 16))), f32(rng.normal(size=(3, 3, 16, 16)) / 12), f32(rng.normal(size=16))
    K = shdr._ops
    ym = K.conv2d(dev(x), dev(wt), dev(b), algo=K.ALGO_MFMA)
    yd = K.conv2d(dev(x), dev(wt), dev(b), algo=K.ALGO_DIRECT)
    assert rel_err(host(ym), host(yd)) <= 2e-6
    with pytest.raises(RuntimeError, match="MFMA path needs"):
        K.conv2d(dev(f32(rng.normal(size=(1, 8, 8, 3)))), dev(f32(rng.normal(size=(3, 3, 3, 16)))), algo=K.ALGO_MFMA)


def test_conv2d_mfma_layout_exact_integers(shdr):
    """A = identity-like / asymmetric integer data: catches transposed fragment maps exactly."""
    rng = np.random.default_rng(8)
    x = f32(rng.integers(-3, 4, size=(1, 16, 16, 32)))
    wt = f32(rng.integers(-2, 3, size=(3, 3, 32, 64)))
    ref = ops.conv2d(x.astype(np.float64), wt.astype(np.float64))
    y = shdr._ops.conv2d(dev(x), dev(wt))
    np.testing.assert_array_equal(host(y), ref.astype(np.float32))


@pytest.mark.parametrize("B", [4, 5, 7, 8, 16, 32])
def test_soft_hist_bit_exact(shdr, B):
    rng = np.random.default_rng(B)
    x = f32(quantised_image(rng, (2, 17, 9, 3)))
    x[0, 0, 0] = [0.0, 1.0, 0.5]
    # values exactly on bin centres / edges
    x[0, 1, :8, 0] = f32(np.arange(8) / 8.0)
    ref = ops.histogram_layer(x, B)                  # float32 evaluation of the reference formula
    y = host(shdr._ops.soft_hist(dev(x), B))
    assert y.dtype == np.float32 and y.shape == ref.shape
    np.testing.assert_array_equal(y.view(np.uint32), ref.view(np.uint32))
    np.testing.assert_array_equal(y > 0, ref > 0)    # "which bins fire" index predicate


@pytest.mark.parametrize("B", [32, 12])
def test_soft_hist_bit_exact_grid_stride(shdr, B):
    """enough pixels that every thread of soft_hist_rows_kernel walks its 4-pixel-unrolled grid-stride loop and its tail
    (B = 12: 9 quads per pixel, the stride unit is 9 blocks)"""
    rng = np.random.default_rng(100 + B)
    x = f32(quantised_image(rng, (5, 163, 160, 3)))
    ref = ops.histogram_layer(x, B)
    y = host(shdr._ops.soft_hist(dev(x), B))
    np.testing.assert_array_equal(y.view(np.uint32), ref.view(np.uint32))


def test_soft_hist_known_answer_lin2(shdr):
    x = f32(np.array([0.63, 0.65, 0.32, 0.84, 0.15]).reshape(1, 1, 5, 1))
    y = host(shdr._ops.soft_hist(dev(x), 5))[0, 0]
    expect = np.array([[0, 0, .35, .65, 0], [0, 0, .25, .75, 0], [0, .9, .1, 0, 0], [0, 0, 0, .3, .7], [.75, .25, 0, 0, 0]])
    np.testing.assert_allclose(y, expect, atol=1e-6)


def test_soft_hist_and_frontend_golden(shdr):
    g = np.load(os.path.join(GOLDEN, "frontend_16.npz"))
    x = dev(g["x"])
    for b in (4, 5, 8, 16, 32):
        np.testing.assert_array_equal(host(shdr._ops.soft_hist(x, b)), g["hist%d" % b])
    fe = host(shdr._ops.lin_frontend(x, 96))
    np.testing.assert_array_equal(fe[..., :3], g["x"])
    np.testing.assert_array_equal(fe[..., 9:93], g["frontend93"][..., 9:93])      # histogram channels: bit exact
    np.testing.assert_allclose(fe[..., 3:9], g["frontend93"][..., 3:9], atol=2e-6)  # sobel: fp32 sum order
    assert float(np.abs(fe[..., 93:]).max()) == 0.0
    fe93 = host(shdr._ops.lin_frontend(x, 93))
    np.testing.assert_array_equal(fe93, fe[..., :93])


def test_lin_frontend_parity_ragged(shdr):
    x = f32(quantised_image(np.random.default_rng(3), (2, 7, 5, 3)))
    ref = ops.lin_frontend(x.astype(np.float64))
    y = host(shdr._ops.lin_frontend(dev(x), 96))
    assert rel_err(y[..., :93], ref) <= 1e-6


@pytest.mark.parametrize("shape", [(2, 8, 12, 16), (1, 6, 6, 64)])
def test_pools_and_resize_parity(shdr, shape):
    K = shdr._ops
    x = f32(np.random.default_rng(5).normal(size=shape))
    xd = dev(x)
    np.testing.assert_allclose(host(K.avgpool2(xd)), ops.avg_pool2(x.astype(np.float64)), atol=1e-6)
    np.testing.assert_array_equal(host(K.maxpool2(xd)), ops.max_pool(x, 2, 2))
    np.testing.assert_array_equal(host(K.maxpool3s2(xd)), ops.max_pool(x, 3, 2))
    np.testing.assert_allclose(host(K.resize2x(xd)), ops.resize_bilinear_2x(x.astype(np.float64)), atol=1e-6)


def test_maxpool3s2_odd_size(shdr):
    x = f32(np.random.default_rng(6).normal(size=(1, 7, 9, 8)))
    np.testing.assert_array_equal(host(shdr._ops.maxpool3s2(dev(x))), ops.max_pool(x, 3, 2))


def test_global_avg_pool_parity(shdr):
    x = f32(np.random.default_rng(7).normal(size=(3, 16, 16, 512)))
    assert rel_err(host(shdr._ops.global_avg_pool(dev(x))), ops.global_avg_pool(x.astype(np.float64))) <= 1e-6
    x = f32(np.random.default_rng(8).normal(size=(2, 5, 3, 24)))
    assert rel_err(host(shdr._ops.global_avg_pool(dev(x))), ops.global_avg_pool(x.astype(np.float64))) <= 1e-6


def test_invcrf_head_parity(shdr, emor_table):
    K = shdr._ops
    rng = np.random.default_rng(9)
    feat, wfc, bfc = f32(rng.normal(size=(4, 512))), f32(rng.normal(size=(512, 11)) * 0.05), f32(rng.normal(size=11))
    w = ops.dense(feat.astype(np.float64), wfc, bfc)
    ref = ops.invcrf_pca_decode(w, emor_table[:, 0], emor_table[:, 1:])
    y = K.invcrf_decode(dev(feat), dev(wfc), dev(bfc), dev(emor_table))
    assert rel_err(host(y), ref) <= 2e-6
    inc = host(K.increase(y))
    ref_inc = ops.increase(host(y).astype(np.float64))
    assert rel_err(inc, ref_inc) <= 2e-6
    assert (inc[:, 0] == 0).all() and np.allclose(inc[:, -1], 1.0, atol=2e-6) and (np.diff(inc, axis=1) >= 0).all()


def test_increase_edge_cases(shdr):
    K = shdr._ops
    rf = f32(np.tile(np.linspace(0, 1, 1024)[None], (2, 1)))          # already monotone -> unchanged
    np.testing.assert_allclose(host(K.increase(dev(rf))), rf, atol=2e-6)
    rf2 = f32(np.random.default_rng(1).normal(size=(1, 37)))            # ragged K
    assert rel_err(host(K.increase(dev(rf2))), ops.increase(rf2.astype(np.float64))) <= 2e-6


def test_apply_rf_parity(shdr):
    K = shdr._ops
    rng = np.random.default_rng(10)
    x = f32(quantised_image(rng, (3, 8, 8, 3)))
    x[0, 0, 0] = [0.0, 1.0, 0.5]
    rf = f32(np.sort(rng.random((3, 1024)), axis=1))
    assert rel_err(host(K.apply_rf(dev(x), dev(rf))), ops.apply_rf(x.astype(np.float64), rf.astype(np.float64))) <= 1e-6
    ident = f32(np.tile(np.linspace(0, 1, 1024)[None], (3, 1)))
    np.testing.assert_allclose(host(K.apply_rf(dev(x), dev(ident))), x, atol=1e-6)
    xo = f32(rng.random((2, 5, 3)))                                      # n_per_batch not a multiple of 4
    assert rel_err(host(K.apply_rf(dev(xo), dev(rf[:2]))), ops.apply_rf(xo.astype(np.float64), rf[:2].astype(np.float64))) <= 1e-6


def test_glue_ops_parity(shdr):
    K = shdr._ops
    rng = np.random.default_rng(11)
    x = f32(rng.random((2, 6, 5, 3)) * 1.4 - 0.2)
    np.testing.assert_array_equal(host(K.clip(dev(x), 0.0, 1.0)), np.clip(x, 0, 1))
    np.testing.assert_array_equal(host(K.reverse3(dev(x))), x[..., ::-1])
    np.testing.assert_allclose(host(K.vgg_preprocess(dev(x))), ops.vgg_preprocess(x.astype(np.float64)), rtol=1e-6, atol=1e-4)
    v4 = host(K.vgg_preprocess(dev(x), 4))
    np.testing.assert_array_equal(v4[..., :3], host(K.vgg_preprocess(dev(x))))
    assert float(np.abs(v4[..., 3]).max()) == 0.0
    xp = np.abs(x)
    np.testing.assert_allclose(host(K.logc(dev(xp))), ops.log_compress(xp.astype(np.float64)), atol=1e-6)
    b = f32(rng.random((2, 6, 5, 3)))
    b[0, 0, 0] = [1.0, 0.2, 0.3]
    b[0, 0, 1] = [0.9, 0.95, 0.1]
    hal = f32(rng.random((2, 6, 5, 3)))
    a, alpha = K.alpha_blend(dev(b), dev(hal), 0.12, return_alpha=True)
    np.testing.assert_allclose(host(a), ops.alpha_blend(b.astype(np.float64), hal.astype(np.float64)), atol=1e-6)
    np.testing.assert_allclose(host(alpha), ops.alpha_mask(b.astype(np.float64))[..., :1], atol=1e-5)
    p = host(K.pack3([dev(x), dev(b), dev(hal)]))
    np.testing.assert_array_equal(p, np.concatenate([x, b, hal], -1))
    p12 = host(K.pack3([dev(x), dev(b), dev(hal)], 12))
    np.testing.assert_array_equal(p12[..., :9], p)
    assert float(np.abs(p12[..., 9:]).max()) == 0


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 128), (1, 13, 18, 32, 64), (1, 8, 8, 256, 256), (3, 6, 4, 128, 16)])
def test_conv2d_winograd_parity(shdr, shape):
    """Winograd F(2x2,3x3) path (filter / input / output transforms + batched MFMA GEMM) vs the float64 oracle"""
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape))
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, cin)))
    wt = f32(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, sc, sh = f32(rng.normal(size=cout)), f32(rng.uniform(0.5, 1.5, cout)), f32(rng.normal(size=cout))
    u = K.winograd_filter(dev(wt))
    ref = oracle_conv(x, wt, b, act1=1, scale=sc, shift=sh, act2=1)
    y = K.conv2d_winograd(dev(x), u, dev(b), act1=K.ACT_RELU, scale=dev(sc), shift=dev(sh), act2=K.ACT_RELU)
    assert tuple(y.shape) == ref.shape and rel_err(host(y), ref) <= TOL
    y0 = K.conv2d_winograd(dev(x), u)
    assert rel_err(host(y0), oracle_conv(x, wt)) <= TOL
    assert rel_err(host(y0), host(K.conv2d(dev(x), dev(wt)))) <= 5e-6     # vs the direct MFMA kernel


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 128), (1, 13, 18, 32, 64), (1, 8, 16, 256, 64), (3, 6, 4, 8, 64),
                                   (1, 21, 37, 72, 192), (2, 32, 32, 128, 64), (1, 16, 16, 512, 512),
                                   (1, 9, 17, 16, 64), (2, 10, 33, 24, 128)])     # 2 / 3 chunks: shorter than the 4-deep pipeline
def test_conv2d_winograd_fused_parity(shdr, shape):
    """One-kernel Winograd F(2x2,3x3) (operands built per lane from the raw patch in LDS) vs the float64 oracle: ragged
    tiles (H % 8, W % 16 != 0, image smaller than one block tile), Cin a multiple of 8 only, the fused epilogue."""
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape))
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, cin)))
    wt = f32(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, sc, sh = f32(rng.normal(size=cout)), f32(rng.uniform(0.5, 1.5, cout)), f32(rng.normal(size=cout))
    u = K.winograd_filter_packed(dev(wt))
    ref = oracle_conv(x, wt, b, act1=2, scale=sc, shift=sh, act2=1)
    y = K.conv2d_winograd_fused(dev(x), u, dev(b), act1=K.ACT_LRELU, scale=dev(sc), shift=dev(sh), act2=K.ACT_RELU)
    assert tuple(y.shape) == ref.shape and rel_err(host(y), ref) <= TOL
    y0 = K.conv2d_winograd_fused(dev(x), u)
    assert rel_err(host(y0), oracle_conv(x, wt)) <= TOL
    K.WINOGRAD, saved = False, K.WINOGRAD
    try:
        assert rel_err(host(y0), host(K.conv2d(dev(x), dev(wt)))) <= 5e-6     # vs the direct MFMA kernel
    finally:
        K.WINOGRAD = saved
    assert K.winograd_path(cin, cout) == ("fused" if cin >= 32 else None)      # ... and this is what conv2d() dispatches to
    if cin >= 32:
        assert torch.equal(K.conv2d(dev(x), dev(wt), dev(b), act1=K.ACT_LRELU, scale=dev(sc), shift=dev(sh), act2=K.ACT_RELU), y)
    with pytest.raises(ValueError, match="Cout"):
        K.winograd_filter_packed(dev(np.ascontiguousarray(wt[..., :48])))


@pytest.mark.parametrize("shape", [(2, 8, 8, 64, 128), (1, 5, 7, 32, 64), (1, 13, 18, 40, 64), (2, 1, 1, 32, 64), (1, 16, 24, 512, 128),
                                   (1, 3, 33, 72, 192), (1, 32, 32, 128, 64), (1, 4, 9, 16, 64), (1, 6, 5, 64, 32), (1, 7, 4, 32, 16)])
def test_conv2d_up2_fused_parity(shdr, shape):
    """Conv2D 3x3 of tf.image.resize(x, 2x, BILINEAR) (hallucination_net.py:86-88, dequantization_net.py:25-27) with the resize
    fused into the one-kernel Winograd (low-res patch staged by DMA, expanded in LDS): vs the float64 oracle (resize, then conv)
    and BIT-IDENTICAL to the two-kernel path (the expansion repeats resize2x_kernel's arithmetic); ragged tiles, a 1 x 1 source,
    2 .. 64 channel chunks; layers the fused plan does not take (Cin < 32, Cout % 64 != 0) go through the workspace."""
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape) + 1)
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, cin)))
    wt = f32(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, sc, sh = f32(rng.normal(size=cout)), f32(rng.uniform(0.5, 1.5, cout)), f32(rng.normal(size=cout))
    up = ops.resize_bilinear_2x(x.astype(np.float64))
    ref = oracle_conv(up, wt, b, act1=1, scale=sc, shift=sh, act2=1)
    y = K.conv2d_up2(dev(x), dev(wt), dev(b), act1=K.ACT_RELU, scale=dev(sc), shift=dev(sh), act2=K.ACT_RELU)
    assert tuple(y.shape) == ref.shape and rel_err(host(y), ref) <= TOL
    two = K.conv2d(K.resize2x(dev(x)), dev(wt), dev(b), act1=K.ACT_RELU, scale=dev(sc), shift=dev(sh), act2=K.ACT_RELU)
    assert torch.equal(y, two)
    y0 = K.conv2d_up2(dev(x), dev(wt))
    assert rel_err(host(y0), oracle_conv(up, wt)) <= TOL
    fused = cin >= 32 and cout % 64 == 0
    assert (K.conv2d_plan((n, 2 * h, 2 * w, cin), wt.shape) == "fused") == fused


X3_CASES = [(2, 64, 96, 32, 0, 64), (1, 128, 128, 64, 0, 128), (1, 100, 140, 64, 64, 64), (4, 64, 64, 96, 0, 192), (1, 80, 80, 512, 0, 256),
            (1, 96, 96, 32, 32, 64)]


@pytest.mark.parametrize("shape", X3_CASES, ids=["%dx%dx%d_%d+%d_%d" % c for c in X3_CASES])
def test_conv2d_x3_split_fp16_kernel_is_fp32_accurate(shdr, shape, monkeypatch):
    """SHDR_PLAN_X3 (csrc/conv_x3.hip): fp32 3x3 convolution as three fp16 MFMA products of split operands.  Held to the SAME bar
    against the float64 oracle as the exact-fp32 kernels (TOL = 1e-5 of the tensor scale) -- with unit-scale, 1e-3-scale and
    1e+3-scale activations (the low term of an operand is stored scaled, the weights are scaled per layer: neither underflows) --
    and compared with what the exact-fp32 path (ALGO_AUTO_EXACT: fused Winograd) reaches on the same input.  Ragged tiles, two
    sources with a skip scale, the fused epilogue."""
    n, h, w, c1, c2, cout = shape
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")       # the plan's fill-the-chip threshold is for speed only; read per call
    rng = np.random.default_rng(sum(shape) + 7)
    K = shdr._ops
    wt = f32(rng.normal(size=(3, 3, c1 + c2, cout)) / np.sqrt(9 * (c1 + c2)))
    b, sc, sh = f32(rng.normal(size=cout)), f32(rng.uniform(0.5, 1.5, cout)), f32(rng.normal(size=cout))
    assert K.conv2d_plan((n, h, w, c1), wt.shape, c2=c2, x2_scale=1.0 / 255 if c2 else 1.0) == "x3"
    for mag in (1.0, 1e-3, 1e3):
        x = f32(rng.normal(size=(n, h, w, c1)) * mag)
        x2 = f32(rng.normal(size=(n, h, w, c2)) * (255.0 * mag if mag <= 1.0 else 8000.0)) if c2 else None    # fp16 range: |x| < 65504
        x2s = 1.0 / 255 if c2 else 1.0
        ref = oracle_conv(x, wt, b * mag, x2=x2, x2_scale=x2s, act1=2, scale=sc, shift=sh * mag, act2=1)
        kw = dict(x2=None if x2 is None else dev(x2), x2_scale=x2s, act1=K.ACT_LRELU, scale=dev(sc), shift=dev(sh * mag), act2=K.ACT_RELU)
        y = K.conv2d(dev(x), dev(wt), dev(b * mag), **kw)
        err_x3 = rel_err(host(y), ref)
        monkeypatch.setattr(K, "EXACT_FP32", True)
        assert K.conv2d_plan((n, h, w, c1), wt.shape, c2=c2, x2_scale=x2s) != "x3"
        err_exact = rel_err(host(K.conv2d(dev(x), dev(wt), dev(b * mag), **kw)), ref)
        monkeypatch.setattr(K, "EXACT_FP32", False)
        assert tuple(y.shape) == ref.shape and err_x3 <= TOL, (mag, err_x3, err_exact)
        assert err_x3 <= 4 * err_exact + 2e-7, (mag, err_x3, err_exact)       # the same accuracy class as the exact-fp32 kernels
    # plain conv, no epilogue; tiny weights (the per-layer scale) and a weight tensor with one huge outlier
    for wmag, outlier in ((1e-4, False), (1.0, True)):
        w2 = wt * np.float32(wmag)
        if outlier:
            w2 = w2.copy()
            w2[1, 1, 0, 0] = 37.0
        x = f32(rng.normal(size=(n, h, w, c1)))
        x2 = f32(rng.normal(size=(n, h, w, c2))) if c2 else None
        y0 = K.conv2d(dev(x), dev(w2), x2=None if x2 is None else dev(x2))
        assert rel_err(host(y0), oracle_conv(x, w2, x2=x2)) <= TOL


def _local_rel(y, ref, x_like, w):
    """worst error of an output pixel relative to the magnitude scale of ITS OWN receptive field (a 3 x 3 box maximum of |input|
    times the filter norm) -- unlike rel_err, which divides by the tensor maximum and hides small-magnitude regions next to large ones"""
    import scipy.ndimage as ndi
    mag = np.abs(x_like).max(axis=-1)
    box = ndi.maximum_filter(mag, size=(1, w.shape[0], w.shape[1]), mode="constant")
    scale = box[..., None] * float(np.abs(w).sum(axis=(0, 1, 2)).max()) + 1e-300
    return float((np.abs(np.asarray(y, np.float64) - ref) / scale).max())


RANGE_CASES = [("x3_64_128", 2, 48, 64, 64, 0, 128, 3), ("x3_two_sources", 1, 64, 64, 32, 32, 64, 3), ("x3_1x1_256", 1, 64, 64, 256, 0, 64, 1),
               ("x3n_7x7_16_16", 1, 64, 64, 16, 0, 16, 7), ("x3n_3x3_16+16_16", 1, 48, 80, 16, 16, 16, 3), ("x3n_5x5_32_32", 1, 64, 64, 32, 0, 32, 5)]


@pytest.mark.parametrize("case", RANGE_CASES, ids=[c[0] for c in RANGE_CASES])
@pytest.mark.parametrize("known_range", [False, True], ids=["measured", "slot_from_producer"])
def test_split_operand_forward_is_range_safe(shdr, case, known_range, monkeypatch):
    """The split-operand plans (x3 / x3n) against the exact-fp32 plan of the SAME layer outside the fp16 range
    (hallucination_net.py:47-48 and vgg16.py:33-35 are fp32 convolutions without a range limit): activations at 1e5 and 1e-7 scale, one
    7e4 outlier in an O(1) tensor, +-inf inputs.  The input is scaled by a power of two taken from its range slot -- measured below the
    ABI when the caller has none, or written by the producer's epilogue.  Finite cases: <= 1e-5 of the tensor maximum against the
    float64 oracle (the bar of the exact kernels) AND <= 1e-5 of each output's own receptive-field scale, so that the O(1) region next
    to the outlier is checked too; non-finite cases: the output is non-finite exactly where the exact plan's is, equal elsewhere."""
    name, n, h, w, c1, c2, cout, k = case
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    K = shdr._ops
    rng = np.random.default_rng(len(name) * 131 + k)
    wt = f32(rng.normal(size=(k, k, c1 + c2, cout)) / np.sqrt(k * k * (c1 + c2)))
    b = f32(rng.normal(size=cout))
    assert K.conv2d_plan((n, h, w, c1), wt.shape, c2=c2) == name.split("_")[0]

    def run(x, x2, exact):
        monkeypatch.setattr(K, "EXACT_FP32", exact)
        xd, x2d = dev(x), (None if x2 is None else dev(x2))
        if known_range and not exact:           # the slots a producing kernel would have written (K: "range slots")
            K.absmax_slot(xd)
            if x2d is not None:
                K.absmax_slot(x2d)
        y = K.conv2d(xd, dev(wt), dev(b * bias_mag), x2=x2d, act1=K.ACT_LRELU)
        if not exact:                           # ... and the slot this launch wrote for its consumer holds max |y|
            assert hasattr(y, "_shdr_range")
            got, want = float(y._shdr_range), float(y[torch.isfinite(y)].abs().max())
            assert got == want or not np.isfinite(got), (got, want)
        monkeypatch.setattr(K, "EXACT_FP32", False)
        return host(y)

    for label, mag, outlier in (("1e5", 1e5, None), ("1e-7", 1e-7, None), ("outlier_7e4", 1.0, 7e4), ("tiny_3e-30", 3e-30, None),
                                ("huge_1e30", 1e30, None)):
        bias_mag = mag
        x = f32(rng.normal(size=(n, h, w, c1)) * mag)
        x2 = f32(rng.normal(size=(n, h, w, c2)) * mag) if c2 else None
        if outlier:
            x[0, h // 3, w // 3, 1] = outlier
        ref = oracle_conv(x, wt, b * bias_mag, x2=x2, act1=2)
        y, ye = run(x, x2, False), run(x, x2, True)
        assert np.isfinite(y).all(), label
        e_split, e_exact = rel_err(y, ref), rel_err(ye, ref)
        assert e_split <= TOL and e_split <= 4 * e_exact + 2e-7, (label, e_split, e_exact)
        xa = x if x2 is None else np.concatenate([x, x2], -1)
        l_split, l_exact = _local_rel(y, ref, xa, wt), _local_rel(ye, ref, xa, wt)
        assert l_split <= TOL and l_split <= 4 * l_exact + 2e-7, (label, l_split, l_exact)
    # non-finite inputs: +inf and -inf pixels
    bias_mag = 1.0
    x = f32(rng.normal(size=(n, h, w, c1)))
    x2 = f32(rng.normal(size=(n, h, w, c2))) if c2 else None
    x[0, 5, 7, 0] = np.inf
    x[n - 1, h - 9, w - 4, c1 - 1] = -np.inf
    y, ye = run(x, x2, False), run(x, x2, True)
    bad, bad_e = ~np.isfinite(y), ~np.isfinite(ye)
    assert bad_e.any() and np.array_equal(bad, bad_e)
    assert np.abs(y[~bad] - ye[~bad]).max() <= TOL * np.abs(ye[~bad]).max()


def test_split_operand_range_slots_travel_with_the_tensors(shdr, monkeypatch):
    """conv -> pool -> conv chains hand the range slot on (no measuring pass), bound-preserving ops keep it, host-known bounds become
    constant slots; a chain run with slots equals the chain run with every range measured, bit for bit (same power-of-two scale)"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    K = shdr._ops
    rng = np.random.default_rng(5)
    x = dev(f32(rng.normal(size=(2, 64, 64, 32)) * 3e5))
    w1 = dev(f32(rng.normal(size=(3, 3, 32, 64)) / 17)).requires_grad_(True)
    w2 = dev(f32(rng.normal(size=(3, 3, 64, 64)) / 24)).requires_grad_(True)
    with torch.no_grad(), K.range_scope():
        y1, p1 = K.conv2d_avgpool2(x, w1, None, act1=K.ACT_RELU)
        assert y1._shdr_range is p1._shdr_range and float(y1._shdr_range) == float(y1.abs().max())
        up = K.resize2x(K.maxpool2(p1))
        assert up._shdr_range is p1._shdr_range
        y2 = K.conv2d(up, w2)
        c = K.clip(y2, 0.0, 1.0)
        assert c._shdr_bound == 1.0 and float(K._range_of(c)) == 1.0
        # the exact-fp32 MFMA / direct kernels track their output range too (their consumer may be a split-operand layer)
        w3 = dev(f32(rng.normal(size=(1, 1, 64, 96)) / 8)).requires_grad_(True)       # (96 couts: not a split-operand layer)
        assert K.conv2d_plan(tuple(y2.shape), tuple(w3.shape)) == "mfma"
        res96 = dev(f32(rng.normal(size=tuple(y2.shape[:3]) + (96,))))
        for kw in (dict(), dict(act1=K.ACT_RELU, residual=res96, act2=K.ACT_RELU)):
            y3 = K.conv2d(y2, w3, **kw)
            assert float(y3._shdr_range) == float(y3.abs().max())
        w4 = dev(f32(rng.normal(size=(1, 1, 64, 3)))).requires_grad_(True)
        assert K.conv2d_plan(tuple(y2.shape), tuple(w4.shape)) == "direct"
        y4 = K.conv2d(y2, w4)
        assert float(y4._shdr_range) == float(y4.abs().max())
        plain = K.conv2d(up.clone(), w2)              # no slot on the clone: measured below the ABI (possibly another power of two)
        assert bool(torch.isfinite(y2).all()) and float((plain - y2).abs().max()) <= 1e-6 * float(y2.abs().max())


def test_backward_elementwise_kernels_track_their_output_range(shdr):
    """the sums / BatchNorm and resize input gradients of a backward pass write max |out| into a range slot in the same pass
    (shdr_add_ranged_f32, shdr_bn_bwd_ranged_f32, shdr_resize2x_bwd_ranged_f32); pooling gradients bounded by dy carry dy's slot"""
    K = shdr._ops
    rng = np.random.default_rng(11)
    with torch.no_grad(), K.range_scope():
        a, b = dev(f32(rng.normal(size=(2, 16, 16, 32)) * 1e-6)), dev(f32(rng.normal(size=(2, 16, 16, 32)) * 1e-6))
        s = K.add(a, b)
        assert torch.equal(s, a + b) and float(s._shdr_range) == float(s.abs().max())
        x = dev(f32(rng.normal(size=(2, 16, 16, 32))))
        mean, var = K.bn_stats(x)
        gamma, beta = dev(f32(rng.normal(size=32))), dev(f32(rng.normal(size=32)))
        y = K.bn_train_apply(x, mean, var, gamma, beta, 1e-3, True)
        dx, _, _ = K.bn_bwd(s, x, y, mean, var, gamma, 1e-3)
        assert float(dx._shdr_range) == float(dx.abs().max()) > 0.0
        dx1, _, _ = K.bn_bwd(s[..., :3].contiguous(), x[..., :3].contiguous(), None, mean[:3].contiguous(), var[:3].contiguous(),
                             gamma[:3].contiguous(), 1e-3)            # the scalar kernel (C % 4 != 0)
        assert float(dx1._shdr_range) == float(dx1.abs().max()) > 0.0
        big = dev(f32(rng.normal(size=(2, 32, 32, 32)) * 3e4))
        r = K.resize2x_bwd(big, (2, 16, 16, 32))
        assert float(r._shdr_range) == float(r.abs().max()) > float(big.abs().max())      # up to nine weighted taps add up
        K.absmax_slot(big)
        for g in (K.avgpool2_bwd(big, (2, 64, 64, 32)), K.maxpool2_bwd(dev(f32(rng.normal(size=(2, 64, 64, 32)))), big),
                  K.upsample_zero2(big, (2, 64, 64, 32))):
            assert g._shdr_range is big._shdr_range and float(g.abs().max()) <= float(big._shdr_range)


def test_projected_output_of_the_split_operand_kernel(shdr, monkeypatch):
    """shdr_conv2d_fwd_prepared_projected_f32: sum_c proj[j, c] y[..., c] from the epilogue that holds y (plain, with the fused 2x2
    max-pool, and behind the bilinear 2x prologue) equals the 1x1 convolution of the written y (hallucination_net.py:179-185)"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    K = shdr._ops
    rng = np.random.default_rng(21)
    x = dev(f32(rng.normal(size=(2, 32, 48, 64))))
    w = dev(f32(rng.normal(size=(3, 3, 64, 64)) / 24)).requires_grad_(True)
    b = dev(f32(rng.normal(size=64)))
    sc, sh = dev(f32(rng.uniform(0.5, 1.5, size=64))), dev(f32(rng.normal(size=64)))
    proj = dev(f32(rng.normal(size=(3, 64))))
    with torch.no_grad(), K.range_scope():
        y, yp = K.conv2d_maxpool2(x, w, b, act1=K.ACT_RELU)
        pj, pp = K.conv2d_maxpool2(x, w, b, act1=K.ACT_RELU, proj=proj)
        want = torch.einsum("nhwc,jc->nhwj", y.double(), proj.double())
        assert torch.equal(pp, yp) and float((pj.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
        assert float(pp._shdr_range) >= float(pp.abs().max())
        lo = dev(f32(rng.normal(size=(2, 16, 24, 64))))
        yu = K.conv2d_up2(lo, w, b, act1=K.ACT_RELU, scale=sc, shift=sh, act2=K.ACT_RELU)
        pu = K.conv2d_up2(lo, w, b, act1=K.ACT_RELU, scale=sc, shift=sh, act2=K.ACT_RELU, proj=proj)
        want = torch.einsum("nhwc,jc->nhwj", yu.double(), proj.double())
        assert float((pu.double() - want).abs().max()) <= 1e-5 * float(want.abs().max())
        # layers the split-operand plan does not take (128 output channels; the exact-fp32 plans) report it: the caller runs two convolutions
        w128 = dev(f32(rng.normal(size=(3, 3, 64, 128)) / 24)).requires_grad_(True)
        assert K.conv2d_maxpool2(x, w128, None, proj=dev(f32(rng.normal(size=(3, 128))))) is None
        K.EXACT_FP32 = True
        try:
            assert K.conv2d_maxpool2(x, w, b, act1=K.ACT_RELU, proj=proj) is None
        finally:
            K.EXACT_FP32 = False


WGRAD_X3_CASES = [("3x3_128_128", 2, 32, 32, 128, 0, 128, 3, 1), ("3x3_256_128_ragged", 1, 21, 37, 256, 0, 128, 3, 1),
                  ("3x3_two_sources_scaled", 2, 16, 24, 128, 128, 128, 3, 1), ("3x3_128_192", 1, 24, 24, 128, 0, 192, 3, 1),
                  ("3x3_stride2_256_128", 2, 32, 32, 256, 0, 128, 3, 2), ("7x7_stride2_96_64_lin_stem", 2, 40, 48, 96, 0, 64, 7, 2)]


@pytest.mark.parametrize("case", WGRAD_X3_CASES, ids=[c[0] for c in WGRAD_X3_CASES])
def test_split_operand_weight_gradient(shdr, case, monkeypatch):
    """csrc/wgrad_x3.hip: dW of the deep layers as three fp16 MFMA products of split operands (both operands are activations: X and dZ
    are split, dZ range-scaled from its measured maximum).  Against the float64 sum (joint_training.py:185-186) at the bar of the
    exact-fp32 weight-gradient kernels (1e-4 of max |dW|; measured next to the exact plan's error), with unit-scale, TINY (1e-7: output
    gradients of the joint step reach 3e-8) and heavy-tailed dZ, and huge / tiny activations."""
    name, n, h, w, c1, c2, cout, k, stride = case
    K = shdr._ops
    rng = np.random.default_rng(len(name) * 17 + k)
    x2s = 1.0 / 255 if c2 else 1.0
    ho, wo = -(-h // stride), -(-w // stride)
    import torch_ref as R
    for label, xmag, zmag, tail in (("unit", 1.0, 1.0, False), ("tiny_dz", 1.0, 1e-7, False), ("heavy_tail_dz", 1.0, 1e-3, True),
                                    ("big_x_tiny_dz", 3e4, 3e-8, False), ("tiny_x", 1e-6, 1.0, True)):
        x = f32(rng.normal(size=(n, h, w, c1)) * xmag)
        x2 = f32(rng.normal(size=(n, h, w, c2)) * xmag * 255.0) if c2 else None
        dz = rng.normal(size=(n, ho, wo, cout)) * zmag
        if tail:
            dz = dz * np.exp(3.0 * rng.normal(size=dz.shape))          # log-normal magnitudes: five decades inside one tensor
        dz = f32(dz)
        tw = R.T(np.zeros((k, k, c1 + c2, cout)), True)
        xin = R.T(x) if c2 == 0 else torch.cat([R.T(x), R.T(x2) * x2s], -1)
        (R.conv2d(xin, tw, None, stride) * R.T(dz)).sum().backward()
        ref = tw.grad.numpy()

        def run(exact):
            monkeypatch.setattr(K, "EXACT_FP32", exact)
            got = K.conv2d_wgrad(dev(x), None if x2 is None else dev(x2), dev(dz), (k, k, c1 + c2, cout), stride, x2s)
            monkeypatch.setattr(K, "EXACT_FP32", False)
            return rel_err(host(got), ref)
        e_split, e_exact = run(False), run(True)
        assert e_split <= 1e-4 and e_split <= 4 * e_exact + 1e-6, (label, e_split, e_exact)
    # the plan: this layer does take the split-operand kernel (and the exact switch does not)
    lib, d = shdr._lib.load(), K._conv_desc((n, h, w, c1), (k, k, c1 + c2, cout), stride, c2, x2s, None)
    import ctypes
    assert lib.shdr_conv2d_wgrad_x3_ok_f32(ctypes.byref(d), 0) == 1 and (min(c1, cout) >= K.WGRAD_X3_MIN_CH or (k * k >= 25 and min(c1, cout) >= 64))
    acc = torch.full((k, k, c1 + c2, cout), 2.0, device="cuda")               # `out=`: accumulates into a flat-gradient view
    g = K.conv2d_wgrad(dev(x), None if x2 is None else dev(x2), dev(dz), (k, k, c1 + c2, cout), stride, x2s)
    K.conv2d_wgrad(dev(x), None if x2 is None else dev(x2), dev(dz), (k, k, c1 + c2, cout), stride, x2s, out=acc)
    assert float((acc - 2.0 - g).abs().max()) <= 1e-5 * float(g.abs().max()) + 1e-6


def test_conv2d_x3_dgrad_and_maxpool_pair(shdr, monkeypatch):
    """the input gradient of a wide 3x3 layer takes the split kernel too (shdr_conv2d_dgrad_f32), and conv + MaxPool2D pairs run as
    x3 + pooling; both vs the float64 reference"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    K = shdr._ops
    rng = np.random.default_rng(11)
    n, h, w, cin, cout = 2, 64, 96, 64, 128
    x = f32(rng.normal(size=(n, h, w, cin)))
    wt = f32(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    dz = f32(rng.normal(size=(n, h, w, cout)))
    dx = K.conv2d_dgrad(dev(dz), dev(wt), (n, h, w, cin), cin, 0, 0)
    wflip = np.ascontiguousarray(wt[::-1, ::-1].transpose(0, 1, 3, 2))
    assert rel_err(host(dx), oracle_conv(dz, wflip)) <= TOL
    assert K.conv2d_plan((n, h, w, cin), wt.shape) == "x3"
    y, yp = K.conv2d_maxpool2(dev(x), dev(wt), None, act1=K.ACT_RELU)          # ONE launch: the pooling window sits in the epilogue
    ref = oracle_conv(x, wt, act1=1)
    assert rel_err(host(y), ref) <= TOL and np.array_equal(host(yp), ops.max_pool(host(y), 2, 2))
    assert torch.equal(K.conv2d_maxpool2(dev(x), dev(wt), None, act1=K.ACT_RELU, keep_y=False), yp)      # y never stored
    assert torch.equal(K.conv2d(dev(x), dev(wt), None, act1=K.ACT_RELU), y)


AVGPOOL_CASES = [  # name, n, h, w, cin, cout, k, expected plan
    ("x3n_7x7_16_16", 2, 32, 48, 16, 16, 7, "x3n"), ("x3n_5x5_32_32_ragged_tiles", 1, 38, 26, 32, 32, 5, "x3n"),
    ("x3n_3x3_16_32", 1, 16, 16, 16, 32, 3, "x3n"), ("x3_3x3_64_64", 2, 32, 48, 64, 64, 3, "x3"),
    ("x3_3x3_128_128_ragged_tiles", 1, 22, 38, 128, 128, 3, "x3"), ("fallback_3x3_48_48", 1, 12, 20, 48, 48, 3, None),
    ("x3_3x3_64_32_half_slice", 2, 22, 38, 64, 32, 3, "x3"),        # a 32-cout layer on one 64-cout slice (deq / ref u2.conv1)
]


@pytest.mark.parametrize("case", AVGPOOL_CASES, ids=[c[0] for c in AVGPOOL_CASES])
def test_conv2d_avgpool2_pair(shdr, case, monkeypatch):
    """conv + AveragePooling2D(2) pairs of the U-Net encoders (dequantization_net.py:9-13): the split-operand kernels write the
    pooled tensor from their own epilogue (desc.pool = SHDR_POOL_AVG), every other plan pools in a second launch below the ABI.
    y is the plain call's y bit for bit, and the pooled tensor is avgpool2(y) bit for bit (same order of additions)."""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    name, n, h, w, cin, cout, k, plan = case
    K = shdr._ops
    rng = np.random.default_rng(len(name) + h)
    x = f32(rng.normal(size=(n, h, w, cin)))
    wt = f32(rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin))
    b = f32(rng.normal(size=cout) * 0.1)
    if plan is not None:
        assert K.conv2d_plan((n, h, w, cin), wt.shape) == plan
    y, yp = K.conv2d_avgpool2(dev(x), dev(wt), dev(b), act1=K.ACT_LRELU)
    assert rel_err(host(y), oracle_conv(x, wt, b, act1=2)) <= TOL
    assert torch.equal(y, K.conv2d(dev(x), dev(wt), dev(b), act1=K.ACT_LRELU))
    assert tuple(yp.shape) == (n, h // 2, w // 2, cout) and torch.equal(yp, K.avgpool2(y))
    np.testing.assert_allclose(host(yp), ops.avg_pool2(host(y).astype(np.float64)), rtol=1e-6, atol=1e-7)


def test_conv2d_x3_thirty_two_couts(shdr, monkeypatch):
    """the 64 -> 32 and 32 + 32 -> 32 decoder layers of the U-Nets (dequantization_net.py:17-29) on the split-operand kernel: one 64-cout
    slice whose upper half is zero filter columns, neither biased nor stored; plain, two sources, behind the bilinear prologue, input
    gradient -- vs the float64 oracle at the exact kernels' bar"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    K = shdr._ops
    rng = np.random.default_rng(77)
    n, h, w = 2, 26, 38
    x = f32(rng.normal(size=(n, h, w, 64)))
    wt = f32(rng.normal(size=(3, 3, 64, 32)) / 24)
    b = f32(rng.normal(size=32) * 0.1)
    assert K.conv2d_plan((n, h, w, 64), wt.shape) == "x3"
    y = K.conv2d(dev(x), dev(wt), dev(b), act1=K.ACT_LRELU)
    assert tuple(y.shape) == (n, h, w, 32) and rel_err(host(y), oracle_conv(x, wt, b, act1=2)) <= TOL
    assert float(y._shdr_range) == float(y.abs().max())
    xa, xb = f32(rng.normal(size=(n, h, w, 32))), f32(rng.normal(size=(n, h, w, 32)))
    assert K.conv2d_plan((n, h, w, 32), wt.shape, c2=32) == "x3"
    y2 = K.conv2d(dev(xa), dev(wt), dev(b), x2=dev(xb), act1=K.ACT_LRELU)
    assert rel_err(host(y2), oracle_conv(np.concatenate([xa, xb], -1), wt, b, act1=2)) <= TOL
    lo = f32(rng.normal(size=(n, h // 2, w // 2, 64)))
    with torch.no_grad():
        yu = K.conv2d_up2(dev(lo), dev(wt), dev(b), act1=K.ACT_LRELU)
        want = K.conv2d(K.resize2x(dev(lo)), dev(wt), dev(b), act1=K.ACT_LRELU)
    assert float((yu - want).abs().max()) <= 1e-5 * float(want.abs().max())
    # the input gradient of a 32 -> 64 layer is a 64 -> 32 convolution with the flipped filter: the same kernel
    wf = f32(rng.normal(size=(3, 3, 32, 64)) / 17)
    dz = f32(rng.normal(size=(n, h, w, 64)))
    dx = K.conv2d_dgrad(dev(dz), dev(wf), (n, h, w, 32), 32, 0, 0)
    assert rel_err(host(dx), oracle_conv(dz, np.ascontiguousarray(wf[::-1, ::-1].transpose(0, 1, 3, 2)))) <= TOL


def test_conv2d_x3_residual_joins(shdr, monkeypatch):
    """the ResNet joins of the Linearization-Net (linearization_net.py:6-48: relu(norm(conv1x1) + shortcut)) on the split-operand kernel:
    folded BatchNorm, residual added between the affine and the relu; 64 -> 256, 128 -> 512 and a 3x3 layer with a wider residual tensor"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    K = shdr._ops
    rng = np.random.default_rng(31)
    for (n, h, w, cin, cout, k, res_c) in ((2, 40, 56, 64, 256, 1, 256), (1, 24, 40, 128, 512, 1, 512), (1, 22, 38, 64, 64, 3, 128)):
        x = f32(rng.normal(size=(n, h, w, cin)))
        wt = f32(rng.normal(size=(k, k, cin, cout)) / np.sqrt(k * k * cin))
        sc, sh = f32(rng.uniform(0.5, 1.5, size=cout)), f32(rng.normal(size=cout))
        res = f32(rng.normal(size=(n, h, w, res_c)))
        assert K.conv2d_plan((n, h, w, cin), wt.shape, has_residual=True) == "x3"
        y = K.conv2d(dev(x), dev(wt), None, scale=dev(sc), shift=dev(sh), residual=dev(res), act2=K.ACT_RELU)
        ref = np.maximum(oracle_conv(x, wt) * sc + sh + res[..., :cout], 0.0)
        assert rel_err(host(y), ref) <= TOL and float(y._shdr_range) == float(y.abs().max())
        monkeypatch.setenv("SHDR_NO_X3_RESIDUAL", "1")
        assert K.conv2d_plan((n, h, w, cin), wt.shape, has_residual=True) != "x3"
        y_exact = K.conv2d(dev(x), dev(wt), None, scale=dev(sc), shift=dev(sh), residual=dev(res), act2=K.ACT_RELU)
        monkeypatch.delenv("SHDR_NO_X3_RESIDUAL")
        assert float((y - y_exact).abs().max()) <= 1e-5 * float(y_exact.abs().max())


@pytest.mark.parametrize("shape", [(2, 40, 56, 256, 512), (1, 33, 47, 256, 128), (1, 18, 22, 64, 64)])
def test_conv2d_x3_1x1_stride2_projections(shdr, shape, monkeypatch):
    """the 1x1 / stride-2 projections of the ResNet blocks (linearization_net.py:6-48, res4) on the split-operand kernel: the 1x1 kernel on
    every other input pixel (TF SAME: no padding, ceil(H / 2) outputs); even and odd sizes, folded BatchNorm, vs the float64 oracle"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape))
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, cin)))
    wt = f32(rng.normal(size=(1, 1, cin, cout)) / np.sqrt(cin))
    sc, sh = f32(rng.uniform(0.5, 1.5, cout)), f32(rng.normal(size=cout))
    assert K.conv2d_plan((n, h, w, cin), wt.shape, stride=2) == "x3"
    y = K.conv2d(dev(x), dev(wt), None, stride=2, scale=dev(sc), shift=dev(sh), act2=K.ACT_RELU)
    ref = oracle_conv(x, wt, None, stride=2, scale=sc, shift=sh, act2=1)
    assert tuple(y.shape) == ref.shape == (n, (h + 1) // 2, (w + 1) // 2, cout) and rel_err(host(y), ref) <= TOL


X3N_CASES = [  # name, n, h, w, c1, c2, cout (filter width), cout_valid, k, act1, residual
    ("deq_conv1_7x7_4_16", 1, 40, 52, 4, 0, 16, 16, 7, 2, False), ("deq_conv2_7x7_16_16", 2, 33, 47, 16, 0, 16, 16, 7, 2, False),
    ("deq_d2_5x5_16_32", 1, 24, 40, 16, 0, 32, 32, 5, 2, False), ("deq_u1_3x3_32_16", 1, 50, 34, 32, 0, 16, 16, 3, 2, False),
    ("deq_u1_concat_16_16_16", 2, 21, 35, 16, 16, 16, 16, 3, 2, False), ("deq_out_3x3_16_3_tanh_residual", 1, 36, 36, 16, 0, 16, 3, 3, 3, True),
    ("ref_conv1_7x7_12_16", 1, 20, 28, 12, 0, 16, 16, 7, 2, False), ("d3_3x3_32_32", 1, 18, 18, 32, 0, 32, 32, 3, 1, True),
    ("hal_conv1_1_3x3_4_64_relu", 2, 40, 52, 4, 0, 64, 64, 3, 1, False),       # the 3 -> 4 channel image layers: 64 couts (NT = 4)
]


@pytest.mark.parametrize("case", X3N_CASES, ids=[c[0] for c in X3N_CASES])
def test_conv2d_x3n_narrow_layers(shdr, case, monkeypatch):
    """SHDR_PLAN_X3N (csrc/conv_x3n.hip): the narrow full-resolution layers of the Dequantization- / Refinement-Net on the split-operand
    arithmetic (whole filter in LDS, 32 / CT taps per MFMA): zero-padded input channels (3 -> 4, 9 -> 12), two sources, 3-channel head
    with tanh + residual, ragged tiles; vs the float64 oracle at the exact-fp32 kernels' bar and next to the exact kernel's error"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    name, n, h, w, c1, c2, cout, cv, k, act, with_res = case
    rng = np.random.default_rng(len(name) + h)
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, c1)))
    x2 = f32(rng.normal(size=(n, h, w, c2))) if c2 else None
    wt = f32(rng.normal(size=(k, k, c1 + c2, cout)) / np.sqrt(k * k * (c1 + c2)))
    wt[..., cv:] = 0.0
    b = f32(rng.normal(size=cv))
    res = f32(rng.normal(size=(n, h, w, cv))) if with_res else None
    assert K.conv2d_plan((n, h, w, c1), wt.shape, c2=c2, has_residual=with_res, cout_valid=cv) == "x3n"
    ref = oracle_conv(x, wt[..., :cv], b, x2=x2, act1=act, residual=res)
    kw = dict(x2=None if x2 is None else dev(x2), act1=act, residual=None if res is None else dev(res), cout_valid=cv)
    y = K.conv2d(dev(x), dev(wt), dev(b), **kw)
    err = rel_err(host(y), ref)
    monkeypatch.setattr(K, "EXACT_FP32", True)
    err_exact = rel_err(host(K.conv2d(dev(x), dev(wt), dev(b), **kw)), ref)
    assert tuple(y.shape) == ref.shape and err <= TOL and err <= 4 * err_exact + 2e-7, (err, err_exact)


@pytest.mark.parametrize("case", [(2, 33, 47, 16, 0, 16, 7, 0), (1, 24, 40, 16, 0, 32, 5, 0), (1, 21, 35, 16, 16, 16, 3, 0), (1, 21, 35, 16, 16, 16, 3, 1),
                                  (1, 30, 30, 32, 0, 16, 3, 0)])
def test_conv2d_x3n_input_gradient(shdr, case, monkeypatch):
    """input gradient of the narrow layers through shdr_conv2d_dgrad_f32 on the split-operand kernel (transposed filter in LDS, output
    gradients range-scaled in the kernel): tiny, heavy-tailed dz against float64 autograd, L2-relative"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    n, h, w, c1, c2, cout, k, which = case
    K = shdr._ops
    g = torch.Generator().manual_seed(sum(case))
    wt = torch.randn((k, k, c1 + c2, cout), generator=g) / (k * (c1 + c2) ** 0.5)
    dz = (torch.randn((n, h, w, cout), generator=g) * torch.exp(2.0 * torch.randn((n, h, w, cout), generator=g)) * 1e-6)
    x = torch.zeros((n, c1 + c2, h, w), dtype=torch.float64, requires_grad=True)
    y = torch.nn.functional.conv2d(x, wt.double().permute(3, 2, 0, 1), padding=k // 2)
    y.backward(dz.double().permute(0, 3, 1, 2))
    ref = x.grad.permute(0, 2, 3, 1)[..., (c1 if which else 0):(c1 + c2 if which else c1)]
    cx = c2 if which else c1
    dx = K.conv2d_dgrad(dz.cuda(), wt.cuda(), (n, h, w, cx), c1, c2, which)
    err = float((dx.double().cpu() - ref).norm() / ref.norm())
    monkeypatch.setattr(K, "EXACT_FP32", True)
    dxe = K.conv2d_dgrad(dz.cuda(), wt.cuda(), (n, h, w, cx), c1, c2, which)
    err_exact = float((dxe.double().cpu() - ref).norm() / ref.norm())
    assert err <= 2e-6 and err <= 4 * err_exact + 2e-7, (err, err_exact)
    assert not torch.equal(dx, dxe)                     # ... and it did take another kernel


@pytest.mark.parametrize("shape", [(2, 40, 56, 128, 128, 128), (1, 33, 47, 256, 0, 64), (1, 16, 16, 512, 512, 512), (3, 20, 20, 256, 0, 128),
                                   (1, 24, 40, 96, 0, 64), (2, 18, 30, 96, 64, 128), (1, 20, 28, 64, 0, 256)])      # (3, 5 and 2 chunks: the two-deep prefetch)
def test_conv2d_x3_1x1_layers(shdr, shape, monkeypatch):
    """1x1 layers (K >= 64) on the split-operand kernel (the hal skip layers on tf.concat with the 1/255 skip scale, the ResNet
    bottleneck convs): one tap per chunk; vs the float64 oracle at the exact-fp32 bar, ragged tiles"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    n, h, w, c1, c2, cout = shape
    rng = np.random.default_rng(sum(shape) + 9)
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, c1)))
    x2 = f32(rng.normal(size=(n, h, w, c2)) * 255.0) if c2 else None
    x2s = 1.0 / 255 if c2 else 1.0
    wt = f32(rng.normal(size=(1, 1, c1 + c2, cout)) / np.sqrt(c1 + c2))
    b, sc, sh = f32(rng.normal(size=cout)), f32(rng.uniform(0.5, 1.5, cout)), f32(rng.normal(size=cout))
    assert K.conv2d_plan((n, h, w, c1), wt.shape, c2=c2, x2_scale=x2s) == "x3"
    ref = oracle_conv(x, wt, b, x2=x2, x2_scale=x2s, act1=1, scale=sc, shift=sh, act2=1)
    y = K.conv2d(dev(x), dev(wt), dev(b), x2=None if x2 is None else dev(x2), x2_scale=x2s, act1=K.ACT_RELU, scale=dev(sc), shift=dev(sh),
                 act2=K.ACT_RELU)
    assert tuple(y.shape) == ref.shape and rel_err(host(y), ref) <= TOL
    # with a fused residual too (test_conv2d_x3_residual_joins)
    res = f32(rng.normal(size=(n, h, w, cout)))
    assert K.conv2d_plan((n, h, w, c1), wt.shape, c2=c2, x2_scale=x2s, has_residual=True) == "x3"
    yr = K.conv2d(dev(x), dev(wt), dev(b), x2=None if x2 is None else dev(x2), x2_scale=x2s, act1=K.ACT_RELU, scale=dev(sc), shift=dev(sh),
                  residual=dev(res), act2=K.ACT_RELU)
    assert rel_err(host(yr), oracle_conv(x, wt, b, x2=x2, x2_scale=x2s, act1=1, scale=sc, shift=sh, residual=res, act2=1)) <= TOL


@pytest.mark.parametrize("shape", [(2, 64, 96, 96, 64), (1, 61, 75, 96, 64), (1, 32, 34, 32, 128), (1, 17, 16, 64, 64)])
def test_conv2d_x3_stride2_stem_as_four_phases(shdr, shape, monkeypatch):
    """the 7x7 / stride-2 stem of the Linearization-Net (linearization_net.py:91) on the split-operand kernel: four stride-1 phase
    launches (4x4, 4x3, 3x4, 3x3 sub-filters over the parity-subsampled input) accumulating in y, folded-BN epilogue in the last;
    even and odd sizes (TF SAME: pad 2 / 3 resp. 3 / 3), vs the float64 oracle at the exact-fp32 kernels' bar"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape) + 3)
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, cin)))
    wt = f32(rng.normal(size=(7, 7, cin, cout)) / np.sqrt(49 * cin))
    b, sc, sh = f32(rng.normal(size=cout)), f32(rng.uniform(0.5, 1.5, cout)), f32(rng.normal(size=cout))
    assert K.conv2d_plan((n, h, w, cin), wt.shape, stride=2) == "x3"
    ref = oracle_conv(x, wt, b, stride=2, scale=sc, shift=sh, act2=1)
    y = K.conv2d(dev(x), dev(wt), dev(b), stride=2, scale=dev(sc), shift=dev(sh), act2=K.ACT_RELU)
    assert tuple(y.shape) == ref.shape and rel_err(host(y), ref) <= TOL
    monkeypatch.setattr(K, "EXACT_FP32", True)
    assert K.conv2d_plan((n, h, w, cin), wt.shape, stride=2) == "mfma"
    err_exact = rel_err(host(K.conv2d(dev(x), dev(wt), dev(b), stride=2, scale=dev(sc), shift=dev(sh), act2=K.ACT_RELU)), ref)
    assert rel_err(host(y), ref) <= 4 * err_exact + 2e-7


@pytest.mark.parametrize("shape", [(2, 24, 40, 64, 128), (1, 13, 19, 32, 64), (1, 8, 8, 256, 64), (1, 1, 3, 32, 64)])
def test_conv2d_x3_up2_prologue(shdr, shape, monkeypatch):
    """bilinear 2x fused into the split-operand kernel's patch loader (low-res patch parked in LDS, up-sampled on the way into the
    fp16 images): vs the float64 oracle and BIT-IDENTICAL to resize2x + the same kernel (the expansion repeats resize2x_kernel's
    arithmetic); ragged tiles, clamped borders, a 1 x 3 source"""
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    monkeypatch.setenv("SHDR_X3_UP_ALWAYS", "1")        # the plan fuses the prologue up to 64 couts only (speed); here every case
    n, h, w, cin, cout = shape
    rng = np.random.default_rng(sum(shape) + 5)
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, cin)))
    wt = f32(rng.normal(size=(3, 3, cin, cout)) / np.sqrt(9 * cin))
    b, sc, sh = f32(rng.normal(size=cout)), f32(rng.uniform(0.5, 1.5, cout)), f32(rng.normal(size=cout))
    assert K.conv2d_plan((n, 2 * h, 2 * w, cin), wt.shape) == "x3"
    up = ops.resize_bilinear_2x(x.astype(np.float64))
    ref = oracle_conv(up, wt, b, act1=1, scale=sc, shift=sh, act2=1)
    y = K.conv2d_up2(dev(x), dev(wt), dev(b), act1=K.ACT_RELU, scale=dev(sc), shift=dev(sh), act2=K.ACT_RELU)
    assert tuple(y.shape) == ref.shape and rel_err(host(y), ref) <= TOL
    two = K.conv2d(K.resize2x(dev(x)), dev(wt), dev(b), act1=K.ACT_RELU, scale=dev(sc), shift=dev(sh), act2=K.ACT_RELU)
    assert torch.equal(y, two)


def test_conv2d_up2_under_tape_is_two_recorded_ops(shdr):
    """with a gradient tape the fused prologue is not taken: resize and conv are recorded, the gradient reaches x and w"""
    K = shdr._ops
    rng = np.random.default_rng(3)
    x = dev(rng.normal(size=(1, 4, 6, 32))).requires_grad_(True)
    wt = dev(rng.normal(size=(3, 3, 32, 64)) / 17.0).requires_grad_(True)
    y = K.conv2d_up2(x, wt, act1=K.ACT_RELU)
    y.sum().backward()
    assert x.grad is not None and wt.grad is not None and float(wt.grad.abs().sum()) > 0
    with torch.no_grad():
        assert torch.equal(K.conv2d_up2(x, wt, act1=K.ACT_RELU), y.detach())


@pytest.mark.parametrize("shape", [(2, 13, 21, 32, 64), (1, 16, 32, 64, 128), (1, 9, 9, 8, 64), (1, 32, 16, 128, 64)])
def test_conv2d_winograd_fused_two_sources(shdr, shape):
    """the fused Winograd kernel on a channel concatenation [x, x2] (skip connections of the U-Net decoders): against the
    oracle on the concatenated tensor, against the single-source kernel on a materialised concatenation (same chunk order:
    bit-identical), and as the AUTO dispatch of conv2d(x, w, x2=...)"""
    n, h, w, c, cout = shape
    rng = np.random.default_rng(sum(shape))
    K = shdr._ops
    x, x2 = f32(rng.normal(size=(n, h, w, c))), f32(rng.normal(size=(n, h, w, c)))
    wt = f32(rng.normal(size=(3, 3, 2 * c, cout)) / np.sqrt(18 * c))
    b = f32(rng.normal(size=cout))
    u = K.winograd_filter_packed(dev(wt))
    xc = np.concatenate([x, x2], -1)
    y = K.conv2d_winograd_fused(dev(x), u, dev(b), act1=K.ACT_LRELU, x2=dev(x2))
    assert rel_err(host(y), oracle_conv(xc, wt, b, act1=2)) <= TOL
    assert torch.equal(y, K.conv2d_winograd_fused(dev(xc), u, dev(b), act1=K.ACT_LRELU))
    if 2 * c >= 32:
        assert torch.equal(K.conv2d(dev(x), dev(wt), dev(b), x2=dev(x2), act1=K.ACT_LRELU), y)
    with pytest.raises(ValueError, match="shape of x"):
        K.conv2d_winograd_fused(dev(x), u, x2=dev(x2[..., :c // 2]))


@pytest.mark.parametrize("shape", [(2, 16, 32, 32, 64), (1, 10, 22, 64, 128), (1, 2, 2, 8, 64), (1, 34, 18, 40, 64)])
def test_conv2d_winograd_fused_maxpool(shdr, shape):
    """MaxPool2D(2) written by the epilogue of the fused Winograd kernel (a Winograd tile is one pooling window): the conv
    output is unchanged bit for bit, the pooled tensor equals maxpool2 of it exactly; ragged block tiles, even H and W"""
    n, h, w, c, cout = shape
    rng = np.random.default_rng(sum(shape))
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, c)))
    wt = f32(rng.normal(size=(3, 3, c, cout)) / np.sqrt(9 * c))
    b = f32(rng.normal(size=cout))
    u = K.winograd_filter_packed(dev(wt))
    y, yp = K.conv2d_winograd_fused(dev(x), u, dev(b), act1=K.ACT_RELU, pool=True)
    y0 = K.conv2d_winograd_fused(dev(x), u, dev(b), act1=K.ACT_RELU)
    assert torch.equal(y, y0) and tuple(yp.shape) == (n, h // 2, w // 2, cout)
    assert torch.equal(yp, K.maxpool2(y0))
    assert torch.equal(K.conv2d_winograd_fused(dev(x), u, dev(b), act1=K.ACT_RELU, pool="only"), yp)    # y never stored
    if c >= 32:
        y2, yp2 = K.conv2d_maxpool2(dev(x), dev(wt), dev(b), act1=K.ACT_RELU)       # the dispatching wrapper
        assert torch.equal(y2, y) and torch.equal(yp2, yp)
    with pytest.raises(ValueError, match="even"):
        K.conv2d_winograd_fused(dev(x[:, :h - 1]), u, pool=True)


def test_conv2d_winograd_fused_tall_tile(shdr, monkeypatch):
    """the 16 x 16-pixel block tile of the fused kernel (experiment switch SHDR_WINOGRAD_TILE=16: one block per CU, 128
    accumulator registers, two raw DMA instructions per wave and chunk) against the oracle and the default 8 x 16 tile"""
    rng = np.random.default_rng(77)
    K = shdr._ops
    x = f32(rng.normal(size=(2, 19, 35, 40)))
    wt = f32(rng.normal(size=(3, 3, 40, 64)) / np.sqrt(360))
    b = f32(rng.normal(size=64))
    u = K.winograd_filter_packed(dev(wt))
    y8 = K.conv2d_winograd_fused(dev(x), u, dev(b), act1=K.ACT_RELU)
    monkeypatch.setenv("SHDR_WINOGRAD_TILE", "16")
    y16 = K.conv2d_winograd_fused(dev(x), u, dev(b), act1=K.ACT_RELU)
    monkeypatch.delenv("SHDR_WINOGRAD_TILE")
    assert rel_err(host(y16), oracle_conv(x, wt, b, act1=1)) <= TOL
    assert rel_err(host(y16), host(y8)) <= 1e-6


@pytest.mark.parametrize("shape", [  # n, h, w, c1, c2, cout, k, cout_valid
    (2, 20, 23, 16, 0, 16, 7, None),     # Dequantization-Net conv2
    (1, 33, 18, 4, 0, 16, 7, None),      # conv1 on the 3-channel image padded to 4
    (2, 16, 16, 12, 0, 16, 7, None),     # Refinement-Net conv1 on [A, B, C] padded to 12
    (1, 19, 21, 16, 0, 32, 5, None),     # d2.conv1
    (1, 18, 17, 32, 0, 32, 5, None),     # d2.conv2 (100 KB of filter in LDS)
    (2, 17, 31, 32, 0, 16, 3, None),     # u1.conv1
    (2, 16, 20, 16, 16, 16, 3, None),    # u1.conv2: concatenated sources
    (1, 24, 16, 16, 0, 16, 3, 3),        # output head: 3 of 16 padded couts stored
    (2, 21, 34, 4, 0, 64, 3, None),      # VGG-shaped conv1_1 (Hallucination-Net, VGG16) on the 3-channel image padded to 4
    (1, 8, 8, 8, 0, 16, 3, None)])
def test_conv2d_register_a_kernel_parity(shdr, shape):
    """conv_rega_kernel (narrow U-Net layers: activations global -> VGPR, filter resident in LDS) vs the float64 oracle and,
    bit for bit where the summation order coincides, vs the LDS-DMA kernel it replaces in AUTO mode"""
    n, h, w, c1, c2, cout, k, cv = shape
    rng = np.random.default_rng(sum(x or 0 for x in shape))
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, c1)))
    x2 = f32(rng.normal(size=(n, h, w, c2))) if c2 else None
    wt = f32(rng.normal(size=(k, k, c1 + c2, cout)) / np.sqrt(k * k * (c1 + c2)))
    if cv:
        wt[..., cv:] = 0
    b = f32(rng.normal(size=cv or cout))
    res = f32(rng.normal(size=(n, h, w, cv or cout)))
    kw = dict(x2=None if x2 is None else dev(x2), x2_scale=0.5 if c2 else 1.0, act1=K.ACT_LRELU, residual=dev(res), act2=K.ACT_RELU,
              cout_valid=cv)
    y = K.conv2d(dev(x), dev(wt), dev(b), **kw)                                    # AUTO -> register-A kernel
    ref = oracle_conv(x if x2 is None else np.concatenate([x, 0.5 * x2], -1), wt[..., :cv] if cv else wt, b, act1=2)
    ref = np.maximum(ref + res, 0)
    assert tuple(y.shape) == ref.shape and rel_err(host(y), ref) <= TOL
    old = K.conv2d(dev(x), dev(wt), dev(b), algo=K.ALGO_MFMA, **kw)               # forced: the LDS-DMA / register-staged kernel
    assert rel_err(host(y), host(old)) <= 5e-6
