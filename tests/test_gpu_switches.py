"""The experiment switches of README.md select alternative kernels for the same operation: each one must give the result of
the default path (bit for bit where the summation order is the same, within the per-layer tolerance otherwise).  The
library reads the environment per launch, so the switches can be flipped inside one process."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def maxrel(a, b):
    return float((a - b).abs().max() / b.abs().max())


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).cuda()


@pytest.mark.parametrize("var,value", [("SHDR_NO_REGA", "1"), ("SHDR_REGA_NO_DPP", "1"), ("SHDR_REGA_PER_CU", "2")])
def test_register_a_switches(shdr, monkeypatch, var, value):
    K = shdr._ops
    for cin, c2, cout, k in [(16, 0, 16, 7), (4, 0, 16, 7), (16, 16, 16, 3), (16, 0, 32, 5)]:
        x, x2 = _rand(2, 21, 35, cin, seed=1), (_rand(2, 21, 35, c2, seed=2) if c2 else None)
        w, b = _rand(k, k, cin + c2, cout, seed=3, scale=0.05), _rand(cout, seed=4)
        ref = K.conv2d(x, w, b, x2=x2, act1=K.ACT_LRELU)
        monkeypatch.setenv(var, value)
        got = K.conv2d(x, w, b, x2=x2, act1=K.ACT_LRELU)
        monkeypatch.delenv(var)
        assert maxrel(got, ref) <= 5e-6, (var, cin, c2, cout, k)


def test_register_a_64_cout_switch(shdr, monkeypatch):
    K = shdr._ops
    x, w, b = _rand(2, 18, 34, 4, seed=5), _rand(3, 3, 4, 64, seed=6, scale=0.1), _rand(64, seed=7)
    ref = K.conv2d(x, w, b, act1=K.ACT_RELU)
    monkeypatch.setenv("SHDR_NO_REGA64", "1")
    got = K.conv2d(x, w, b, act1=K.ACT_RELU)
    assert maxrel(got, ref) <= 5e-6


def test_legacy_epilogue_switch_is_bit_identical(shdr, monkeypatch):
    K = shdr._ops
    x, w, b = _rand(1, 19, 23, 64, seed=8), _rand(1, 1, 64, 128, seed=9, scale=0.1), _rand(128, seed=10)
    res = _rand(1, 19, 23, 128, seed=11)
    ref = K.conv2d(x, w, b, act1=K.ACT_RELU, residual=res, act2=K.ACT_RELU)
    monkeypatch.setenv("SHDR_CONV_LEGACY_EPILOGUE", "1")
    assert torch.equal(K.conv2d(x, w, b, act1=K.ACT_RELU, residual=res, act2=K.ACT_RELU), ref)   # same accumulators, other stores


@pytest.mark.parametrize("var,shape", [("SHDR_NO_ALLTAPS", (2, 20, 33, 16, 16, 7, 1)), ("SHDR_NO_ALLTAPS", (1, 16, 40, 32, 16, 3, 1)),
                                       ("SHDR_NO_WGRAD96", (2, 30, 34, 96, 64, 7, 2))])
def test_weight_gradient_switches(shdr, monkeypatch, var, shape):
    K = shdr._ops
    n, h, w, cin, cout, k, s = shape
    x = _rand(n, h, w, cin, seed=12)
    dz = _rand(n, -(-h // s), -(-w // s), cout, seed=13)
    ref = K.conv2d_wgrad(x, None, dz, (k, k, cin, cout), s)
    monkeypatch.setenv(var, "1")
    got = K.conv2d_wgrad(x, None, dz, (k, k, cin, cout), s)
    assert maxrel(got, ref) <= 2e-5          # atomically accumulated partial sums in another order


def test_batchnorm_scalar_switch(shdr, monkeypatch):
    K = shdr._ops
    x, dy = _rand(3, 17, 19, 64, seed=14) * 2.0 + 0.5, _rand(3, 17, 19, 64, seed=15)
    g, b = torch.rand(64).cuda() + 0.5, _rand(64, seed=16)
    mean, var = K.bn_stats(x)
    y = K.bn_train_apply(x, mean, var, g, b, 1e-3, True)
    ref = K.bn_bwd(dy, x, y, mean, var, g, 1e-3)
    monkeypatch.setenv("SHDR_BN_SCALAR", "1")
    m2, v2 = K.bn_stats(x)
    got = K.bn_bwd(dy, x, y, mean, var, g, 1e-3)
    assert maxrel(m2, mean) <= 1e-6 and maxrel(v2, var) <= 1e-6
    for a, r in zip(got, ref):
        assert maxrel(a, r) <= 1e-5


def test_frontend_quad_switch_is_bit_identical(shdr, monkeypatch):
    K = shdr._ops
    img = torch.rand(2, 33, 47, 3, generator=torch.Generator().manual_seed(17)).cuda()
    refs = [K.soft_hist(img, 32), K.soft_hist(img, 12), K.lin_frontend(img, 96)]
    monkeypatch.setenv("SHDR_FRONTEND_QUADS", "1")
    gots = [K.soft_hist(img, 32), K.soft_hist(img, 12), K.lin_frontend(img, 96)]
    for a, r in zip(gots, refs):
        assert torch.equal(a, r)


@pytest.mark.parametrize("var,shape", [
    ("SHDR_NO_X3", (2, 48, 64, 64, 0, 128, 3, 1)), ("SHDR_NO_X3", (1, 40, 40, 64, 64, 64, 3, 1)), ("SHDR_NO_X3_1X1", (1, 40, 56, 256, 256, 128, 1, 1)),
    ("SHDR_NO_X3_STRIDE2", (1, 64, 80, 96, 0, 64, 7, 2)), ("SHDR_NO_X3N", (1, 40, 56, 16, 0, 16, 7, 1)), ("SHDR_NO_X3N", (1, 40, 56, 16, 16, 16, 3, 1)),
    ("SHDR_NO_WINOGRAD", (1, 48, 48, 64, 0, 64, 3, 1)), ("SHDR_NO_X3_COUT32", (2, 40, 56, 64, 0, 32, 3, 1)),
    ("SHDR_NO_X3_COUT32", (1, 40, 56, 32, 32, 32, 3, 1))])
def test_split_operand_switches(shdr, monkeypatch, var, shape):
    """the split-operand fp16 kernels (plans "x3" / "x3n") against the exact-fp32 kernel the switch falls back to: both fp32-grade"""
    K = shdr._ops
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    n, h, w, c1, c2, cout, k, s = shape
    x, x2 = _rand(n, h, w, c1, seed=20), (_rand(n, h, w, c2, seed=21) if c2 else None)
    wt, b = _rand(k, k, c1 + c2, cout, seed=22, scale=1.0 / (k * (c1 + c2) ** 0.5)), _rand(cout, seed=23)
    plan = K.conv2d_plan((n, h, w, c1), tuple(wt.shape), c2=c2, stride=s)
    assert plan in ("x3", "x3n")
    got = K.conv2d(x, wt, b, stride=s, x2=x2, act1=K.ACT_RELU)
    monkeypatch.setenv(var, "1")
    assert K.conv2d_plan((n, h, w, c1), tuple(wt.shape), c2=c2, stride=s) not in ("x3", "x3n")
    ref = K.conv2d(x, wt, b, stride=s, x2=x2, act1=K.ACT_RELU)
    assert maxrel(got, ref) <= 5e-6, (var, plan)


def test_stem_phase_launches_switch(shdr, monkeypatch):
    """the 7x7 / stride-2 stem in one launch (partial sums of the four parity phases in registers) vs four launches accumulating in y:
    the same products in the same order within a phase, the phases summed in another order"""
    K = shdr._ops
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    for n, h, w in ((1, 64, 80), (2, 70, 46)):               # (ragged tiles, odd output sizes in the second case)
        x, wt, b = _rand(n, h, w, 96, seed=30), _rand(7, 7, 96, 64, seed=31, scale=1.0 / (7 * 96 ** 0.5)), _rand(64, seed=32)
        assert K.conv2d_plan((n, h, w, 96), tuple(wt.shape), stride=2) == "x3"
        one = K.conv2d(x, wt, b, stride=2, act1=K.ACT_RELU)
        monkeypatch.setenv("SHDR_X3_STEM_PHASE_LAUNCHES", "1")
        four = K.conv2d(x, wt, b, stride=2, act1=K.ACT_RELU)
        monkeypatch.delenv("SHDR_X3_STEM_PHASE_LAUNCHES")
        assert maxrel(one, four) <= 2e-6


def test_x3_up_always_switch_is_bit_identical(shdr, monkeypatch):
    """the bilinear prologue fused into the split kernel at 512 couts (the plan fuses it up to 256) = resize2x + the same kernel"""
    K = shdr._ops
    monkeypatch.setenv("SHDR_X3_MIN_BLOCKS", "1")
    x, wt, b = _rand(1, 12, 20, 64, seed=24), _rand(3, 3, 64, 512, seed=25, scale=0.04), _rand(512, seed=26)
    ref = K.conv2d_up2(x, wt, b, act1=K.ACT_RELU)
    monkeypatch.setenv("SHDR_X3_UP_ALWAYS", "1")
    assert torch.equal(K.conv2d_up2(x, wt, b, act1=K.ACT_RELU), ref)


@pytest.mark.parametrize("var", ["SHDR_NO_W3", "SHDR_NO_PATCH"])
def test_fp16_specialised_conv_switches(shdr, monkeypatch, var):
    """native-fp16 path: the 3x3 raw-patch kernel / the narrow-layer patch kernel vs the general implicit-GEMM kernel (same products,
    fp32 accumulation in another order, fp16 output rounding)"""
    K = shdr._ops
    monkeypatch.setenv("SHDR_W3_MIN_BLOCKS", "1")
    if var == "SHDR_NO_W3":
        x, wt = _rand(2, 32, 48, 64, seed=27).half(), _rand(3, 3, 64, 64, seed=28, scale=0.04)
    else:
        x, wt = _rand(2, 32, 48, 16, seed=27).half(), _rand(7, 7, 16, 16, seed=28, scale=0.04)
    b = _rand(wt.shape[3], seed=29)
    with K.precision("fp16"):
        got = K.conv2d(x, wt, b, act1=K.ACT_RELU)
        monkeypatch.setenv(var, "1")
        ref = K.conv2d(x, wt, b, act1=K.ACT_RELU)
    assert got.dtype == torch.float16 and maxrel(got.float(), ref.float()) <= 2e-3
