"""GPU parity of the backward / training kernels against the float64 autograd reference
(tests/torch_ref.py, itself pinned to the NumPy oracle by tests/test_oracle_grad.py).
Tolerance: <= 1e-4 tensor-relative per gradient tensor (fp32 accumulation, atomics)."""
import numpy as np
import pytest
import torch

import torch_ref as R
from conftest import quantised_image, rel_err
from oracle import nets

pytestmark = pytest.mark.gpu
TOL = 1e-4
# Whole-net gradients of the randomly initialised training-mode nets (measured on MI355X, tools/dbg/grad_err.py): the flat gradient
# of the Hallucination-Net is within 2-4e-4 (relative L2) of the float64 reference, its worst single variable within 1.2e-3; the
# Linearization-Net 2e-4 .. 3.4e-3 / 7.5e-3 depending on the seed -- its head routes a gradient through min() (linearization_net.py:
# 376-380), which jumps when two neighbouring slopes of the predicted curve tie.  Bars: NET_L2_TOL per variable, WHOLE_* on the
# flat gradient (round 1 stated 5e-2 for both).
NET_L2_TOL = 2e-2
WHOLE_TOL_HAL = 1e-3
WHOLE_TOL_LIN = 5e-3


def dev(x, grad=False):
    t = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()
    return t.requires_grad_(grad)


def host(t):
    return t.detach().cpu().numpy()


def f32(x):
    return np.asarray(x, dtype=np.float32)


CONV_GRAD_CASES = [
    # name, N, H, W, C1, C2, Cout, k, stride, act, x2_scale
    ("c16_16_3x3_lrelu", 2, 12, 10, 16, 0, 16, 3, 1, 2, 1.0),
    ("c32_64_3x3_relu", 1, 16, 16, 32, 0, 64, 3, 1, 1, 1.0),
    ("c64_128_3x3", 1, 12, 12, 64, 0, 128, 3, 1, 0, 1.0),
    ("c128_128_3x3_relu", 1, 8, 8, 128, 0, 128, 3, 1, 1, 1.0),
    ("c256_64_1x1", 1, 8, 8, 256, 0, 64, 1, 1, 0, 1.0),
    ("c64_32_5x5_lrelu", 1, 12, 12, 64, 0, 32, 5, 1, 2, 1.0),
    ("c16_16_7x7_lrelu", 1, 14, 14, 16, 0, 16, 7, 1, 2, 1.0),
    ("concat16_16_to16", 1, 12, 12, 16, 16, 16, 3, 1, 2, 1.0),
    ("concat64_64_skipscale_1x1", 2, 8, 8, 64, 64, 64, 1, 1, 0, 1.0 / 255),
    ("c64_128_1x1_s2", 1, 16, 16, 64, 0, 128, 1, 2, 0, 1.0),
    ("c16_3_tanh_direct", 1, 12, 12, 16, 0, 3, 3, 1, 3, 1.0),
    ("c3p3_3_1x1_relu_direct", 1, 12, 12, 3, 3, 3, 1, 1, 1, 1.0 / 255),
    ("c96_64_big_slices", 2, 40, 40, 96, 0, 64, 3, 1, 1, 1.0),
    ("wino_wgrad_ragged_64_64", 2, 13, 19, 64, 0, 64, 3, 1, 2, 1.0),          # Winograd-domain wgrad: odd H, W % 16 != 0
    ("wino_wgrad_concat_64_64_to_128", 1, 18, 34, 64, 64, 128, 3, 1, 1, 0.5),   # two sources, x2 scale
    ("wino_wgrad_256_64_many_units", 3, 32, 48, 256, 0, 64, 3, 1, 0, 1.0),
    ("wino_wgrad_512_512_deep", 2, 16, 16, 512, 0, 512, 3, 1, 1, 1.0),        # the widest layers of the Hallucination-Net
    ("alltaps_wgrad_32_16_ragged", 2, 11, 37, 32, 0, 16, 3, 1, 2, 1.0),        # all-taps narrow wgrad, W % 32 != 0
    ("alltaps_wgrad_16_32_5x5", 1, 20, 33, 16, 0, 32, 5, 1, 2, 1.0),
    ("stem_96_64_7x7_s2", 2, 34, 30, 96, 0, 64, 7, 2, 1, 1.0),              # Linearization-Net stem: the 96-ci wgrad tile
    ("c96_32_1x1_tile96", 1, 24, 24, 96, 0, 32, 1, 1, 0, 1.0),
    ("c96_128_5x5_tile96", 1, 12, 20, 96, 0, 128, 5, 1, 2, 1.0),
]


@pytest.mark.parametrize("case", CONV_GRAD_CASES, ids=[c[0] for c in CONV_GRAD_CASES])
def test_conv2d_backward_parity(shdr, case):
    name, n, h, w, c1, c2, cout, k, stride, act, x2s = case
    rng = np.random.default_rng(len(name) * 131 + h)
    K = shdr._ops
    x = f32(rng.normal(size=(n, h, w, c1)))
    x2 = f32(rng.normal(size=(n, h, w, c2)) * (1.0 / x2s if x2s != 1 else 1.0)) if c2 else None
    wt = f32(rng.normal(size=(k, k, c1 + c2, cout)) / np.sqrt(k * k * (c1 + c2)))
    b = f32(rng.normal(size=cout) * 0.1)
    ho, wo = -(-h // stride), -(-w // stride)
    gy = f32(rng.normal(size=(n, ho, wo, cout)))
    # reference
    tx, tw, tb = R.T(x, True), R.T(wt, True), R.T(b, True)
    tx2 = R.T(x2, True) if c2 else None
    xin = tx if tx2 is None else torch.cat([tx, tx2 * x2s], -1)
    z = R.conv2d(xin, tw, tb, stride)
    y = {0: z, 1: torch.relu(z), 2: R.lrelu(z), 3: torch.tanh(z)}[act]
    (y * R.T(gy)).sum().backward()
    # HIP
    dx, dwt, db = dev(x, True), dev(wt, True), dev(b, True)
    dx2 = dev(x2, True) if c2 else None
    yy = K.conv2d(dx, dwt, db, stride=stride, x2=dx2, x2_scale=x2s, act1=act)
    assert rel_err(host(yy), y.detach().numpy()) <= 1e-5
    (yy * dev(gy)).sum().backward()
    assert rel_err(host(dwt.grad), tw.grad.numpy()) <= TOL, "dW"
    assert rel_err(host(db.grad), tb.grad.numpy()) <= TOL, "db"
    assert rel_err(host(dx.grad), tx.grad.numpy()) <= TOL, "dx"
    if c2:
        assert rel_err(host(dx2.grad), tx2.grad.numpy()) <= TOL, "dx2"


def test_conv2d_padded_filter_gradients(shdr):
    """first layer (Cin 3 -> 4, data input) and 3-channel head (Cout 3 -> 16): the variable keeps its shape"""
    rng = np.random.default_rng(5)
    x = f32(quantised_image(rng, (1, 16, 16, 3)))
    p = {"conv.kernel": f32(rng.normal(size=(7, 7, 3, 16)) * 0.1), "conv.bias": f32(rng.normal(size=16) * 0.1),
         "out.kernel": f32(rng.normal(size=(3, 3, 16, 3)) * 0.1), "out.bias": f32(rng.normal(size=3) * 0.1)}
    tp = {k: R.T(v, True) for k, v in p.items()}
    ty = torch.tanh(R.conv2d(R.lrelu(R.conv2d(R.T(x), tp["conv.kernel"], tp["conv.bias"])), tp["out.kernel"], tp["out.bias"])) + R.T(x)
    (ty ** 2).sum().backward()
    L, K = shdr._layers, shdr._ops
    c1, c2 = L.Conv2D(3, 16, 7), L.Conv2D(16, 3, 3)
    with torch.no_grad():
        c1.kernel.copy_(dev(p["conv.kernel"])); c1.bias.copy_(dev(p["conv.bias"]))
        c2.kernel.copy_(dev(p["out.kernel"])); c2.bias.copy_(dev(p["out.bias"]))
    xd = dev(x)
    t = c1.call_padded(K.pack3([xd], 4), cin_pad=4, act1=K.ACT_LRELU)
    y = K.add(c2.call_padded(t, cout_pad=16, act1=K.ACT_TANH), xd)
    assert rel_err(host(y), ty.detach().numpy()) <= 1e-5
    (y * y).sum().backward()
    assert tuple(c1.kernel.grad.shape) == (7, 7, 3, 16) and tuple(c2.kernel.grad.shape) == (3, 3, 16, 3)
    for g, name in ((c1.kernel.grad, "conv.kernel"), (c1.bias.grad, "conv.bias"), (c2.kernel.grad, "out.kernel"), (c2.bias.grad, "out.bias")):
        assert rel_err(host(g), tp[name].grad.numpy()) <= TOL, name


@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("c", [3, 64])
def test_batchnorm_training_parity(shdr, relu, c):
    rng = np.random.default_rng(c)
    K, A = shdr._ops, shdr._autograd
    x = f32(rng.normal(size=(3, 9, 7, c)) * 2 + 1)
    g, b = f32(rng.uniform(0.5, 1.5, c)), f32(rng.normal(size=c))
    gy = f32(rng.normal(size=x.shape))
    tx, tg, tb = R.T(x, True), R.T(g, True), R.T(b, True)
    ty = R.bn({"n.gamma": tg, "n.beta": tb}, "n", tx, True)
    if relu:
        ty = torch.relu(ty)
    (ty * R.T(gy)).sum().backward()
    dx, dg, db = dev(x, True), dev(g, True), dev(b, True)
    mm, mv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    y = A.batch_norm_train(dx, dg, db, mm, mv, 1e-3, 0.99, relu)
    assert rel_err(host(y), ty.detach().numpy()) <= 1e-5
    (y * dev(gy)).sum().backward()
    assert rel_err(host(dx.grad), tx.grad.numpy()) <= TOL
    assert rel_err(host(dg.grad), tg.grad.numpy()) <= TOL and rel_err(host(db.grad), tb.grad.numpy()) <= TOL
    # Keras moving-average update: momentum 0.99, unbiased variance
    n = x.size // c
    np.testing.assert_allclose(host(mm), 0.01 * x.reshape(-1, c).mean(0), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(host(mv), 0.99 + 0.01 * x.reshape(-1, c).var(0) * n / (n - 1), rtol=1e-4)


def test_pool_resize_gap_backward_parity(shdr):
    rng = np.random.default_rng(7)
    K = shdr._ops
    x = f32(rng.normal(size=(2, 8, 12, 16)))
    for name, hip, ref in (("avgpool2", K.avgpool2, R.avg_pool2), ("maxpool2", K.maxpool2, lambda t: R.max_pool(t, 2, 2)),
                           ("maxpool3s2", K.maxpool3s2, lambda t: R.max_pool(t, 3, 2)), ("resize2x", K.resize2x, R.resize2x),
                           ("gap", K.global_avg_pool, lambda t: t.mean(dim=(1, 2)))):
        tx = R.T(x, True)
        ty = ref(tx)
        gy = f32(rng.normal(size=tuple(ty.shape)))
        (ty * R.T(gy)).sum().backward()
        dx = dev(x, True)
        y = hip(dx)
        (y * dev(gy)).sum().backward()
        assert rel_err(host(dx.grad), tx.grad.numpy()) <= 1e-5, name
    xo = f32(rng.normal(size=(1, 7, 9, 8)))        # odd size for the overlapping 3x3/2 pool
    tx = R.T(xo, True)
    ty = R.max_pool(tx, 3, 2)
    gy = f32(rng.normal(size=tuple(ty.shape)))
    (ty * R.T(gy)).sum().backward()
    dx = dev(xo, True)
    (K.maxpool3s2(dx) * dev(gy)).sum().backward()
    assert rel_err(host(dx.grad), tx.grad.numpy()) <= 1e-6


@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("shape", [(2, 16, 12, 16), (1, 7, 9, 8), (1, 1, 1, 8), (1, 2, 3, 8)])
def test_maxpool3s2_backward_ties_go_to_the_first_maximum(shdr, half, shape):
    """post-ReLU maps are full of equal values: the gradient of a window goes to its FIRST maximum in scan order, as
    tf.nn.max_pool's CPU gradient and the oracle do, and to nobody else.  Values drawn from {0,1,2} tie in most windows."""
    rng = np.random.default_rng(shape[1] * 31 + shape[2])
    K = shdr._ops
    x = f32(rng.integers(0, 3, size=shape))
    tx = R.T(x, True)
    ty = R.max_pool(tx, 3, 2)
    gy = f32(rng.integers(-4, 5, size=tuple(ty.shape)))          # small integers: the sums are exact in fp16 too
    (ty * R.T(gy)).sum().backward()
    dx = dev(x).half().requires_grad_(True) if half else dev(x, True)
    y = K.maxpool3s2(dx)
    (y.float() * dev(gy)).sum().backward()
    np.testing.assert_array_equal(host(dx.grad.float()), tx.grad.numpy())
    assert float(dx.grad.float().sum()) == float(gy.sum())      # every window's gradient lands exactly once


def test_crf_head_backward_parity(shdr, emor_table):
    rng = np.random.default_rng(8)
    K = shdr._ops
    feat, wfc, bfc = f32(rng.normal(size=(3, 512))), f32(rng.normal(size=(512, 11)) * 0.05), f32(rng.normal(size=11) * 0.1)
    x = f32(quantised_image(rng, (3, 8, 8, 3)))
    gy = f32(rng.normal(size=x.shape))
    tf_, tw, tb = R.T(feat, True), R.T(wfc, True), R.T(bfc, True)
    tab = R.T(emor_table)
    inv = R.increase(tab[:, 0][None] + (tf_ @ tw + tb) @ tab[:, 1:].T)
    (R.apply_rf(R.T(x), inv) * R.T(gy)).sum().backward()
    df, dw, db = dev(feat, True), dev(wfc, True), dev(bfc, True)
    inv_h = K.increase(K.invcrf_decode(df, dw, db, dev(emor_table)))
    assert rel_err(host(inv_h), inv.detach().numpy()) <= 1e-5
    (K.apply_rf(dev(x), inv_h) * dev(gy)).sum().backward()
    for g, t, nm in ((df.grad, tf_.grad, "dfeat"), (dw.grad, tw.grad, "dWfc"), (db.grad, tb.grad, "dbfc")):
        assert rel_err(host(g), t.numpy()) <= 2e-4, nm
    # apply_rf gradient w.r.t. x (fine-tuning chain)
    rf = f32(np.sort(rng.random((3, 1024)), axis=1))
    tx, trf = R.T(x * 0.98 + 0.01, True), R.T(rf, True)
    (R.apply_rf(tx, trf) * R.T(gy)).sum().backward()
    dxx, drf = dev(x * 0.98 + 0.01, True), dev(rf, True)
    (K.apply_rf(dxx, drf) * dev(gy)).sum().backward()
    assert rel_err(host(drf.grad), trf.grad.numpy()) <= TOL and rel_err(host(dxx.grad), tx.grad.numpy()) <= TOL


def test_losses_and_glue_backward_parity(shdr):
    rng = np.random.default_rng(9)
    K = shdr._ops
    a, b = f32(rng.random((3, 10, 8, 3))), f32(rng.random((3, 10, 8, 3)))
    g = f32(rng.random(3) + 0.5)
    for mode, ref in ((0, lambda p, q: ((p - q) ** 2).mean(dim=(1, 2, 3))), (1, lambda p, q: (p - q).abs().mean(dim=(1, 2, 3)))):
        ta = R.T(a, True)
        tl = ref(ta, R.T(b))
        (tl * R.T(g)).sum().backward()
        da = dev(a, True)
        l = K.diff_loss(da, dev(b), mode)
        assert rel_err(host(l), tl.detach().numpy()) <= 1e-5
        (l * dev(g)).sum().backward()
        assert rel_err(host(da.grad), ta.grad.numpy()) <= 1e-5
    ta = R.T(a, True)
    tl = R.tv_loss(R.logc(ta))
    (tl * 3.0).backward()
    da = dev(a, True)
    l = K.tv_loss(K.logc(da))
    assert abs(float(l) - float(tl)) <= 1e-6 * max(1.0, abs(float(tl)))
    (l * 3.0).sum().backward()
    assert rel_err(host(da.grad), ta.grad.numpy()) <= 1e-5
    # clip, blend with constant alpha, vgg preprocess, channel reversal
    base = f32(rng.random((2, 6, 5, 3)))
    base[0, 0, 0] = [1.0, 0.95, 0.2]
    hal = f32(rng.random((2, 6, 5, 3)) * 1.4 - 0.2)
    gy = f32(rng.normal(size=hal.shape))
    th = R.T(hal, True)
    ty = R.vgg_preprocess(torch.clamp(R.T(base) + R.alpha_mask(R.T(base)) * th.flip(-1), 0, 1)).flip(-1)
    (ty * R.T(gy)).sum().backward()
    dh = dev(hal, True)
    alpha = K.alpha_mask(dev(base))
    y = K.reverse3(K.vgg_preprocess(K.clip(K.blend_const(dev(base), alpha, dh), 0.0, 1.0)))
    assert rel_err(host(y), ty.detach().numpy()) <= 1e-5
    (y * dev(gy)).sum().backward()
    assert rel_err(host(dh.grad), th.grad.numpy()) <= 1e-5


def _grad_check(model, tparams):
    """(worst relative-L2 error, worst max-norm error) over the trainable variables.

    Deep nets are compared in relative L2: a relu / max-pool mask that flips because the fp32 forward
    differs from the float64 reference by ~1e-6 moves ONE channel's gradient sum by ~1/(samples per
    channel), i.e. percent-level in max-norm but ~1e-4 in L2.  (Measured: every block matches the
    reference to 1e-6 in max-norm when both are fed the same input -- see the op- and block-level
    tests in this file; the flips come from the input, not from the kernels.)  A variable whose true
    gradient vanishes (a bias in front of a batch-norm) is measured against the model-wide scale."""
    named = [(n, t) for n, t, tr in model.named_weights() if tr]
    gmax = max(float(tparams[n].grad.abs().max()) for n, _ in named)
    worst_l2, worst_max = ("", 0.0), ("", 0.0)
    sq_d = sq_r = 0.0
    for name, t in named:
        ref = tparams[name].grad.numpy()
        assert t.grad is not None, name
        d = host(t.grad).astype(np.float64) - ref
        sq_d += float((d ** 2).sum())
        sq_r += float((ref ** 2).sum())
        l2 = float(np.linalg.norm(d) / max(np.linalg.norm(ref), 1e-3 * gmax * np.sqrt(ref.size)))
        mx = float(np.abs(d).max() / max(np.abs(ref).max(), 1e-3 * gmax))
        if l2 > worst_l2[1]:
            worst_l2 = (name, l2)
        if mx > worst_max[1]:
            worst_max = (name, mx)
    _grad_check.whole = float(np.sqrt(sq_d / max(sq_r, 1e-300)))      # relative L2 of the flat gradient of the whole net
    return worst_l2, worst_max


def test_dequantization_net_gradients(shdr):
    p = nets.init_params(nets.deq_spec(), 61)
    m = shdr.dequantization_net.model().load_numpy(p)
    rng = np.random.default_rng(1)
    x, tgt = quantised_image(rng, (2, 32, 32, 3)), quantised_image(rng, (2, 32, 32, 3))
    tp = R.params_to_torch(p)
    (((torch.clamp(R.deq_forward(tp, R.T(x)), 0, 1) - R.T(tgt)) ** 2).mean(dim=(1, 2, 3))).sum().backward()
    K = shdr._ops
    K.diff_loss(K.clip(m(dev(x), training=True), 0.0, 1.0), dev(tgt), 0).sum().backward()
    (n2, e2), (nm, em) = _grad_check(m, tp)
    assert e2 <= TOL and em <= 5 * TOL, (n2, e2, nm, em)


def test_linearization_net_gradients_training_bn(shdr, emor_table):
    p = nets.init_params(nets.lin_spec(), 62)
    m = shdr.linearization_net.model().load_numpy(p)
    rng = np.random.default_rng(2)
    x = quantised_image(rng, (2, 64, 64, 3))
    inv = np.cumsum(rng.random((2, 1024)), axis=1)
    inv /= inv[:, -1:]
    tp = R.params_to_torch(p)
    tinv = R.lin_forward(tp, R.T(x), emor_table, True)
    ((tinv - R.T(inv)) ** 2).mean(dim=1).sum().backward()
    K = shdr._ops
    pred = m(dev(x), training=True)
    assert rel_err(host(pred), tinv.detach().numpy()) <= TOL
    K.diff_loss(pred, dev(inv), 0).sum().backward()
    (n2, e2), (nm, em) = _grad_check(m, tp)
    assert e2 <= NET_L2_TOL, (n2, e2, nm, em)
    assert _grad_check.whole <= WHOLE_TOL_LIN, _grad_check.whole


def test_hallucination_net_gradients_training_bn(shdr):
    p = nets.init_params(nets.hal_spec(), 63)
    m = shdr.hallucination_net.model().load_numpy(p)
    rng = np.random.default_rng(3)
    x, tgt = quantised_image(rng, (2, 64, 64, 3)), rng.random((2, 64, 64, 3))
    tp = R.params_to_torch(p)
    ty = R.hal_forward(tp, R.T(x), True)
    (ty - R.T(tgt)).abs().mean(dim=(1, 2, 3)).sum().backward()
    K = shdr._ops
    y = m(dev(x), training=True)
    assert rel_err(host(y), ty.detach().numpy()) <= TOL
    K.diff_loss(y, dev(tgt), 1).sum().backward()
    (n2, e2), (nm, em) = _grad_check(m, tp)
    assert e2 <= 5e-3, (n2, e2, nm, em)
    assert _grad_check.whole <= WHOLE_TOL_HAL, _grad_check.whole


def test_inference_mode_networks_under_a_tape(shdr, emor_table):
    """`net(x, training=False)` while gradients are recorded (fine-tuning with frozen BatchNorm statistics): the fused inference
    epilogues (folded BN, residual join, second activation) are recorded as conv + affine / join / activation entries, so the
    result stays on the graph and every variable -- the zero-padded 93 -> 96 channel stem filter included -- receives its
    gradient.  (Before: the fused kernel ran on detached tensors and silently cut the graph at the first conv + BN.)"""
    K = shdr._ops
    rng = np.random.default_rng(5)
    # Linearization-Net, moving statistics
    p = nets.init_params(nets.lin_spec(), 64)
    m = shdr.linearization_net.model().load_numpy(p)
    x = quantised_image(rng, (2, 64, 64, 3))
    inv = np.cumsum(rng.random((2, 1024)), axis=1)
    inv /= inv[:, -1:]
    tp = R.params_to_torch(p)
    tinv = R.lin_forward(tp, R.T(x), emor_table, False)
    ((tinv - R.T(inv)) ** 2).mean(dim=1).sum().backward()
    pred = m(dev(x), training=False)
    assert pred.requires_grad and rel_err(host(pred), tinv.detach().numpy()) <= TOL
    with torch.no_grad():
        assert rel_err(host(m(dev(x), training=False)), host(pred)) <= 1e-5        # same numbers as the fused inference kernels
    K.diff_loss(pred, dev(inv), 0).sum().backward()
    named = dict((n, t) for n, t, tr in m.named_weights() if tr)
    assert all(t.grad is not None for t in named.values())
    stem = named["crf_feature_net.conv1.kernel"] if "crf_feature_net.conv1.kernel" in named else next(t for n, t in named.items() if n.endswith("conv1.kernel") and t.shape[2] == 93)
    assert float(stem.grad.abs().max()) > 0.0
    (n2, e2), (nm, em) = _grad_check(m, tp)
    assert e2 <= NET_L2_TOL, (n2, e2, nm, em)
    # Hallucination-Net, moving statistics (3-channel BN + relu head included)
    p = nets.init_params(nets.hal_spec(), 65)
    m = shdr.hallucination_net.model().load_numpy(p)
    x, tgt = quantised_image(rng, (1, 64, 64, 3)), rng.random((1, 64, 64, 3))
    tp = R.params_to_torch(p)
    ty = R.hal_forward(tp, R.T(x), False)
    (ty - R.T(tgt)).abs().mean(dim=(1, 2, 3)).sum().backward()
    y = m(dev(x), training=False)
    assert y.requires_grad and rel_err(host(y), ty.detach().numpy()) <= TOL
    K.diff_loss(y, dev(tgt), 1).sum().backward()
    (n2, e2), (nm, em) = _grad_check(m, tp)
    assert e2 <= NET_L2_TOL, (n2, e2, nm, em)
    # raw-kernel options cannot be taped: loud, not silent
    w = dev(rng.normal(size=(3, 3, 16, 16)), True)
    with pytest.raises(NotImplementedError):
        K.conv2d(dev(rng.normal(size=(1, 8, 8, 16))), w, pad=(1, 1), out_hw=(8, 8))


def test_training_gradients_are_mask_flip_sensitive(shdr):
    """Documents WHY whole-net gradient parity is stated in L2 at the percent level: on the same kernels a
    1e-6 relative perturbation of the input moves the gradients of the randomly initialised, training-mode
    Linearization-Net by far more than it moves the forward (relu / max-pool masks flip), while identical
    inputs reproduce to ~1e-5 (fp32 atomics)."""
    p = nets.init_params(nets.lin_spec(), 62)
    m = shdr.linearization_net.model().load_numpy(p)
    rng = np.random.default_rng(2)
    K = shdr._ops
    x = quantised_image(rng, (2, 64, 64, 3))
    inv = np.cumsum(rng.random((2, 1024)), axis=1)
    inv /= inv[:, -1:]

    def grads(xin):
        for _, t, _ in m.named_weights():
            t.grad = None
        K.diff_loss(m(dev(xin), training=True), dev(inv), 0).sum().backward()
        return np.concatenate([host(t.grad).ravel() for _, t, tr in m.named_weights() if tr]).astype(np.float64)

    g0, g1 = grads(x), grads(x)
    gp = grads(x * (1 + 1e-6 * rng.standard_normal(x.shape)))
    same = np.linalg.norm(g1 - g0) / np.linalg.norm(g0)
    pert = np.linalg.norm(gp - g0) / np.linalg.norm(g0)
    assert same <= 1e-4, same                 # run-to-run: atomics only
    assert pert <= NET_L2_TOL, pert           # ...and the sensitivity itself stays inside the stated bar


def test_residual_blocks_exact_on_shared_inputs(shdr):
    """Type-1 (stride 2) -> type-2 residual chain with training-mode BN, reference fed the same input:
    max-norm parity of every gradient (this is what the kernels are responsible for)."""
    LN = shdr.linearization_net
    spec = nets._res1("a", 256, 512, [128, 128, 512]) + nets._res2("b", 512, [128, 128, 512])
    p = nets.init_params(spec, 7)
    blk_a = LN.resBlock_type1(256, 512, [128, 128, 512], (2, 2)).load_numpy({k[2:]: v for k, v in p.items() if k.startswith("a.")})
    blk_b = LN.resBlock_type2(512, [128, 128, 512]).load_numpy({k[2:]: v for k, v in p.items() if k.startswith("b.")})
    rng = np.random.default_rng(0)
    x = f32(np.maximum(rng.normal(size=(2, 8, 8, 256)), 0))
    tp = R.params_to_torch(p)
    tx = R.T(x, True)

    def bnr(n, t, relu):
        y = R.bn(tp, n, t, True)
        return torch.relu(y) if relu else y

    n1 = bnr("a.norm1", R._c(tp, "a.conv1", tx, 2), False)
    a3 = bnr("a.norm3", R._c(tp, "a.conv3", bnr("a.norm2", R._c(tp, "a.conv2", tx, 2), True)), True)
    mid = torch.relu(n1 + bnr("a.norm4", R._c(tp, "a.conv4", a3), False))
    b2 = bnr("b.norm2", R._c(tp, "b.conv2", bnr("b.norm1", R._c(tp, "b.conv1", mid), True)), True)
    ty = torch.relu(mid + bnr("b.norm3", R._c(tp, "b.conv3", b2), False))
    gy = f32(rng.normal(size=tuple(ty.shape)))
    (ty * R.T(gy)).sum().backward()
    dx = dev(x, True)
    y = blk_b(blk_a(dx, training=True), training=True)
    assert rel_err(host(y), ty.detach().numpy()) <= 1e-5
    (y * dev(gy)).sum().backward()
    assert rel_err(host(dx.grad), tx.grad.numpy()) <= 1e-5
    for blk, pre in ((blk_a, "a."), (blk_b, "b.")):
        for nm, t, tr in blk.named_weights():
            if tr:
                assert rel_err(host(t.grad), tp[pre + nm].grad.numpy()) <= 1e-5, pre + nm


@pytest.mark.parametrize("c", [16, 64, 512, 1024, 12, 3])
def test_fused_activation_backward_and_bias_gradient(shdr, c):
    """act_bwd_bias = act_bwd followed by bias_grad (the fused kernel for C/4 a power of two, the pair otherwise)"""
    K = shdr._ops
    g = torch.Generator().manual_seed(c)
    dy = torch.randn((3, 9, 11, c), generator=g).cuda()
    y = torch.randn((3, 9, 11, c), generator=g).cuda()
    for act in (K.ACT_NONE, K.ACT_RELU, K.ACT_LRELU, K.ACT_TANH):
        dz, db = K.act_bwd_bias(dy, y, act)
        want = K.act_bwd(dy, y, act) if act != K.ACT_NONE else dy
        assert torch.equal(dz, want)
        ref = want.double().sum(dim=(0, 1, 2))
        assert float((db.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max() + 1)
