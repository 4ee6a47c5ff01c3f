"""GPU tests of the chained fine-tuning step (finetune_real_dataset.py:144-183) and its extra backward ops."""
import numpy as np
import pytest
import torch

import torch_ref as R
from conftest import quantised_image, rel_err
from oracle import nets

pytestmark = pytest.mark.gpu
TOL = 1e-4


def dev(x, grad=False):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda().requires_grad_(grad)


def host(t):
    return t.detach().cpu().numpy()


def f32(x):
    return np.asarray(x, dtype=np.float32)


def test_frontend_backward_parity(shdr):
    rng = np.random.default_rng(1)
    x = f32(rng.random((2, 12, 10, 3)) * 0.98 + 0.01)     # off the bin edges / centres (kinks of |.|)
    g = f32(rng.normal(size=(2, 12, 10, 96)))
    g[..., 93:] = 0
    tx = R.T(x, True)
    (R.lin_frontend(tx) * R.T(g[..., :93])).sum().backward()
    dx = dev(x, True)
    (shdr._ops.lin_frontend(dx, 96) * dev(g)).sum().backward()
    assert rel_err(host(dx.grad), tx.grad.numpy()) <= TOL


def test_strided_7x7_dgrad_parity(shdr):
    """polyphase input gradient of stride-2 convolutions: even and odd sizes (the SAME padding changes with the parity of
    the size), 7x7 (Linearization-Net conv1), 5x5 and 3x3"""
    rng = np.random.default_rng(2)
    for h, w, k in ((16, 16, 7), (14, 18, 7), (15, 17, 7), (13, 16, 5), (12, 11, 3), (2, 2, 7)):
        x = f32(rng.normal(size=(2, h, w, 32)))
        wt = f32(rng.normal(size=(k, k, 32, 64)) / 40)
        gy = f32(rng.normal(size=(2, (h + 1) // 2, (w + 1) // 2, 64)))
        tx, tw = R.T(x, True), R.T(wt, True)
        (R.conv2d(tx, tw, None, 2) * R.T(gy)).sum().backward()
        dx, dw = dev(x, True), dev(wt, True)
        (shdr._ops.conv2d(dx, dw, stride=2) * dev(gy)).sum().backward()
        assert rel_err(host(dx.grad), tx.grad.numpy()) <= TOL and rel_err(host(dw.grad), tw.grad.numpy()) <= TOL


def test_blend_pack_meannorm_backward_parity(shdr):
    rng = np.random.default_rng(3)
    K = shdr._ops
    b = f32(rng.random((2, 9, 7, 3)))
    b[0, 0, 0] = [0.95, 0.2, 0.3]            # 0 < alpha < 1
    b[0, 0, 1] = [1.03, 0.99, 0.5]           # alpha clamped to 1: no gradient through alpha (exact ties follow
                                             # TF's tf.minimum(1.0, x) rule -- gradient to the constant -- not torch.clamp's)
    hal = f32(rng.random((2, 9, 7, 3)))
    c = f32(rng.random((2, 9, 7, 3)))
    gy = f32(rng.normal(size=(2, 9, 7, 12)))
    tb, th, tc = R.T(b, True), R.T(hal, True), R.T(c, True)
    ta = tb + R.alpha_mask(tb) * th.flip(-1)
    packed = torch.cat([ta, tb, tc, torch.zeros(2, 9, 7, 3, dtype=R.DT)], -1)
    r = packed[..., 0:3] * 2.0 + packed[..., 3:6] + 0.1
    out = r / (1e-6 + r.mean(dim=(1, 2, 3), keepdim=True)) * 0.5
    ((out ** 2).sum() + (packed * R.T(gy)).sum()).backward()
    db, dh, dc = dev(b, True), dev(hal, True), dev(c, True)
    a = K.alpha_blend(db, dh, 0.12)
    hp = K.pack3([a, db, dc], 12)
    s0, s1 = K.unpack3(hp, 2)
    hr = K.add(K.add(K.add(s0, s0), s1), dev(np.full((2, 9, 7, 3), 0.1)))
    ho = K.mean_norm(hr, 1e-6, 0.5)
    assert rel_err(host(ho), out.detach().numpy()) <= 1e-5
    ((ho * ho).sum() + (hp * dev(gy)).sum()).backward()
    for g, t, nm in ((db.grad, tb.grad, "dB"), (dh.grad, th.grad, "dhal"), (dc.grad, tc.grad, "dC")):
        assert rel_err(host(g), t.numpy()) <= TOL, nm


def test_refinement_net_gradients(shdr):
    p = nets.init_params(nets.ref_spec(), 91)
    m = shdr.refinement_net.model().load_numpy(p)
    rng = np.random.default_rng(4)
    x = rng.random((2, 32, 32, 9))
    tgt = rng.random((2, 32, 32, 3))
    tp = R.params_to_torch(p)
    tx = R.T(x, True)
    (R.ref_forward(tp, tx) - R.T(tgt)).abs().mean(dim=(1, 2, 3)).sum().backward()
    x12 = dev(np.concatenate([x, np.zeros((2, 32, 32, 3))], -1), True)
    K = shdr._ops
    y = m(x12, training=True)
    K.diff_loss(y, dev(tgt), 1).sum().backward()
    assert rel_err(host(x12.grad)[..., :9], tx.grad.numpy()) <= 5 * TOL
    for name, t, tr in m.named_weights():
        assert rel_err(host(t.grad), tp[name].grad.numpy()) <= 5 * TOL, name


@pytest.fixture(scope="module")
def ft(shdr, emor_table):
    rng = np.random.default_rng(12)
    P = {k: nets.init_params(getattr(nets, k + "_spec")(), 95 + i) for i, k in enumerate(("deq", "lin", "hal", "ref"))}
    ldr = quantised_image(rng, (2, 64, 64, 3))
    ldr[0, :12, :12] = 1.0
    hdr = rng.random((2, 64, 64, 3)) * 1.5
    hdr = hdr / (1e-6 + hdr.mean(axis=(1, 2, 3), keepdims=True)) * 0.5
    mods = dict(deq="dequantization_net", lin="linearization_net", hal="hallucination_net", ref="refinement_net")
    ms = {k: getattr(shdr, mods[k]).model().load_numpy(P[k]) for k in mods}
    step = shdr.pipeline.FinetuneStep(ms["deq"], ms["lin"], ms["hal"], ms["ref"], lr=1e-4)
    tP = {k: R.params_to_torch(v) for k, v in P.items()}
    ref = R.finetune_forward(tP, R.T(ldr), R.T(hdr), emor_table)
    ref["loss"].sum().backward()
    return dict(step=step, ms=ms, ldr=dev(ldr), hdr=dev(hdr), ref=ref, tP=tP, P=P, np=(ldr, hdr))


def test_finetune_forward_matches_oracle(ft, emor_table):
    out = ft["step"](ft["ldr"], ft["hdr"], apply=False)
    oracle = nets.finetune_forward(ft["P"], ft["np"][0], ft["np"][1], emor_table)
    for k in ("C_pred", "B_pred", "A_pred", "refinement_output"):
        assert rel_err(host(out[k]), oracle[k]) <= TOL, k
    assert rel_err(host(out["loss_sum"]), oracle["loss"].sum(axis=(1, 2, 3))) <= TOL
    assert ft["step"].params.num_params == 29008591


def test_finetune_gradients(ft):
    ft["step"](ft["ldr"], ft["hdr"], apply=False)
    worst = 0.0
    for net in ("deq", "lin", "hal", "ref"):
        got = np.concatenate([host(t.grad).ravel() for t in ft["ms"][net].trainable_variables]).astype(np.float64)
        ref = np.concatenate([ft["tP"][net][n].grad.numpy().ravel() for n, _, tr in ft["ms"][net].named_weights() if tr])
        worst = max(worst, np.linalg.norm(got - ref) / np.linalg.norm(ref))
    assert worst <= 2e-2, worst                                # per-net flat gradient of the CHAINED step (four nets deep), relative L2 (1.3e-2 measured)


def test_finetune_reduces_the_loss(ft):
    first = float(ft["step"](ft["ldr"], ft["hdr"])["loss_sum"].detach().sum())
    for _ in range(4):
        last = float(ft["step"](ft["ldr"], ft["hdr"])["loss_sum"].detach().sum())
    assert np.isfinite(last) and last < first, (first, last)
