"""CPU tests pinning tests/torch_ref.py (the gradient reference) to the NumPy oracle."""
import numpy as np
import torch

import torch_ref as R
from conftest import quantised_image
from oracle import nets, ops


def _np(t):
    return t.detach().numpy()


def test_torch_ref_forward_equals_numpy_oracle(emor_table):
    rng = np.random.default_rng(0)
    x = quantised_image(rng, (2, 32, 32, 3))
    P = {k: nets.init_params(getattr(nets, k + "_spec")(), 40 + i) for i, k in enumerate(("deq", "lin", "hal"))}
    np.testing.assert_allclose(_np(R.deq_forward(R.params_to_torch(P["deq"]), R.T(x))), nets.deq_forward(P["deq"], x), atol=1e-10)
    for training in (False, True):
        np.testing.assert_allclose(_np(R.lin_forward(R.params_to_torch(P["lin"]), R.T(x), emor_table, training)),
                                   nets.lin_forward(P["lin"], x, emor_table, training=training), atol=1e-9)
        np.testing.assert_allclose(_np(R.hal_forward(R.params_to_torch(P["hal"]), R.T(x), training)),
                                   nets.hal_forward(P["hal"], x, training=training), rtol=1e-9, atol=1e-8)
    V = nets.init_params(nets.vgg_spec(), 43)
    for a, b in zip(R.vgg_forward(R.params_to_torch(V, False), R.T(x)), nets.vgg_forward(V, x)):
        np.testing.assert_allclose(_np(a), b, rtol=1e-9, atol=1e-8)


def test_torch_ref_joint_losses_equal_numpy_oracle(emor_table):
    rng = np.random.default_rng(1)
    b, s = 2, 32
    clipped = quantised_image(rng, (b, s, s, 3))
    clipped[0, :6, :6] = 1.0
    batch = (quantised_image(rng, (b, s, s, 3)), quantised_image(rng, (b, s, s, 3)), clipped,
             clipped * (1 + 3 * rng.random((b, s, s, 3)) * (clipped >= 1.0)), np.array([1.0, 0.0]).reshape(b, 1, 1, 1))
    inv = np.cumsum(rng.random((b, 1024)), axis=1)
    inv = inv / inv[:, -1:]
    P = {k: nets.init_params(getattr(nets, k + "_spec")(), 50 + i) for i, k in enumerate(("deq", "lin", "hal"))}
    V = nets.init_params(nets.vgg_spec(), 53)
    ref = nets.joint_losses(P, V, batch, inv, emor_table)
    out = R.joint_losses({k: R.params_to_torch(v) for k, v in P.items()}, R.params_to_torch(V, False),
                         tuple(R.T(t) for t in batch), R.T(inv), emor_table)
    for k in ("loss_deq", "loss_lin", "loss_hal", "total", "crf_loss"):
        assert tuple(out[k].shape) == ref[k].shape, k
        np.testing.assert_allclose(_np(out[k]), ref[k], rtol=1e-9, atol=1e-10, err_msg=k)
    # the reference's shapes (joint_training.py:152-160,182-183): per-sample terms [b,1,1,1], crf_loss [b,1], and therefore
    # loss_lin / total_loss broadcast to [b,1,b,1]; what tape.gradient differentiates is the sum over all b*b elements
    assert ref["loss_deq"].shape == (b, 1, 1, 1) and ref["crf_loss"].shape == (b, 1)
    assert ref["loss_lin"].shape == (b, 1, b, 1) and ref["total"].shape == (b, 1, b, 1)
    mask = batch[4].reshape(-1)
    l2 = ((ref["B_pred"] - batch[2]) ** 2).mean(axis=(1, 2, 3))
    for i in range(b):
        for j in range(b):
            want = mask[i] * (10.0 * l2[i] + ref["crf_loss"][j, 0])
            assert abs(ref["loss_lin"][i, 0, j, 0] - want) <= 1e-12 * max(1.0, abs(want))
    closed = (b * (ref["loss_deq"].sum() + (10.0 * l2 * mask).sum() + ref["loss_hal"].sum()) + mask.sum() * ref["crf_loss"].sum())
    assert abs(ref["total"].sum() - closed) <= 1e-12 * abs(closed)
    # a masked sample (mask[1] = 0) still receives crf gradient: d total.sum() / d crf_1 = sum(mask) = 1, not b * mask_1 = 0
    out["total"].sum().backward()


def test_increase_and_apply_rf_match_oracle():
    rng = np.random.default_rng(2)
    rf = np.cumsum(rng.normal(0.001, 0.01, (2, 64)), axis=1)
    np.testing.assert_allclose(_np(R.increase(R.T(rf))), ops.increase(rf), atol=1e-12)
    x = rng.random((2, 5, 5, 3))
    lut = np.sort(rng.random((2, 64)), axis=1)
    np.testing.assert_allclose(_np(R.apply_rf(R.T(x), R.T(lut))), ops.apply_rf(x, lut), atol=1e-12)
