"""GPU tests of the joint training step (joint_training.py:137-194) against the float64 reference."""
import importlib
import math

import numpy as np
import pytest
import torch

import torch_ref as R
from conftest import quantised_image, rel_err
from oracle import nets

pytestmark = pytest.mark.gpu


def dev(x):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def make_batch(rng, b, s):
    """SURVEY.md section 8d config 4 input law"""
    clipped = quantised_image(rng, (b, s, s, 3))
    clipped[0, :10, :10] = 1.0                                  # saturated patch: alpha mask active
    hdr_t = clipped * np.where(clipped >= 1.0, 1 + 3 * rng.random((b, s, s, 3)), 1.0)
    inv = np.cumsum(rng.random((b, 1024)), axis=1)
    inv = (inv - inv[:, :1]) / (inv[:, -1:] - inv[:, :1])
    mask = np.ones((b, 1, 1, 1))
    mask[-1] = 0.0                                              # one masked sample
    return (quantised_image(rng, (b, s, s, 3)), quantised_image(rng, (b, s, s, 3)), clipped, hdr_t, mask), inv


@pytest.fixture(scope="module")
def setup(shdr, emor_table):
    rng = np.random.default_rng(11)
    P = {k: nets.init_params(getattr(nets, k + "_spec")(), 70 + i) for i, k in enumerate(("deq", "lin", "hal"))}
    V = nets.init_params(nets.vgg_spec(), 73)
    batch, inv = make_batch(rng, 3, 64)
    ms = {"deq": shdr.dequantization_net.model().load_numpy(P["deq"]),
          "lin": shdr.linearization_net.model().load_numpy(P["lin"]),
          "hal": shdr.hallucination_net.model().load_numpy(P["hal"])}
    dd = {n: [V[n + ".kernel"], V[n + ".bias"]] for n in ("conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3")}
    vgg = shdr.vgg16.Vgg16(data_dict=dd)
    step = shdr.pipeline.JointTrainStep(ms["deq"], ms["lin"], ms["hal"], vgg, lr=1e-3)
    tP = {k: R.params_to_torch(v) for k, v in P.items()}
    ref = R.joint_losses(tP, R.params_to_torch(V, False), tuple(R.T(t) for t in batch), R.T(inv), emor_table)
    ref["total"].sum().backward()
    return dict(step=step, models=ms, batch=tuple(dev(t) for t in batch), inv=dev(inv), ref=ref, tP=tP, P=P)


def test_joint_losses_match_reference(setup, split_forced):
    out = setup["step"](setup["batch"], setup["inv"], apply=False)
    b = setup["batch"][0].shape[0]
    assert tuple(out["loss_lin"].shape) == (b, 1, b, 1) == tuple(out["total"].shape)     # the reference's broadcast (:158-160,183)
    for k in ("loss_deq", "loss_lin", "loss_hal", "total", "crf_loss"):
        want = setup["ref"][k].detach().numpy()
        assert rel_err(host(out[k]).reshape(want.shape), want) <= 1e-4, k
    assert float(out["total"][-1].detach().abs().max()) == 0.0           # row of the masked sample: nothing but TV * 0
    # what is differentiated is the sum over the [b,1,b,1] tensor
    assert abs(float(out["objective"]) - float(out["total"].sum())) <= 1e-5 * abs(float(out["total"].sum()))
    for k in ("C_pred", "B_pred", "A_pred"):
        assert rel_err(host(out[k]), setup["ref"][k].detach().numpy()) <= 1e-4, k


def test_joint_gradients_and_flat_buffer(setup, split_forced):
    step = setup["step"]
    step(setup["batch"], setup["inv"], apply=False)
    assert step.params.num_params == 27741644                   # deq+lin+hal trainable parameters
    worst = 0.0
    for net in ("deq", "lin", "hal"):
        got = np.concatenate([host(t.grad).ravel() for t in setup["models"][net].trainable_variables]).astype(np.float64)
        ref = np.concatenate([setup["tP"][net][n].grad.numpy().ravel() for n, _, tr in setup["models"][net].named_weights() if tr])
        worst = max(worst, np.linalg.norm(got - ref) / np.linalg.norm(ref))
    assert worst <= 1e-2, worst                                  # per-net flat gradient, relative L2 (6e-3 measured: the lin head, test_gpu_grad.py)
    # the flat gradient buffer holds exactly these gradients (alignment gaps stay zero)
    assert abs(float(step.params.grad.double().abs().sum()) -
               sum(float(t.grad.double().abs().sum()) for m in setup["models"].values() for t in m.trainable_variables)) <= 1e-6 * float(step.params.grad.double().abs().sum())
    assert all(t.data_ptr() % 16 == 0 and t.grad.data_ptr() % 16 == 0 for m in setup["models"].values() for t in m.trainable_variables)
    # variables alias the flat buffers (one Adam kernel / one all-reduce per step)
    v0 = setup["models"]["deq"].trainable_variables[0]
    assert v0.data_ptr() == step.params.flat.data_ptr() and v0.grad.data_ptr() == step.params.grad.data_ptr()


def test_keras_adam_update(setup):
    step = setup["step"]
    step(setup["batch"], setup["inv"], apply=False)
    p0, g = host(step.params.flat).astype(np.float64), host(step.params.grad).astype(np.float64)
    m0, v0, t0 = host(step.params.m).astype(np.float64), host(step.params.v).astype(np.float64), step.optimizer.t
    step.optimizer.step()
    t = t0 + 1
    m = 0.9 * m0 + 0.1 * g
    v = 0.999 * v0 + 0.001 * g * g
    lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
    expect = p0 - lr_t * m / (np.sqrt(v) + 1e-7)                 # Keras epsilon placement
    np.testing.assert_allclose(host(step.params.flat), expect, rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(host(step.params.m), m, rtol=1e-5, atol=1e-12)


def test_training_reduces_the_loss(setup):
    step = setup["step"]
    first = float(step(setup["batch"], setup["inv"])["total"].sum())
    for _ in range(4):
        last = float(step(setup["batch"], setup["inv"])["total"].sum())
    assert np.isfinite(last) and last < first, (first, last)


def test_joint_step_at_its_own_size_properties(shdr):
    """BASELINE configs[3] at its own size -- batch 32 of 256 x 256, where the split-operand kernels take the wide layers in the
    forward AND in the input gradient (range-scaled dz) -- as properties: finite losses and gradients, a decreasing loss, and the
    split-operand input gradient against the exact-fp32 one on the SAME tape (forward under EXACT_FP32, so both backward passes see
    identical relu / max-pool masks): flat gradient within 6e-5 in relative L2 (3.0e-5 measured, Winograd-fp32 vs
    direct-fp32 2.4e-6, atomics noise 1e-6)."""
    K, P = shdr._ops, shdr.pipeline
    torch.manual_seed(1234)
    g = torch.Generator().manual_seed(9)
    b, sz = 32, 256

    def q(*shape):
        return (torch.round(torch.rand(shape, generator=g) * 255.0) / 255.0).cuda()
    clipped = q(b, sz, sz, 3)
    clipped[0, :40, :40] = 1.0
    hdr_t = clipped * torch.where(clipped >= 1.0, 1 + 3 * torch.rand((b, sz, sz, 3), generator=g).cuda(), torch.ones(()).cuda())
    inv = torch.cumsum(torch.rand((b, 1024), generator=g), dim=1).cuda()
    inv = (inv - inv[:, :1]) / (inv[:, -1:] - inv[:, :1])
    mask = torch.ones((b, 1, 1, 1)).cuda()
    mask[-1] = 0.0
    ds = (q(b, sz, sz, 3), q(b, sz, sz, 3), clipped, hdr_t, mask)
    nets3 = [shdr.dequantization_net.model(), shdr.linearization_net.model(), shdr.hallucination_net.model()]
    vgg = shdr.vgg16.Vgg16(data_dict={n: [torch.randn(s, generator=g).mul_(0.05).numpy(), np.zeros(s[-1], np.float32)] for n, s in (
        ("conv1_1", (3, 3, 3, 64)), ("conv1_2", (3, 3, 64, 64)), ("conv2_1", (3, 3, 64, 128)), ("conv2_2", (3, 3, 128, 128)),
        ("conv3_1", (3, 3, 128, 256)), ("conv3_2", (3, 3, 256, 256)), ("conv3_3", (3, 3, 256, 256)))})
    step = P.JointTrainStep(*nets3, vgg, lr=1e-5)
    assert K.conv2d_plan((b, sz, sz, 64), (3, 3, 64, 64)) == "x3"              # the wide layers do run on the split-operand kernel here

    def grads(exact_backward):
        step.params.zero_grad()
        K.EXACT_FP32 = True                                                    # the same exact-fp32 forward for both: identical masks
        try:
            out = step.losses(ds, inv)
            K.EXACT_FP32 = exact_backward
            out["objective"].backward()
            step._join_streams()
        finally:
            K.EXACT_FP32 = False
        torch.cuda.synchronize()
        return step.params.grad.double().clone(), out
    g_exact, out = grads(True)
    g_split, _ = grads(False)
    assert bool(torch.isfinite(g_exact).all()) and bool(torch.isfinite(g_split).all()) and float(g_exact.norm()) > 0.0
    rel = float((g_split - g_exact).norm() / g_exact.norm())
    print("joint step b=32 x 256^2: split-operand vs exact-fp32 input gradients, flat gradient relative L2 = %.2e" % rel)
    assert rel <= 6e-5, rel
    first = float(step(ds, inv)["total"].sum())                                # default plan: split-operand forward and backward
    for _ in range(3):
        o = step(ds, inv)
    last = float(o["total"].sum())
    assert all(bool(torch.isfinite(o[k]).all()) for k in ("total", "C_pred", "B_pred", "A_pred"))
    assert np.isfinite(last) and last < first, (first, last)


def test_per_network_train_steps_match_the_joint_pieces(shdr, emor_table):
    """train.py:164-244: `deq_train_step` and `hal_train_step` use exactly the joint step's loss terms, so their gradients
    must equal the joint step's gradients of those networks; `lin_train_step` weighs its terms differently
    (L2 + 0.1 * crf instead of 10 * L2 + crf) and is checked against the float64 reference."""
    rng = np.random.default_rng(21)
    P = {k: nets.init_params(getattr(nets, k + "_spec")(), 80 + i) for i, k in enumerate(("deq", "lin", "hal"))}
    V = nets.init_params(nets.vgg_spec(), 83)
    batch, inv = make_batch(rng, 2, 64)
    ldr, jpeg, clipped, hdr_t, mask = (dev(t) for t in batch)
    mods = dict(deq="dequantization_net", lin="linearization_net", hal="hallucination_net")

    def fresh():
        return {k: getattr(shdr, mods[k]).model().load_numpy(P[k]) for k in mods}
    dd = {n: [V[n + ".kernel"], V[n + ".bias"]] for n in ("conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3")}
    vgg = shdr.vgg16.Vgg16(data_dict=dd)
    a = fresh()
    joint = shdr.pipeline.JointTrainStep(a["deq"], a["lin"], a["hal"], vgg, multi_stream=False)
    jout = joint((ldr, jpeg, clipped, hdr_t, mask), dev(inv), apply=False)
    # the joint step differentiates the sum over its broadcast [b,1,b,1] loss: every per-sample term carries the factor b
    jgrad = {k: torch.cat([t.grad.reshape(-1) for t in a[k].trainable_variables]).clone() / float(ldr.shape[0]) for k in mods}

    b = fresh()
    s_deq = shdr.pipeline.TrainStep("deq", b["deq"])
    (pred,) = s_deq((ldr, jpeg, mask), apply=False)
    assert torch.equal(pred, jout["C_pred"])
    assert rel_err(host(s_deq.last_loss), host(jout["loss_deq"])) <= 1e-6        # the loss reduction sums with atomics
    g = torch.cat([t.grad.reshape(-1) for t in b["deq"].trainable_variables])
    assert float((g - jgrad["deq"]).abs().max()) <= 1e-6 * float(jgrad["deq"].abs().max())

    s_hal = shdr.pipeline.TrainStep("hal", b["hal"], vgg)
    pred_rgb, y_final, alpha = s_hal((hdr_t, clipped, mask), apply=False)
    assert torch.equal(y_final, jout["A_pred"]) and torch.equal(alpha, jout["alpha"])
    assert rel_err(host(s_hal.last_loss), host(jout["loss_hal"])) <= 1e-6
    g = torch.cat([t.grad.reshape(-1) for t in b["hal"].trainable_variables])
    assert float((g - jgrad["hal"]).norm() / jgrad["hal"].norm()) <= 1e-4       # same kernels, atomics reorder the sums
    assert tuple(pred_rgb.shape) == tuple(y_final.shape)

    s_lin = shdr.pipeline.TrainStep("lin", b["lin"])
    b_pred, crf_mean = s_lin((ldr, clipped, mask, dev(inv)), apply=False)
    assert torch.equal(b_pred, jout["B_pred"])
    tP = R.params_to_torch(P["lin"])
    want, crf, t_b2 = R.lin_train_loss(tP, R.T(batch[0]), R.T(batch[2]), R.T(batch[4]), R.T(inv), emor_table)
    assert tuple(want.shape) == (2, 1, 2, 1) == tuple(s_lin.last_loss.shape)   # train.py:189-191 broadcasts like the joint step
    assert rel_err(host(s_lin.last_loss), want.detach().numpy()) <= 1e-4
    assert abs(float(crf_mean) - float(crf.mean())) <= 1e-4 * float(crf.mean())
    want.sum().backward()
    got = torch.cat([t.grad.reshape(-1) for t in b["lin"].trainable_variables]).double().cpu()
    ref = torch.cat([tP[n].grad.reshape(-1) for n, _, tr in b["lin"].named_weights() if tr])
    assert float((got - ref).norm() / ref.norm()) <= 5e-3                      # whole-net bar (test_gpu_grad.WHOLE_TOL_LIN)
    # and the steps train: Adam 1e-4 on the step's own variables only
    before = b["deq"].trainable_variables[0].detach().clone()
    s_deq((ldr, jpeg, mask))
    assert not torch.equal(b["deq"].trainable_variables[0].detach(), before)
    assert s_deq.optimizer.lr == 1e-4 and s_deq.params.num_params == 1999779


def test_inference_caches_follow_in_place_updates(shdr):
    """Adam and the BatchNorm moving statistics are updated by kernels through raw pointers; the per-version caches of the
    layers (padded / x2-scaled filters, folded BN constants) must notice: an inference call after a training step uses the
    updated values."""
    K = shdr._ops
    L = importlib.import_module("singlehdr-tf2_amd._layers")
    torch.manual_seed(3)
    conv = L.Conv2D(3, 3, (1, 1))
    bn = L.BatchNormalization(16)
    x4 = torch.rand(1, 8, 8, 4, device="cuda")
    params = shdr.pipeline.FlatParams([conv])
    with torch.no_grad():
        y0 = conv.call_padded(x4, cin_pad=4, cout_pad=16).clone()             # fills the padded-filter cache
    s0 = bn.folded()[0].clone()
    params.grad.fill_(1.0)
    shdr.pipeline.KerasAdam(params, 0.1).step()                                # kernel writes params.flat in place
    with torch.no_grad():
        y1 = conv.call_padded(x4, cin_pad=4, cout_pad=16)
    assert float((y1 - y0).abs().max()) > 1e-3                                 # not the stale cached filter
    ref = K.conv2d(x4, torch.nn.functional.pad(conv.kernel.detach(), (0, 13, 0, 1)).contiguous(), conv.bias.detach(), cout_valid=3)
    assert torch.equal(y1, ref)
    # the packed Winograd filter kept on a persistent kernel follows the update as well
    wide = L.Conv2D(32, 64, (3, 3))
    p2 = shdr.pipeline.FlatParams([wide])
    x32 = torch.rand(1, 16, 16, 32, device="cuda")
    with torch.no_grad():
        z0 = wide(x32).clone()
        assert getattr(wide.kernel, "_shdr_packed", None) is not None and torch.equal(wide(x32), z0)    # second call: cache hit
    p2.grad.fill_(-1.0)
    shdr.pipeline.KerasAdam(p2, 0.05).step()
    with torch.no_grad():
        z1 = wide(x32)
        fresh = K.conv2d_winograd_fused(x32, K.winograd_filter_packed(wide.kernel.detach()), wide.bias.detach())
    assert torch.equal(z1, fresh) and float((z1 - z0).abs().max()) > 1e-2
    K.bn_stats(torch.randn(2, 8, 8, 16, device="cuda") * 3 + 1, bn.moving_mean, bn.moving_variance)   # moving stats updated in place
    assert float((bn.folded()[0] - s0).abs().max()) > 1e-4
