"""BASELINE configs[4] ("finetune_real_dataset.py with Refinement-Net, 1024x1024 tiles, fp16 MFMA conv path") against the
float64 ORACLE -- not against the HIP fp32 kernels (tests/test_gpu_fp16.py does that and only shows that two HIP paths agree).

Stated fp16 bounds (unit round-off u = 2^-11 = 4.9e-4; operands -- and, in the native-fp16 layout, stored activations -- are
rounded to nearest-even, accumulation is fp32):
  * one conv layer, forward / input gradient / weight gradient:  max|err| <= 3e-3 * max|reference tensor|;
  * the chained fine-tuning step (finetune_real_dataset.py:144-183, ~80 conv layers deep): intermediates and the loss within
    2e-2 / 1e-2 relative of the oracle, flat 29 M-parameter gradient within cosine >= 0.95 of the float64 autograd reference
    (whole-net gradients of these training-mode nets move by 1-3.5 % under a 1e-6 input perturbation, test_gpu_grad.py);
  * 4 x 1024^2 (the configuration's own size; the oracle would take hours): size-independent properties -- finite, no skipped
    step, deterministic forward, the loss decreases over Adam steps.
"""
import numpy as np
import pytest
import torch

import torch_ref as R
from conftest import quantised_image, rel_err
from oracle import nets, ops

pytestmark = pytest.mark.gpu
LAYER_TOL = 3e-3


def dev(x, grad=False):
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda().requires_grad_(grad)


def host(t):
    return t.detach().float().cpu().numpy()


CASES = [
    # name, N, H, W, C1, C2, Cout, k, stride, act, x2_scale
    ("hal_3x3_64_128_relu", 2, 24, 20, 64, 0, 128, 3, 1, 1, 1.0),
    ("hal_3x3_128_64", 1, 32, 32, 128, 0, 64, 3, 1, 1, 1.0),
    ("unet_concat_32_32_64_lrelu", 2, 17, 23, 32, 32, 64, 3, 1, 2, 1.0),
    ("skip_1x1_64_64_scaled", 2, 16, 16, 64, 64, 64, 1, 1, 0, 1.0 / 255),
    ("deq_7x7_16_16_lrelu", 1, 32, 32, 16, 0, 16, 7, 1, 2, 1.0),
    ("deq_5x5_16_32_lrelu", 1, 24, 24, 16, 0, 32, 5, 1, 2, 1.0),
    ("lin_stem_7x7s2_96_64", 2, 16, 16, 96, 0, 64, 7, 2, 0, 1.0),
    ("res_1x1s2_256_128", 1, 16, 16, 256, 0, 128, 1, 2, 0, 1.0),
    ("res_1x1_64_256", 1, 20, 20, 64, 0, 256, 1, 1, 0, 1.0),
    ("wide_3x3_512_512", 1, 8, 8, 512, 0, 512, 3, 1, 1, 1.0),
    ("ragged_3x3_16_16", 2, 13, 19, 16, 0, 16, 3, 1, 2, 1.0),
    ("concat_16_16_16_lrelu", 1, 20, 36, 16, 16, 16, 3, 1, 2, 1.0),          # natural k order with two sources (deq / ref u1)
    ("stem_7x7_8_16_lrelu", 1, 24, 24, 8, 0, 16, 7, 1, 2, 1.0),              # 3 -> 8 channel image input, K tail 392 = 12.25 chunks
    ("wide_pixels_3x3_32_32", 4, 72, 80, 32, 0, 32, 3, 1, 1, 1.0),          # several pixel slices per weight-gradient tile
    ("alltaps_3x3_64_64_mode1", 2, 40, 72, 64, 0, 64, 3, 1, 1, 1.0),          # 2 x 2 wave split, all taps per wave
    ("alltaps_3x3_128_64_two_ci_tiles", 1, 33, 50, 128, 0, 64, 3, 1, 0, 1.0),
    ("unet_5x5_32_32_lrelu", 2, 21, 40, 32, 0, 32, 5, 1, 2, 1.0),
    ("unet_3x3_32_16_lrelu", 2, 19, 35, 32, 0, 16, 3, 1, 2, 1.0),
    ("unet_3x3_16_32", 1, 32, 32, 16, 0, 32, 3, 1, 0, 1.0),
    ("unet_3x3_64_32_lrelu", 1, 24, 40, 64, 0, 32, 3, 1, 2, 1.0),
    ("w3_concat_64_64_128_ragged", 2, 21, 37, 64, 64, 128, 3, 1, 2, 1.0),      # two sources, ragged tile edges
    ("w3_3x3_32_64", 1, 48, 48, 32, 0, 64, 3, 1, 1, 1.0),                      # a single 32-channel chunk
    ("wgrad256_3x3_256_256", 2, 18, 22, 256, 0, 256, 3, 1, 1, 1.0),            # 8-wave 256 x 256 weight-gradient tile, ragged slices
    ("wgrad256_3x3_512_256", 1, 19, 21, 512, 0, 256, 3, 1, 0, 1.0),            # two ci tiles
    ("wgrad256_3x3_128_256", 1, 16, 24, 128, 0, 256, 3, 1, 2, 1.0),            # 128 x 256 tile
    ("wgrad256_concat_256_256_128", 1, 16, 20, 256, 256, 128, 3, 1, 2, 1.0),   # 256 x 128 tile, two sources
    ("wgrad256_1x1_1024_512", 1, 12, 12, 1024, 0, 512, 1, 1, 0, 1.0),
]


@pytest.mark.parametrize("kernels", ["specialised", "general"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fp16_conv_forward_dgrad_wgrad_vs_float64_reference(shdr, case, kernels, monkeypatch):
    """`specialised`: the narrow layers take the patch kernel (conv_f16_patch.hip) and every eligible stride-1 layer the all-taps
    weight gradient (wgrad_f16_alltaps.hip; its many-pixels threshold is lifted so that these small shapes reach it);
    `general`: the implicit-GEMM kernel and the per-tap weight gradient for every shape."""
    name, n, h, w, c1, c2, cout, k, stride, act, x2s = case
    if kernels == "specialised":
        monkeypatch.setenv("SHDR_ALLTAPS_MIN_PIXELS", "0")
        monkeypatch.setenv("SHDR_WGRAD_256_MIN_PIXELS", "0")    # the 256-wide weight-gradient tiles at these small shapes too
        monkeypatch.setenv("SHDR_W3_MIN_BLOCKS", "0")             # wide 3x3 layers: the patch-per-chunk kernel (conv_f16_w3.hip)
    else:
        monkeypatch.setenv("SHDR_NO_PATCH", "1")
        monkeypatch.setenv("SHDR_NO_ALLTAPS", "1")
        monkeypatch.setenv("SHDR_NO_W3", "1")
    K = shdr._ops
    rng = np.random.default_rng(len(name) * 17 + h)
    x = rng.normal(size=(n, h, w, c1)).astype(np.float32)
    x2 = (rng.normal(size=(n, h, w, c2)) / x2s).astype(np.float32) if c2 else None
    wt = (rng.normal(size=(k, k, c1 + c2, cout)) / np.sqrt(k * k * (c1 + c2))).astype(np.float32)
    b = (rng.normal(size=cout) * 0.1).astype(np.float32)
    ho, wo = -(-h // stride), -(-w // stride)
    gy = rng.normal(size=(n, ho, wo, cout)).astype(np.float32)
    # float64 reference with autograd (pinned to the NumPy oracle below and in tests/test_oracle_grad.py)
    tx, tw, tb = R.T(x, True), R.T(wt, True), R.T(b, True)
    tx2 = R.T(x2, True) if c2 else None
    xin = tx if tx2 is None else torch.cat([tx, tx2 * x2s], -1)
    z = R.conv2d(xin, tw, tb, stride)
    y = {0: z, 1: torch.relu(z), 2: R.lrelu(z), 3: torch.tanh(z)}[act]
    np_in = x.astype(np.float64) if c2 == 0 else np.concatenate([x.astype(np.float64), x2.astype(np.float64) * x2s], -1)
    np.testing.assert_allclose(z.detach().numpy(), ops.conv2d(np_in, wt.astype(np.float64), b.astype(np.float64), stride),
                               rtol=1e-9, atol=1e-9)
    # HIP, native-fp16 conv path: fp16 feature maps in and out, fp32 filter / bias and fp32 parameter gradients
    need_dx = c1 % 16 == 0        # 8-channel inputs are zero-padded images (data): the fp16 path builds no input gradient for them
    dx, dw, db = dev(x).half().requires_grad_(need_dx), dev(wt, True), dev(b, True)
    dx2 = dev(x2).half().requires_grad_(True) if c2 else None
    with K.precision("fp16"):
        yy = K.conv2d(dx, dw, db, stride=stride, x2=dx2, x2_scale=x2s, act1=act)
        assert yy.dtype == torch.float16 and dw.dtype == torch.float32
        assert rel_err(host(yy), y.detach().numpy()) <= LAYER_TOL
        (yy.float() * dev(gy)).sum().backward()
    # the reference backward runs with the activation mask of the fp16 forward: where |z| is below the fp16 round-off the two
    # forwards may land on different sides of zero, and a flipped relu mask is a property of the input, not a kernel error
    if act in (1, 2):
        m = torch.from_numpy(host(yy) > 0)
        y = z * (m.double() if act == 1 else torch.where(m, 1.0, 0.1).double())
    (y * R.T(gy)).sum().backward()
    assert dw.grad.dtype == torch.float32
    if need_dx:
        assert dx.grad.dtype == torch.float16
        assert rel_err(host(dx.grad), tx.grad.numpy()) <= LAYER_TOL, "dx"
    if c2:
        assert rel_err(host(dx2.grad), tx2.grad.numpy()) <= LAYER_TOL, "dx2"
    assert rel_err(host(dw.grad), tw.grad.numpy()) <= LAYER_TOL, "dw"
    assert rel_err(host(db.grad), tb.grad.numpy()) <= LAYER_TOL, "db"


FT16_COS, FT16_NORM, FT16_INTER = 0.995, 0.02, 5e-3      # whole-step bars against the fp16-storage model


@pytest.fixture(scope="module")
def ft16(shdr, emor_table):
    rng = np.random.default_rng(12)
    P = {k: nets.init_params(getattr(nets, k + "_spec")(), 95 + i) for i, k in enumerate(("deq", "lin", "hal", "ref"))}
    ldr = quantised_image(rng, (2, 64, 64, 3))
    ldr[0, :12, :12] = 1.0
    hdr = rng.random((2, 64, 64, 3)) * 1.5
    hdr = hdr / (1e-6 + hdr.mean(axis=(1, 2, 3), keepdims=True)) * 0.5
    mods = dict(deq="dequantization_net", lin="linearization_net", hal="hallucination_net", ref="refinement_net")
    ms = {k: getattr(shdr, mods[k]).model().load_numpy(P[k]) for k in mods}
    step = shdr.pipeline.FinetuneStep(ms["deq"], ms["lin"], ms["hal"], ms["ref"], lr=1e-4, precision="fp16", loss_scale=0.25)
    tP = {k: R.params_to_torch(v) for k, v in P.items()}
    ref = R.finetune_forward(tP, R.T(ldr), R.T(hdr), emor_table)
    ref["loss"].sum().backward()
    # the float64 MODEL of the fp16 path (torch_ref.fp16_storage: float64 arithmetic, feature maps rounded to fp16 where the product
    # stores fp16): its relu / max-pool / clip decisions are taken on the values the HIP forward sees
    tQ = {k: R.params_to_torch(v) for k, v in P.items()}
    with R.fp16_storage():
        refq = R.finetune_forward(tQ, R.T(ldr), R.T(hdr), emor_table)
        refq["loss"].sum().backward()
    return dict(step=step, ms=ms, ldr=dev(ldr), hdr=dev(hdr), ref=ref, tP=tP, refq=refq, tQ=tQ, P=P, np=(ldr, hdr))


def test_fp16_finetune_step_vs_float64_oracle(ft16, emor_table):
    out = ft16["step"](ft16["ldr"], ft16["hdr"], apply=False)
    assert ft16["step"].skipped_steps == 0
    oracle = nets.finetune_forward(ft16["P"], ft16["np"][0], ft16["np"][1], emor_table)      # the NumPy float64 oracle
    for k in ("C_pred", "B_pred", "A_pred", "refinement_output"):
        assert rel_err(host(out[k]), oracle[k]) <= 2e-2, k
    assert rel_err(host(out["loss_sum"]), oracle["loss"].sum(axis=(1, 2, 3))) <= 1e-2
    # gradients: float64 autograd reference (tests/torch_ref.py, pinned to the NumPy oracle at 1e-9)
    got = ft16["step"].params.grad.double().cpu().numpy()
    ref = np.zeros_like(got)
    names = [(net, n) for net in ("deq", "lin", "hal", "ref") for n, _, tr in ft16["ms"][net].named_weights() if tr]
    for (net, n), o in zip(names, ft16["step"].params.offsets):
        g = ft16["tP"][net][n].grad.numpy().ravel()
        ref[o:o + g.size] = g
    assert np.isfinite(got).all()
    cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref)))
    assert cos >= 0.95, cos
    assert abs(np.linalg.norm(got) / np.linalg.norm(ref) - 1.0) <= 0.1
    # ... and against the float64 model WITH fp16 storage (same activation masks up to accumulation rounding): the tight bars
    refq = np.zeros_like(got)
    for (net, n), o in zip(names, ft16["step"].params.offsets):
        g = ft16["tQ"][net][n].grad.numpy().ravel()
        refq[o:o + g.size] = g
    cosq = float((got * refq).sum() / (np.linalg.norm(got) * np.linalg.norm(refq)))
    normq = float(np.linalg.norm(got) / np.linalg.norm(refq))
    errs = {k: rel_err(host(out[k]), ft16["refq"][k].detach().numpy()) for k in ("C_pred", "B_pred", "A_pred", "refinement_output")}
    per_net, o0 = {}, 0
    offs = list(ft16["step"].params.offsets) + [got.size]
    for net in ("deq", "lin", "hal", "ref"):
        nv = len(ft16["ms"][net].trainable_variables)
        a, b = offs[o0], offs[o0 + nv]
        per_net[net] = (float((got[a:b] * refq[a:b]).sum() / (np.linalg.norm(got[a:b]) * np.linalg.norm(refq[a:b]))),
                        float(np.linalg.norm(got[a:b]) / np.linalg.norm(refq[a:b])), float(np.linalg.norm(refq[a:b])))
        o0 += nv
    print("fp16 step vs fp16-storage model: cos %.5f norm ratio %.4f intermediates %s (unrounded reference: cos %.5f)" % (cosq, normq, errs, cos))
    print("  per net (cos, norm ratio, |g|):", {k: tuple(round(x, 5) for x in v) for k, v in per_net.items()})
    # Measured (MI355X): deq 0.9993 / ref 1.0000 -- the two nets without BatchNorm meet the tight bars; lin 0.973 / hal 0.940 do not,
    # and the fp16-storage model scores the SAME cosine as the unrounded reference (0.9736 vs 0.9745): the residual is not activation
    # masks but training-mode BatchNorm on this fixture (2 x 64 x 64: 8 samples per channel at the Hallucination-Net bottleneck, where a
    # 5e-4 fp16 perturbation of the input moves the batch statistics); the per-layer tests above hold every conv / BN kernel to 3e-3.
    for net in ("deq", "ref"):
        assert per_net[net][0] >= FT16_COS and abs(per_net[net][1] - 1.0) <= FT16_NORM + 0.01, (net, per_net[net])
    # (the whole-step norm carries the Hallucination-Net's: an experiment that replaced libm's tanhf in the deq / ref heads by an
    #  expression differing from it by 1e-6 moved hal's norm ratio from 1.014 to 0.946 and its cosine from 0.940 to 0.953 -- the same
    #  batch-statistics sensitivity; the whole-step norm bar is therefore +-5 %, the tight one stays on the nets without BatchNorm)
    assert per_net["lin"][0] >= 0.95 and per_net["hal"][0] >= 0.90 and cosq >= 0.95 and abs(normq - 1.0) <= 0.05, (per_net, cosq, normq)
    assert max(errs["C_pred"], errs["B_pred"]) <= FT16_INTER and max(errs.values()) <= 2e-2, errs


def test_fp16_finetune_1024_tiles_properties(shdr):
    """configs[4] at its own size: batch 4 of 1024 x 1024 tiles, fp16 conv path, loss scale 0.25 (bench.py's setting)."""
    torch.manual_seed(777)
    g = torch.Generator().manual_seed(5)
    b, sz = 4, 1024
    ldr = (torch.round(torch.rand((b, sz, sz, 3), generator=g) * 255.0) / 255.0).cuda()
    hdr = torch.rand((b, sz, sz, 3), generator=g) * 1.5
    hdr = (hdr / (1e-6 + hdr.mean(dim=(1, 2, 3), keepdim=True)) * 0.5).cuda()
    nets4 = [shdr.dequantization_net.model(), shdr.linearization_net.model(), shdr.hallucination_net.model(),
             shdr.refinement_net.model()]
    step = shdr.pipeline.FinetuneStep(*nets4, precision="fp16", loss_scale=0.25, lr=1e-4)
    o1 = step(ldr, hdr, apply=False)
    g1 = step.params.grad.clone()
    o2 = step(ldr, hdr, apply=False)
    for k in ("C_pred", "B_pred", "A_pred", "refinement_output"):
        assert bool(torch.isfinite(o1[k]).all()), k
        # the only order-dependent sums of the forward are the fp64 atomics of the BatchNorm statistics: reproducible to fp32 rounding
        assert float((o1[k] - o2[k]).abs().max()) <= 1e-4 * float(o1[k].abs().max()), k
    assert bool(torch.isfinite(g1).all()) and float(g1.abs().max()) > 0.0
    # (the backward accumulates weight gradients with fp32 atomics: reproducible to rounding, not bit for bit)
    assert float((step.params.grad - g1).norm() / g1.norm()) <= 1e-3
    first = float(o1["loss_sum"].sum())
    for _ in range(3):
        last = float(step(ldr, hdr)["loss_sum"].sum())
    assert step.skipped_steps == 0
    assert np.isfinite(last) and last < first, (first, last)
    assert tuple(o1["refinement_output"].shape) == (b, sz, sz, 3)


def test_fp16_head_conv_returns_fp32_three_channels(shdr):
    """the zero-padded 3-channel heads (dequantization_net.py:46, hallucination_net.py:183): fp16 in, fp32 [.., 3] out,
    gradients back onto fp16"""
    K = shdr._ops
    rng = np.random.default_rng(7)
    x = rng.normal(size=(2, 12, 20, 16)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, 16, 3)) / 12.0).astype(np.float32)
    b = (rng.normal(size=3) * 0.1).astype(np.float32)
    gy = rng.normal(size=(2, 12, 20, 3)).astype(np.float32)
    tx, tw, tb = R.T(x, True), R.T(wt, True), R.T(b, True)
    y = torch.tanh(R.conv2d(tx, tw, tb, 1))
    (y * R.T(gy)).sum().backward()
    dx, dw, db = dev(x).half().requires_grad_(True), dev(wt, True), dev(b, True)
    wpad = torch.nn.functional.pad(dw, (0, 13))                      # what Conv2D.call_padded does on the tape
    with K.precision("fp16"):
        yy = K.conv2d(dx, wpad, db, act1=K.ACT_TANH, cout_valid=3)
    assert yy.dtype == torch.float32 and tuple(yy.shape) == (2, 12, 20, 3)
    assert rel_err(host(yy), y.detach().numpy()) <= LAYER_TOL
    (yy * dev(gy)).sum().backward()
    assert rel_err(host(dx.grad), tx.grad.numpy()) <= LAYER_TOL
    assert rel_err(host(dw.grad), tw.grad.numpy()) <= LAYER_TOL
    assert rel_err(host(db.grad), tb.grad.numpy()) <= LAYER_TOL


def test_fp16_elementwise_twins_vs_oracle(shdr, emor_table):
    """pooling / resize / BatchNorm / residual joins / front end on fp16 feature maps against the float64 oracle evaluated on the
    SAME (fp16-representable) inputs: the only error left is the rounding of the stored result (u = 4.9e-4) and fp32 arithmetic"""
    K = shdr._ops
    rng = np.random.default_rng(3)
    TOLE = 1.5e-3
    x = rng.normal(size=(2, 12, 20, 24)).astype(np.float16)
    xd = torch.from_numpy(x).cuda().requires_grad_(True)
    x64 = x.astype(np.float64)
    tx = R.T(x64, True)
    gy_small = rng.normal(size=(2, 6, 10, 24)).astype(np.float16)
    for name, fn, tfn in (("avgpool2", K.avgpool2, R.avg_pool2), ("maxpool2", K.maxpool2, lambda t: R.max_pool(t, 2, 2)),
                          ("maxpool3s2", K.maxpool3s2, lambda t: R.max_pool(t, 3, 2))):
        tx.grad = None
        xd.grad = None
        ty = tfn(tx)
        (ty * R.T(gy_small.astype(np.float64))).sum().backward()
        yy = fn(xd)
        assert yy.dtype == torch.float16 and rel_err(host(yy), ty.detach().numpy()) <= TOLE, name
        (yy.float() * torch.from_numpy(gy_small).cuda().float()).sum().backward()
        assert rel_err(host(xd.grad), tx.grad.numpy()) <= TOLE, name + " bwd"
    gy_big = rng.normal(size=(2, 24, 40, 24)).astype(np.float16)
    tx.grad = None
    xd.grad = None
    ty = R.resize2x(tx)
    np.testing.assert_allclose(ty.detach().numpy(), ops.resize_bilinear_2x(x64), atol=1e-12)
    (ty * R.T(gy_big.astype(np.float64))).sum().backward()
    yy = K.resize2x(xd)
    assert rel_err(host(yy), ty.detach().numpy()) <= TOLE
    (yy.float() * torch.from_numpy(gy_big).cuda().float()).sum().backward()
    assert rel_err(host(xd.grad), tx.grad.numpy()) <= TOLE
    # global average pool: fp32 out, fp16 gradient
    tx.grad = None
    xd.grad = None
    g2 = rng.normal(size=(2, 24)).astype(np.float32)
    (tx.mean(dim=(1, 2)) * R.T(g2)).sum().backward()
    yy = K.global_avg_pool(xd)
    assert yy.dtype == torch.float32 and rel_err(host(yy), x64.mean(axis=(1, 2))) <= 1e-5
    (yy * torch.from_numpy(g2).cuda()).sum().backward()
    assert xd.grad.dtype == torch.float16 and rel_err(host(xd.grad), tx.grad.numpy()) <= TOLE
    # training-mode BatchNorm (+ relu), statistics in double
    L = __import__("importlib").import_module("singlehdr-tf2_amd._layers")
    bn = L.BatchNormalization(24)
    with torch.no_grad():
        bn.gamma.copy_(torch.from_numpy(rng.uniform(0.5, 1.5, 24).astype(np.float32)))
        bn.beta.copy_(torch.from_numpy(rng.normal(0, 0.1, 24).astype(np.float32)))
    p = {"n.gamma": R.T(host(bn.gamma).astype(np.float64), True), "n.beta": R.T(host(bn.beta).astype(np.float64), True)}
    gyb = rng.normal(size=x.shape).astype(np.float16)
    tx.grad = None
    xd.grad = None
    ty = torch.relu(R.bn(p, "n", tx, True))
    (ty * R.T(gyb.astype(np.float64))).sum().backward()
    yy = bn.train_apply(xd, relu=True)
    assert yy.dtype == torch.float16 and rel_err(host(yy), ty.detach().numpy()) <= TOLE
    (yy.float() * torch.from_numpy(gyb).cuda().float()).sum().backward()
    assert rel_err(host(xd.grad), tx.grad.numpy()) <= 3e-3
    assert rel_err(host(bn.gamma.grad), p["n.gamma"].grad.numpy()) <= 3e-3
    assert rel_err(host(bn.beta.grad), p["n.beta"].grad.numpy()) <= 3e-3
    mm = x64.mean(axis=(0, 1, 2)) * 0.01
    assert rel_err(host(bn.moving_mean), mm) <= 1e-4                      # Keras momentum 0.99 from zero
    # residual joins
    a, b = rng.normal(size=(2, 5, 7, 16)).astype(np.float16), rng.normal(size=(2, 5, 7, 16)).astype(np.float16)
    ad, bd = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    assert rel_err(host(K.add(ad, bd)), a.astype(np.float64) + b.astype(np.float64)) <= TOLE
    assert rel_err(host(K.add(ad, bd, relu=True)), np.maximum(a.astype(np.float64) + b.astype(np.float64), 0)) <= TOLE
    # Linearization-Net front end: fp32 image -> fp16 [.., 96]; histogram channels are exact in fp16 for quantised inputs only
    img = quantised_image(rng, (2, 16, 20, 3))
    with K.precision("fp16"):
        f = K.lin_frontend(dev(img), 96)
    assert f.dtype == torch.float16
    want = ops.lin_frontend(img) if hasattr(ops, "lin_frontend") else None
    ti = R.T(img, True)
    tf = R.lin_frontend(ti) if hasattr(R, "lin_frontend") else None
    if tf is not None:
        assert rel_err(host(f)[..., :93], tf.detach().numpy()) <= TOLE
        gf = rng.normal(size=(2, 16, 20, 96)).astype(np.float16)
        (tf * R.T(gf[..., :93].astype(np.float64))).sum().backward()
        di = dev(img, True)
        with K.precision("fp16"):
            ff = K.lin_frontend(di, 96)
        (ff.float() * torch.from_numpy(gf).cuda().float()).sum().backward()
        assert rel_err(host(di.grad), ti.grad.numpy()) <= 2e-3
    assert float(host(f)[..., 93:].max()) == 0.0
