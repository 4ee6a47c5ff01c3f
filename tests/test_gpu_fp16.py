"""GPU tests of the round-1 reduced-precision OPERAND modes ("fp16op" / "bf16"; the native-fp16 path of BASELINE configs[4] is
tested against the float64 oracle in tests/test_gpu_fp16_oracle.py).

Tensors stay fp32 in HBM; the conv forward / dgrad / wgrad kernels round activations, filters and output
gradients to fp16 (bf16) when they pack the operands of v_mfma_f32_16x16x32_{f16,bf16} and accumulate in fp32.
The tolerance of this configuration is fp16-level, not the 1e-4 of the fp32 parity path:
  * one layer: |y16 - y32| <= 2e-3 * max|y32|  (fp16, unit round-off 4.9e-4),  2e-2 (bf16, 3.9e-3);
  * the chained fine-tuning step: losses within 1e-2 relative of the fp32 step, whole-net gradients within the
    same direction (cosine >= 0.97 over all 29 M parameters) and the step still reduces the loss.
"""
import numpy as np
import pytest
import torch

from conftest import quantised_image
from oracle import nets

pytestmark = pytest.mark.gpu
TOL = {"fp16op": 2e-3, "bf16": 2e-2}


def maxrel(a, b):
    a, b = a.detach(), b.detach()
    return float((a - b).abs().max() / b.abs().max())


SHAPES = [  # n, h, w, c1, c2, cout, k, stride
    (2, 24, 20, 64, 0, 128, 3, 1),     # FAST chunking, <128,128> tile
    (2, 17, 23, 32, 32, 64, 3, 1),     # two sources on a 32-channel boundary, ragged tile edges
    (1, 32, 32, 16, 0, 16, 7, 1),      # natural k order (Ct < 32), <128,16> tile
    (2, 16, 16, 96, 0, 64, 7, 2),      # strided 7x7 (Linearization-Net conv1)
    (2, 16, 16, 12, 0, 16, 3, 1),      # K tail: 108 is not a multiple of 32
    (1, 20, 20, 64, 0, 32, 1, 1),      # 1x1
]


@pytest.mark.parametrize("prec", ["fp16op", "bf16"])
@pytest.mark.parametrize("shape", SHAPES)
def test_conv_forward_reduced_precision(shdr, prec, shape):
    K = shdr._ops
    n, h, w, c1, c2, cout, k, s = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn((n, h, w, c1), generator=g).cuda()
    x2 = torch.randn((n, h, w, c2), generator=g).cuda() if c2 else None
    wt = (torch.randn((k, k, c1 + c2, cout), generator=g) / (k * (c1 + c2) ** 0.5)).cuda()
    b = torch.randn((cout,), generator=g).cuda()
    K.WINOGRAD, saved = False, K.WINOGRAD
    try:
        ref = K.conv2d(x, wt, b, stride=s, x2=x2, act1=K.ACT_LRELU)
        with K.precision(prec):
            got = K.conv2d(x, wt, b, stride=s, x2=x2, act1=K.ACT_LRELU)
        forced = K.conv2d(x, wt, b, stride=s, x2=x2, act1=K.ACT_LRELU,
                          algo=K.ALGO_MFMA_F16 if prec == "fp16op" else K.ALGO_MFMA_BF16)
    finally:
        K.WINOGRAD = saved
    err = maxrel(forced, ref)
    assert 0.0 < err <= TOL[prec], err                    # > 0: the reduced-precision kernel really ran
    if c2 == 0 and cout == 16 and c1 <= 16 and s == 1:
        # narrow single-source layers: AUTO_* keeps the exact-fp32 register-A kernel (faster than the fp16-operand kernel there)
        assert torch.equal(got, ref)
    else:
        assert torch.equal(got, forced)                   # AUTO_* takes the reduced-precision MFMA path on these shapes


def test_fp16_operands_are_rounded_to_nearest_even(shdr):
    """1x1 conv with a one-hot filter = the kernel's fp32 -> fp16 operand conversion, observable exactly."""
    K = shdr._ops
    x = torch.zeros((1, 4, 16, 32), device="cuda")
    vals = torch.tensor([1.0 + 2.0 ** -11, 1.0 + 3 * 2.0 ** -11, 1.0 + 2.0 ** -12, 65504.0, 1e-7, -0.3333333])
    x[0, 0, :6, 0] = vals.cuda()
    w = torch.zeros((1, 1, 32, 16), device="cuda")
    w[0, 0, 0, 0] = 1.0
    y = K.conv2d(x, w, algo=K.ALGO_MFMA_F16)[0, 0, :6, 0].cpu()
    assert torch.equal(y, vals.half().float())            # torch .half() is round-to-nearest-even
    yb = K.conv2d(x, w, algo=K.ALGO_MFMA_BF16)[0, 0, :6, 0].cpu()
    assert torch.equal(yb, vals.bfloat16().float())


def test_reduced_precision_needs_folded_x2_scale(shdr):
    K = shdr._ops
    x = torch.randn((1, 8, 8, 32), device="cuda")
    w = torch.randn((1, 1, 64, 16), device="cuda")
    with pytest.raises(RuntimeError, match="x2_scale"):
        K.conv2d(x, w, x2=x, x2_scale=0.5, algo=K.ALGO_MFMA_F16)
    # AUTO_F16 keeps such a layer on the fp32 kernels instead
    with K.precision("fp16op"):
        y = K.conv2d(x, w, x2=x, x2_scale=0.5)
    # (the fp32 path folds the scale into the prepared filter below the ABI: equal up to one rounding of w * 0.5 -- exact here)
    assert float((y - K.conv2d(x, w, x2=x, x2_scale=0.5)).abs().max()) <= 1e-5 * float(y.abs().max())


@pytest.mark.parametrize("prec", ["fp16op", "bf16"])
def test_conv_backward_reduced_precision(shdr, prec):
    """dgrad and wgrad (MFMA over pixels) with reduced-precision operands vs the exact-fp32 kernels."""
    K = shdr._ops
    for n, h, w, c1, c2, cout, k, s in SHAPES[:3] + [(2, 16, 16, 32, 0, 64, 1, 2)]:
        g = torch.Generator().manual_seed(7)
        xs = [torch.randn((n, h, w, c), generator=g).cuda().requires_grad_(True) for c in (c1, c2) if c]
        wt = (torch.randn((k, k, c1 + c2, cout), generator=g) / (k * (c1 + c2) ** 0.5)).cuda().requires_grad_(True)
        gy = None
        grads = {}
        for p in ("fp32", prec):
            for t in xs + [wt]:
                t.grad = None
            with K.precision(p):
                y = K.conv2d(xs[0], wt, stride=s, x2=xs[1] if c2 else None, x2_scale=0.25 if c2 else 1.0)   # no activation: a relu mask would flip where y ~ 0
                if gy is None:
                    gy = torch.randn(y.shape, generator=g).cuda()
                (y * gy).sum().backward()
            grads[p] = [t.grad.clone() for t in xs + [wt]]
        errs = [maxrel(a, b) for a, b in zip(grads[prec], grads["fp32"])]
        assert all(e <= 2 * TOL[prec] for e in errs), errs
        # dgrad runs on reduced-precision operands, except for the narrow single-source layers (<= 16 channels -> 16) that keep
        # the exact-fp32 register-A kernel in every mode; the weight gradient of the narrow layers (16 / 32 channels) takes the
        # exact-fp32 all-taps kernel in every mode (both faster than the fp16-operand narrow kernels), so those errors may be 0
        if not (c2 == 0 and c1 <= 16 and cout == 16 and s == 1):
            assert errs[0] > 0.0, errs


@pytest.fixture(scope="module")
def steps(shdr):
    rng = np.random.default_rng(21)
    P = {k: nets.init_params(getattr(nets, k + "_spec")(), 70 + i) for i, k in enumerate(("deq", "lin", "hal", "ref"))}
    ldr = quantised_image(rng, (2, 64, 64, 3))
    ldr[0, :12, :12] = 1.0
    hdr = rng.random((2, 64, 64, 3)) * 1.5
    hdr = hdr / (1e-6 + hdr.mean(axis=(1, 2, 3), keepdims=True)) * 0.5
    mods = dict(deq="dequantization_net", lin="linearization_net", hal="hallucination_net", ref="refinement_net")
    out = {}
    for prec in ("fp32", "fp16op", "bf16"):
        ms = {k: getattr(shdr, mods[k]).model().load_numpy(P[k]) for k in mods}
        out[prec] = (shdr.pipeline.FinetuneStep(ms["deq"], ms["lin"], ms["hal"], ms["ref"], lr=1e-4, precision=prec,
                                                loss_scale=1.0 if prec == "fp32" else 0.25), ms)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    return out, t(ldr), t(hdr)


@pytest.mark.parametrize("prec", ["fp16op", "bf16"])
def test_finetune_step_reduced_precision_tracks_fp32(steps, prec):
    out, ldr, hdr = steps
    ref = out["fp32"][0](ldr, hdr, apply=False)
    got = out[prec][0](ldr, hdr, apply=False)
    scale = 1.0 if prec == "fp16op" else 16.0      # bf16 keeps 8 mantissa bits: ~8x the fp16 round-off per layer
    for k in ("C_pred", "B_pred", "A_pred", "refinement_output"):
        assert maxrel(got[k], ref[k]) <= 2e-2 * scale, k     # 0.6-1.2e-2 observed, depending on which layers run exact
    assert maxrel(got["loss_sum"], ref["loss_sum"]) <= 1e-2 * scale
    g32, g16 = out["fp32"][0].params.grad.double(), out[prec][0].params.grad.double()
    assert torch.isfinite(g16).all()
    # whole-net gradients of these randomly initialised training-mode nets move by 1-3.5 % in L2 under a 1e-6 input
    # perturbation (relu / max-pool masks flip, test_gpu_grad.py); operand rounding is a 5e-4 (4e-3) perturbation of
    # every layer, so the bar here is the DIRECTION of the flat 29 M-element gradient, and that the step still descends
    cos = float((g16 * g32).sum() / (g16.norm() * g32.norm()))
    assert cos >= (0.97 if prec == "fp16op" else 0.85), cos


def test_finetune_fp16_reduces_the_loss(steps):
    out, ldr, hdr = steps
    step = out["fp16op"][0]
    first = float(step(ldr, hdr)["loss_sum"].detach().sum())
    for _ in range(4):
        last = float(step(ldr, hdr)["loss_sum"].detach().sum())
    assert np.isfinite(last) and last < first, (first, last)


def test_fp16_overflow_skips_the_step(steps):
    """An absurd loss scale overflows the fp16 output-gradient operands: the step is dropped, weights untouched."""
    out, ldr, hdr = steps
    step = out["fp16op"][0]
    before = step.params.flat.clone()
    saved, step.loss_scale = step.loss_scale, 1e9
    try:
        step(ldr, hdr)
    finally:
        step.loss_scale = saved
    assert step.skipped_steps == 1 and torch.equal(step.params.flat, before)
