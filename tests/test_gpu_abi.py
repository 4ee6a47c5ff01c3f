"""The drop-in boundary driven with NOTHING but ctypes (no singlehdr-tf2_amd/_ops.py): a host that binds libshdr.so the way
INTEGRATION.md section 2 shows gets the fast kernels -- the dispatch policy lives below the C ABI (csrc/conv_plan.hip).

  * a Hallucination-Net layer (d2.conv2: 3x3 128 -> 128, hallucination_net.py:47-49) through plan / prepare / fwd_prepared: the
    plan is the one-kernel Winograd, the result matches the float64 oracle, the conv + MaxPool2D pair is one launch, and the call
    costs what the raw fused kernel costs;
  * the input gradients of Dequantization-Net layers (dequantization_net.py:35-46) and of the strided Linearization-Net convs
    (linearization_net.py:12,16,91) through shdr_conv2d_dgrad_f32 against the float64 autograd reference.
torch is used for device memory only."""
import ctypes
import os

import numpy as np
import pytest
import torch

import torch_ref as R
from conftest import rel_err
from oracle import ops

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class ConvDesc(ctypes.Structure):            # the mirror of INTEGRATION.md section 2
    _fields_ = [(n, ctypes.c_int32) for n in ("N", "H", "W", "C1", "C2", "Cout", "KH", "KW", "stride",
                                              "pad_t", "pad_l", "Ho", "Wo")] + \
               [("x2_scale", ctypes.c_float)] + \
               [(n, ctypes.c_int32) for n in ("act1", "act2", "res_cstride", "y_cstride", "algo", "cout_valid")] + \
               [("w_batch_stride", ctypes.c_int64)] + \
               [(n, ctypes.c_int32) for n in ("y_pix_stride", "y_off_h", "y_off_w", "y_H", "y_W", "prologue", "pool")]


@pytest.fixture(scope="module")
def lib():
    L = ctypes.CDLL(os.path.join(ROOT, "singlehdr-tf2_amd", "libshdr.so"))
    P, D, I, V = ctypes.POINTER(ConvDesc), ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p
    for f, res, args in (("shdr_conv2d_plan_f32", I, [P, I]), ("shdr_conv2d_prepared_filter_elems_f32", ctypes.c_int64, [P, I]),
                         ("shdr_conv2d_filter_is_plain_f32", I, [P, I]), ("shdr_conv2d_prepare_filter_f32", I, [P, I, V, V, V]),
                         ("shdr_conv2d_workspace_bytes_f32", ctypes.c_int64, [P, I]), ("shdr_conv2d_fwd_prepared_f32", I, [P] + [V] * 11),
                         ("shdr_conv2d_fwd_prepared_ranged_f32", I, [P] + [V] * 14), ("shdr_absmax_f32", I, [V, ctypes.c_int64, V, V]),
                         ("shdr_conv2d_dgrad_workspace_bytes_f32", ctypes.c_int64, [P, I]), ("shdr_conv2d_dgrad_f32", I, [P, I] + [V] * 5),
                         ("shdr_conv2d_winograd_fused2_f32", I, [V] * 8 + [I] * 8 + [V]), ("shdr_same_pad", I, [I, I, I, ctypes.POINTER(I), ctypes.POINTER(I)]),
                         ("shdr_workspace_bytes", ctypes.c_int64, [I, P, I]), ("shdr_last_error", ctypes.c_char_p, [])):
        getattr(L, f).restype, getattr(L, f).argtypes = res, args
    return L


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(lib, rc):
    assert rc == 0, lib.shdr_last_error().decode()


def same_desc(lib, n, h, w, c1, c2, cout, k, stride, **kw):
    ho, wo, pt, pl = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    lib.shdr_same_pad(h, k, stride, ctypes.byref(ho), ctypes.byref(pt))
    lib.shdr_same_pad(w, k, stride, ctypes.byref(wo), ctypes.byref(pl))
    return ConvDesc(N=n, H=h, W=w, C1=c1, C2=c2, Cout=cout, KH=k, KW=k, stride=stride, pad_t=pt.value, pad_l=pl.value, Ho=ho.value,
                    Wo=wo.value, x2_scale=1.0, **kw)


def test_hal_layer_through_plan_prepare_run(lib):
    rng = np.random.default_rng(0)
    n, h, w, c = 2, 40, 56, 128
    x = rng.normal(size=(n, h, w, c)).astype(np.float32)
    wt = (rng.normal(size=(3, 3, c, c)) / np.sqrt(9 * c)).astype(np.float32)
    b = (rng.normal(size=c) * 0.1).astype(np.float32)
    d = same_desc(lib, n, h, w, c, 0, c, 3, 1, act1=1)                       # SHDR_ACT_RELU
    assert lib.shdr_conv2d_plan_f32(ctypes.byref(d), 0) == 2                 # SHDR_PLAN_WINOGRAD_FUSED
    assert lib.shdr_conv2d_filter_is_plain_f32(ctypes.byref(d), 0) == 0
    assert lib.shdr_conv2d_workspace_bytes_f32(ctypes.byref(d), 0) == 0 == lib.shdr_workspace_bytes(0, ctypes.byref(d), 0)
    xd, wd, bd = dev(x), dev(wt), dev(b)
    prepared = torch.empty(lib.shdr_conv2d_prepared_filter_elems_f32(ctypes.byref(d), 0), device="cuda")
    check(lib, lib.shdr_conv2d_prepare_filter_f32(ctypes.byref(d), 0, ptr(wd), ptr(prepared), stream()))
    y = torch.empty((n, h, w, c), device="cuda")
    yp = torch.empty((n, h // 2, w // 2, c), device="cuda")
    check(lib, lib.shdr_conv2d_fwd_prepared_f32(ctypes.byref(d), ptr(xd), None, ptr(prepared), ptr(bd), None, None, None, ptr(y), ptr(yp),
                                                None, stream()))
    ref = np.maximum(ops.conv2d(x.astype(np.float64), wt.astype(np.float64), b.astype(np.float64), 1), 0)
    assert rel_err(y.cpu().numpy(), ref) <= 1e-5
    assert rel_err(yp.cpu().numpy(), ops.max_pool(ref, 2, 2)) <= 1e-5       # conv + MaxPool2D(2) pair: one launch
    # a fused residual takes the layer off the Winograd path (its epilogue has none): still one call, same result + residual
    res = rng.normal(size=(n, h, w, c)).astype(np.float32)
    d2 = same_desc(lib, n, h, w, c, 0, c, 3, 1, act1=1, res_cstride=c)
    assert lib.shdr_conv2d_plan_f32(ctypes.byref(d2), 1) == 1                # SHDR_PLAN_MFMA
    assert lib.shdr_conv2d_filter_is_plain_f32(ctypes.byref(d2), 1) == 1     # the HWIO filter itself is the prepared filter
    y2, resd = torch.empty_like(y), dev(res)
    check(lib, lib.shdr_conv2d_fwd_prepared_f32(ctypes.byref(d2), ptr(xd), None, ptr(wd), ptr(bd), None, None, ptr(resd), ptr(y2), None,
                                                None, stream()))
    assert rel_err(y2.cpu().numpy(), ref + res) <= 1e-5


def test_planned_call_costs_what_the_fused_kernel_costs(lib):
    """hal d2.conv2 at its bench size (batch 16 x 256 x 256 x 128 -> 128): the ABI-level call IS the fused Winograd launch"""
    n, h, w, c = 16, 256, 256, 128
    g = torch.Generator().manual_seed(1)
    xd = torch.randn((n, h, w, c), generator=g).cuda()
    wd = (torch.randn((3, 3, c, c), generator=g) / (9 * c) ** 0.5).cuda()
    d = same_desc(lib, n, h, w, c, 0, c, 3, 1, act1=1, algo=8)          # SHDR_ALGO_AUTO_EXACT: the exact-fp32 plan of this layer
    assert lib.shdr_conv2d_plan_f32(ctypes.byref(d), 0) == 2            # SHDR_PLAN_WINOGRAD_FUSED
    prepared = torch.empty(lib.shdr_conv2d_prepared_filter_elems_f32(ctypes.byref(d), 0), device="cuda")
    check(lib, lib.shdr_conv2d_prepare_filter_f32(ctypes.byref(d), 0, ptr(wd), ptr(prepared), stream()))
    y = torch.empty((n, h, w, c), device="cuda")

    def planned():
        check(lib, lib.shdr_conv2d_fwd_prepared_f32(ctypes.byref(d), ptr(xd), None, ptr(prepared), None, None, None, None, ptr(y), None, None, stream()))

    def raw():
        check(lib, lib.shdr_conv2d_winograd_fused2_f32(ptr(xd), None, ptr(prepared), None, None, None, ptr(y), None, n, h, w, c, 0, c, 1, 0, stream()))

    def timeit(fn):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10
    t_raw, t_planned = timeit(raw), timeit(planned)
    assert t_planned <= 1.05 * t_raw + 0.02, (t_planned, t_raw)
    tflops = 2.0 * n * h * w * c * c * 9 / (t_planned * 1e-3) / 1e12
    assert tflops >= 150.0, tflops            # direct-form FLOPs: far above what the direct kernel reaches (~125)
    # SHDR_ALGO_AUTO plans the split-operand fp16 kernel for this layer: same call sequence, fp32-level result, faster again
    y_exact = y.clone()
    dx = same_desc(lib, n, h, w, c, 0, c, 3, 1, act1=1)
    assert lib.shdr_conv2d_plan_f32(ctypes.byref(dx), 0) == 4           # SHDR_PLAN_X3
    px = torch.empty(lib.shdr_conv2d_prepared_filter_elems_f32(ctypes.byref(dx), 0), device="cuda")
    check(lib, lib.shdr_conv2d_prepare_filter_f32(ctypes.byref(dx), 0, ptr(wd), ptr(px), stream()))

    # the split-operand plans want the workspace the library asks for (scratch range slots: the range of x is measured when the caller
    # has none) ...
    nws = lib.shdr_conv2d_workspace_bytes_f32(ctypes.byref(dx), 0)
    assert nws >= 8
    ws = torch.empty(nws, dtype=torch.uint8, device="cuda")

    def planned_x3():
        check(lib, lib.shdr_conv2d_fwd_prepared_f32(ctypes.byref(dx), ptr(xd), None, ptr(px), None, None, None, None, ptr(y), None, ptr(ws), stream()))
    assert lib.shdr_conv2d_fwd_prepared_f32(ctypes.byref(dx), ptr(xd), None, ptr(px), None, None, None, None, ptr(y), None, None, stream()) == -5
    t_x3 = timeit(planned_x3)
    assert float((y - y_exact).abs().max()) <= 2e-5 * float(y_exact.abs().max())
    assert t_x3 <= t_planned, (t_x3, t_planned)
    # ... or the range slot of x (here from shdr_absmax_f32; in a network, the y_range of the producing layer): no measuring pass, no
    # workspace, and max |y| comes back in y_range for the next layer
    slots = torch.zeros(2, device="cuda")
    check(lib, lib.shdr_absmax_f32(ptr(xd), xd.numel(), ptr(slots), stream()))
    assert float(slots[0]) == float(xd.abs().max())
    y_measured = y.clone()

    def ranged_x3():
        check(lib, lib.shdr_conv2d_fwd_prepared_ranged_f32(ctypes.byref(dx), ptr(xd), None, ptr(px), None, None, None, None, ptr(y), None, None,
                                                           ptr(slots), None, ctypes.c_void_p(slots.data_ptr() + 4), stream()))
    t_ranged = timeit(ranged_x3)
    assert torch.equal(y, y_measured) and float(slots[1]) == float(y.abs().max())
    for _ in range(2):          # the chip is power-limited under this kernel and its clock drifts: alternate the two, keep the best of each
        t_x3, t_ranged = min(t_x3, timeit(planned_x3)), min(t_ranged, timeit(ranged_x3))
    assert t_ranged <= 1.03 * t_x3 + 0.01, (t_ranged, t_x3)      # (the measuring pass it saves is 3 % of this layer; run-to-run noise is 2 %)


DGRAD_CASES = [
    # name, N, H, W, C1, C2, Cout (filter width), cout_valid, k, stride, x2_scale
    ("deq_conv2_7x7_16_16", 1, 20, 24, 16, 0, 16, 16, 7, 1, 1.0),
    ("deq_d2_5x5_32_32", 2, 12, 16, 32, 0, 32, 32, 5, 1, 1.0),
    ("deq_u1_concat_16_16_16", 1, 18, 22, 16, 16, 16, 16, 3, 1, 1.0),
    ("deq_out_head_16_3", 1, 16, 16, 16, 0, 16, 3, 3, 1, 1.0),
    ("deq_u4_3x3_256_128_winograd", 1, 8, 12, 256, 0, 128, 128, 3, 1, 1.0),
    ("hal_skip_1x1_64_64_scaled", 1, 10, 10, 64, 64, 64, 64, 1, 1, 1.0 / 255),
    ("lin_res4_1x1s2_256_128", 1, 16, 16, 256, 0, 128, 128, 1, 2, 1.0),
    ("lin_stem_7x7s2_96_64", 2, 18, 22, 96, 0, 64, 64, 7, 2, 1.0),
    ("lin_stem_7x7s2_odd", 1, 17, 21, 96, 0, 64, 64, 7, 2, 1.0),
]


@pytest.mark.parametrize("case", DGRAD_CASES, ids=[c[0] for c in DGRAD_CASES])
def test_dgrad_entry_point_vs_float64_reference(lib, case):
    name, n, h, w, c1, c2, cout, cv, k, stride, x2s = case
    rng = np.random.default_rng(len(name))
    wt = (rng.normal(size=(k, k, c1 + c2, cout)) / np.sqrt(k * k * (c1 + c2))).astype(np.float32)
    wt[..., cv:] = 0.0                                                       # zero-padded filter columns of a narrow head
    d = same_desc(lib, n, h, w, c1, c2, cout, k, stride, cout_valid=cv)
    d.x2_scale = x2s
    dz = rng.normal(size=(n, d.Ho, d.Wo, cv)).astype(np.float32)
    tx = R.T(rng.normal(size=(n, h, w, c1)), True)
    tx2 = R.T(rng.normal(size=(n, h, w, c2)), True) if c2 else None
    xin = tx if tx2 is None else torch.cat([tx, tx2 * x2s], -1)
    (R.conv2d(xin, R.T(wt), None, stride)[..., :cv] * R.T(dz)).sum().backward()
    for which, t in ((0, tx),) + (((1, tx2),) if c2 else ()):
        nws = lib.shdr_conv2d_dgrad_workspace_bytes_f32(ctypes.byref(d), which)
        assert nws > 0 and nws == lib.shdr_workspace_bytes(1, ctypes.byref(d), which)
        ws = torch.empty(nws, device="cuda", dtype=torch.uint8)
        dx = torch.full((n, h, w, c2 if which else c1), float("nan"), device="cuda")      # every element must be written
        dzd, wd = dev(dz), dev(wt)                                                         # (the caller keeps its buffers alive)
        check(lib, lib.shdr_conv2d_dgrad_f32(ctypes.byref(d), which, ptr(dzd), ptr(wd), ptr(dx), ptr(ws), stream()))
        torch.cuda.synchronize()
        assert rel_err(dx.cpu().numpy(), t.grad.numpy()) <= 1e-5, (name, which)
