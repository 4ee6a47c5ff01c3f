#!/usr/bin/env python3
"""what the range tracking of the backward elementwise kernels costs: each kernel with and without a range slot, same tensors"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("singlehdr-tf2_amd")
K, L = pkg._ops, pkg._lib
lib = L.load()
P = K._ptr
st = lambda: K._stream()


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for shape in ((32, 64, 64, 256), (32, 128, 128, 64), (32, 32, 32, 512)):
    n, h, w, c = shape
    x, dy, y = (torch.randn(shape, device="cuda") for _ in range(3))
    mean, var, gamma = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda"), torch.ones(c, device="cuda")
    ws = K._bn_ws(c, x.device)
    dg, db, dx = torch.zeros(c, device="cuda"), torch.zeros(c, device="cuda"), torch.empty_like(x)
    slot = torch.zeros(1, device="cuda")
    npix = n * h * w
    t0 = timeit(lambda: lib.shdr_bn_bwd_ranged_f32(P(dy), P(x), P(y), P(mean), P(var), P(gamma), P(ws), P(dg), P(db), P(dx), npix, c, 1e-3, None, st()))
    t1 = timeit(lambda: lib.shdr_bn_bwd_ranged_f32(P(dy), P(x), P(y), P(mean), P(var), P(gamma), P(ws), P(dg), P(db), P(dx), npix, c, 1e-3, P(slot), st()))
    a0 = timeit(lambda: lib.shdr_add_ranged_f32(P(x), P(dy), P(dx), x.numel(), None, st()))
    a1 = timeit(lambda: lib.shdr_add_ranged_f32(P(x), P(dy), P(dx), x.numel(), P(slot), st()))
    m0 = timeit(lambda: lib.shdr_bn_train_apply_ranged_f32(P(x), P(mean), P(var), P(gamma), P(mean), P(dx), npix, c, 1e-3, 1, None, st()))
    m1 = timeit(lambda: lib.shdr_bn_train_apply_ranged_f32(P(x), P(mean), P(var), P(gamma), P(mean), P(dx), npix, c, 1e-3, 1, P(slot), st()))
    print("%s  bn_bwd %.1f -> %.1f us   add %.1f -> %.1f us   bn_apply %.1f -> %.1f us" % (shape, t0, t1, a0, a1, m0, m1))
