#!/usr/bin/env python3
"""Bilinear 2x + Conv2D 3x3 of the decoders' `up` blocks at the inference shapes (batch 16 x 512^2), fused prologue vs the two
kernels:   python tools/up2_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shdr = importlib.import_module("singlehdr-tf2_amd")
K = shdr._ops

SHAPES = [  # n, low-res h, w, cin, cout
    (16, 16, 16, 512, 512), (16, 32, 32, 512, 512), (16, 64, 64, 512, 256), (16, 128, 128, 256, 128), (16, 256, 256, 128, 64),
    (16, 32, 32, 256, 128), (16, 64, 64, 128, 64), (16, 128, 128, 64, 32), (16, 256, 256, 32, 16),
]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    for n, h, w, cin, cout in SHAPES:
        x = torch.randn(n, h, w, cin, device="cuda")
        wt = (torch.randn(3, 3, cin, cout, device="cuda") / (3 * cin ** 0.5)).requires_grad_(True)    # a persistent variable: filter prepared once
        b = torch.randn(cout, device="cuda")
        xu = K.resize2x(x)
        t_r = timeit(lambda: K.resize2x(x))
        t_c = timeit(lambda: K.conv2d(xu, wt, b, act1=K.ACT_RELU))
        t_f = timeit(lambda: K.conv2d_up2(x, wt, b, act1=K.ACT_RELU))
        flops = 2.0 * n * 4 * h * w * cin * cout * 9
        print("%3dx%-3d -> x2  %3d->%-3d  resize %6.3f + conv %6.3f = %6.3f ms   fused %6.3f ms (%6.1f TF/s algorithmic)  plan %s"
              % (h, w, cin, cout, t_r, t_c, t_r + t_c, t_f, flops / t_f / 1e9, K.conv2d_plan((n, 2 * h, 2 * w, cin), wt.shape)), flush=True)
