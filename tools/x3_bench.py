#!/usr/bin/env python3
"""The wide 3x3 layers of the inference step (batch 16 x 512^2) on the split-operand fp16 kernel (plan "x3") vs the exact-fp32 plan
(fused Winograd):   python tools/x3_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shdr = importlib.import_module("singlehdr-tf2_amd")
K = shdr._ops

SHAPES = [  # n, h, w, c1, c2, cout
    (16, 512, 512, 64, 0, 64), (16, 256, 256, 64, 0, 128), (16, 256, 256, 128, 0, 128), (16, 128, 128, 128, 0, 256),
    (16, 128, 128, 256, 0, 256), (16, 64, 64, 256, 0, 512), (16, 64, 64, 512, 0, 512), (16, 32, 32, 512, 0, 512),
    (16, 512, 512, 128, 0, 64), (16, 256, 256, 256, 0, 128), (16, 128, 128, 512, 0, 256), (16, 128, 128, 64, 64, 64),
    (16, 128, 128, 32, 0, 64), (16, 64, 64, 128, 128, 128),
    (16, 512, 512, 96, 0, 64, 7, 2),          # the Linearization-Net stem: four phase launches
    (16, 32, 32, 512, 512, 512, 1, 1), (16, 64, 64, 512, 512, 512, 1, 1), (16, 128, 128, 256, 256, 256, 1, 1),      # hal skip layers (1x1 on a concat)
    (16, 256, 256, 128, 128, 128, 1, 1), (16, 512, 512, 64, 64, 64, 1, 1), (16, 128, 128, 256, 0, 64, 1, 1), (16, 64, 64, 512, 0, 128, 1, 1),
    # the narrow full-resolution layers of the U-Nets (plan "x3n")
    (16, 512, 512, 4, 0, 16, 7, 1), (16, 512, 512, 16, 0, 16, 7, 1), (16, 256, 256, 16, 0, 32, 5, 1), (16, 256, 256, 32, 0, 32, 5, 1),
    (16, 512, 512, 32, 0, 16, 3, 1), (16, 512, 512, 16, 16, 16, 3, 1), (16, 512, 512, 16, 0, 16, 3, 1),
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


import contextlib
with torch.no_grad(), (K.range_scope() if hasattr(K, "range_scope") else contextlib.nullcontext()):
    for shape in SHAPES:
        n, h, w, c1, c2, cout = shape[:6]
        k, st = (shape[6], shape[7]) if len(shape) > 6 else (3, 1)
        x = torch.randn(n, h, w, c1, device="cuda")
        x2 = torch.randn(n, h, w, c2, device="cuda") if c2 else None
        if hasattr(K, "absmax_slot"):          # the range slots a producing kernel would have written (none: measured per call)
            K.absmax_slot(x)
            if x2 is not None:
                K.absmax_slot(x2)
        wt = (torch.randn(k, k, c1 + c2, cout, device="cuda") / (k * (c1 + c2) ** 0.5)).requires_grad_(True)   # persistent: prepared once
        b = torch.randn(cout, device="cuda")
        flops = 2.0 * n * (h // st) * (w // st) * (c1 + c2) * cout * k * k
        res = {}
        for exact in (False, True):
            K.EXACT_FP32 = exact
            plan = K.conv2d_plan((n, h, w, c1), tuple(wt.shape), c2=c2, stride=st)
            t = timeit(lambda: K.conv2d(x, wt, b, stride=st, x2=x2, act1=K.ACT_RELU))
            res[exact] = (plan, t, K.conv2d(x, wt, b, stride=st, x2=x2, act1=K.ACT_RELU))
        K.EXACT_FP32 = False
        d = float((res[False][2] - res[True][2]).abs().max() / res[True][2].abs().max())
        print("%3dx%-3d %3d+%-3d->%-3d  %-6s %6.3f ms %6.1f TF/s alg   | exact %-6s %6.3f ms %6.1f TF/s alg   x%.2f   max|diff|/max %.1e"
              % (h, w, c1, c2, cout, res[False][0], res[False][1], flops / res[False][1] / 1e9, res[True][0], res[True][1],
                 flops / res[True][1] / 1e9, res[True][1] / res[False][1], d), flush=True)
