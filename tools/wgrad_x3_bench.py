#!/usr/bin/env python3
"""The deep layers' weight gradients of the joint step (batch 32 x 256^2) on the split-operand kernel (csrc/wgrad_x3.hip, incl. its split
passes) vs the exact-fp32 plan (Winograd-domain / MFMA kernels):   python tools/wgrad_x3_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("singlehdr-tf2_amd")._ops
SHAPES = [  # n, h, w, c1, c2, cout, k
    (32, 32, 32, 512, 0, 512, 3), (32, 16, 16, 512, 0, 512, 3), (32, 64, 64, 256, 0, 256, 3), (32, 64, 64, 512, 0, 256, 3),
    (32, 128, 128, 256, 0, 128, 3), (32, 128, 128, 128, 0, 128, 3), (32, 64, 64, 128, 0, 256, 3), (32, 32, 32, 256, 0, 512, 3),
    (32, 32, 32, 512, 512, 512, 1), (32, 64, 64, 256, 256, 256, 1), (32, 128, 128, 128, 128, 128, 1), (32, 256, 256, 128, 0, 64, 3),
    (4, 256, 256, 256, 0, 256, 3), (4, 512, 512, 128, 0, 128, 3),
]


def timeit(fn, reps=5):
    for _ in range(2):
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
    for n, h, w, c1, c2, cout, k in SHAPES:
        x = torch.randn(n, h, w, c1, device="cuda")
        x2 = torch.randn(n, h, w, c2, device="cuda") if c2 else None
        dz = torch.randn(n, h, w, cout, device="cuda") * 1e-4
        ws = (k, k, c1 + c2, cout)
        out = torch.zeros(ws, device="cuda")
        fl = 2.0 * n * h * w * (c1 + c2) * cout * k * k
        res = {}
        for exact in (False, True):
            K.EXACT_FP32 = exact
            res[exact] = (timeit(lambda: K.conv2d_wgrad(x, x2, dz, ws, 1, 1.0, out=out)), K.conv2d_wgrad(x, x2, dz, ws, 1, 1.0))
        K.EXACT_FP32 = False
        d = float((res[False][1] - res[True][1]).abs().max() / res[True][1].abs().max())
        print("%3dx%3dx%-3d %3d+%-3d->%-3d k%d  split %6.3f ms %6.1f TF/s alg | exact %6.3f ms %6.1f TF/s alg  x%.2f  max|diff|/max %.1e"
              % (n, h, w, c1, c2, cout, k, res[False][0], fl / res[False][0] / 1e9, res[True][0], fl / res[True][0] / 1e9,
                 res[True][0] / res[False][0], d), flush=True)
