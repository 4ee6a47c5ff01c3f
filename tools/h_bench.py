#!/usr/bin/env python3
"""Per-layer timing of the native-fp16 conv kernels (forward = dgrad kernel, and the weight gradient) at the shapes of the
fine-tuning step (batch 4 x 1024^2):   python tools/h_bench.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shdr = importlib.import_module("singlehdr-tf2_amd")
K = shdr._ops

SHAPES = [  # n, h, w, c1, c2, cout, k, stride
    (4, 1024, 1024, 64, 0, 64, 3, 1), (4, 512, 512, 64, 0, 128, 3, 1), (4, 512, 512, 128, 0, 128, 3, 1),
    (4, 256, 256, 256, 0, 256, 3, 1), (4, 128, 128, 512, 0, 512, 3, 1), (4, 1024, 1024, 128, 0, 64, 3, 1),
    (4, 512, 512, 256, 0, 128, 3, 1), (4, 256, 256, 512, 0, 256, 3, 1), (4, 1024, 1024, 64, 64, 64, 1, 1),
    (4, 1024, 1024, 16, 0, 16, 7, 1), (4, 1024, 1024, 8, 0, 16, 7, 1), (4, 1024, 1024, 32, 0, 16, 3, 1),
    (4, 1024, 1024, 16, 16, 16, 3, 1), (4, 512, 512, 16, 0, 32, 5, 1), (4, 512, 512, 32, 0, 32, 5, 1),
    (4, 1024, 1024, 96, 0, 64, 7, 2), (4, 256, 256, 64, 0, 256, 1, 1), (4, 256, 256, 256, 0, 64, 1, 1),
    (4, 1024, 1024, 16, 0, 64, 3, 1), (4, 64, 64, 512, 0, 512, 3, 1), (4, 128, 128, 256, 0, 512, 3, 1), (4, 256, 256, 128, 0, 256, 3, 1),
    (4, 128, 128, 1024, 0, 512, 1, 1), (4, 128, 128, 128, 0, 512, 1, 1), (4, 128, 128, 512, 0, 128, 1, 1),
]


def timeit(fn, reps=20):
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


if len(sys.argv) > 1 and sys.argv[1] == "wide":       # the layers of the 256-wide weight-gradient tiles
    SHAPES = [t for t in SHAPES if (t[3] + t[4]) % 128 == 0 and t[5] % 128 == 0 and (t[3] + t[4] >= 256 or t[5] >= 256)]
for n, h, w, c1, c2, cout, k, s in SHAPES:
    x = torch.randn(n, h, w, c1, device="cuda").half()
    x2 = torch.randn(n, h, w, c2, device="cuda").half() if c2 else None
    wt = torch.randn(k, k, c1 + c2, cout, device="cuda") / (k * (c1 + c2) ** 0.5)
    b = torch.randn(cout, device="cuda")
    wp = K.pack_filter_h(wt, c1, c2)
    y = K.conv2d_h(x, wp, b, (k, k), cout, stride=s, x2=x2, act1=K.ACT_RELU)
    flops = 2.0 * y.shape[0] * y.shape[1] * y.shape[2] * (c1 + c2) * cout * k * k
    tf = timeit(lambda: K.conv2d_h(x, wp, b, (k, k), cout, stride=s, x2=x2, act1=K.ACT_RELU))
    dz = torch.randn_like(y)
    tw = timeit(lambda: K.conv2d_wgrad_h(x, x2, dz, (k, k, c1 + c2, cout), s))
    byts = (x.numel() + (x2.numel() if c2 else 0) + y.numel()) * 2
    print("%4dx%-4d %3d+%-3d->%-3d k%d s%d  fwd %7.3f ms %7.1f TF/s (%5.2f TB/s min)   wgrad %7.3f ms %7.1f TF/s"
          % (h, w, c1, c2, cout, k, s, tf, flops / tf / 1e9, byts / tf / 1e9, tw, flops / tw / 1e9), flush=True)
