#!/usr/bin/env python3
"""the image-side layers of the three networks (4 -> 16 7x7, 4 -> 64 3x3, 16 -> 16 7x7 at 16 x 512^2) through K.conv2d, input bound known"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("singlehdr-tf2_amd")
K = pkg._ops
torch.manual_seed(0)
for cin, cout, k in ((4, 16, 7), (4, 64, 3), (16, 16, 7), (64, 64, 3)):
    x = torch.rand(16, 512, 512, cin, device="cuda")
    if hasattr(K, "set_bound"):
        K.set_bound(x, 1.0)
    w = torch.randn(k, k, cin, cout, device="cuda") * 0.05
    b = torch.zeros(cout, device="cuda")
    import contextlib
    with torch.no_grad(), (K.range_scope() if hasattr(K, "range_scope") else contextlib.nullcontext()):
        for _ in range(3):
            K.conv2d(x, w, b, act1=K.ACT_RELU)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            K.conv2d(x, w, b, act1=K.ACT_RELU)
        e1.record(); torch.cuda.synchronize()
    print("%2d -> %2d %dx%d %s: %.3f ms" % (cin, cout, k, k, K.conv2d_plan(tuple(x.shape), tuple(w.shape)), e0.elapsed_time(e1) / 20))
