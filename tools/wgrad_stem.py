#!/usr/bin/env python3
"""the weight gradient of the Linearization-Net stem (7x7 / 2, 96 -> 64) at the joint step's size (32 x 256^2)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("singlehdr-tf2_amd")._ops
x = torch.randn(32, 256, 256, 96, device="cuda")
dz = torch.randn(32, 128, 128, 64, device="cuda") * 1e-4
for exact in (False, True):
    K.EXACT_FP32 = exact
    with torch.no_grad(), K.range_scope():
        for _ in range(2):
            K.conv2d_wgrad(x, None, dz, (7, 7, 96, 64), 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(5):
            K.conv2d_wgrad(x, None, dz, (7, 7, 96, 64), 2)
        e1.record(); torch.cuda.synchronize()
    print("exact" if exact else "split", "%.3f ms" % (e0.elapsed_time(e1) / 5))
