#!/usr/bin/env python3
"""the 1x1 layers of the Linearization-Net's ResNet blocks without residual, default plan vs SHDR_X3_1X1_MIN_K=64 (two processes)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("singlehdr-tf2_amd")._ops
for hw, cin, cout in ((128, 64, 256), (128, 64, 64), (64, 128, 512), (128, 256, 64)):
    x = torch.randn(16, hw, hw, cin, device="cuda")
    w = (torch.randn(1, 1, cin, cout, device="cuda") * 0.05).requires_grad_(True)
    sc, sh = torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    with torch.no_grad(), K.range_scope():
        K.absmax_slot(x)
        for _ in range(3):
            K.conv2d(x, w, None, scale=sc, shift=sh, act2=K.ACT_RELU)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            K.conv2d(x, w, None, scale=sc, shift=sh, act2=K.ACT_RELU)
        e1.record(); torch.cuda.synchronize()
    print("%3d^2 %3d -> %3d  %-5s %.3f ms" % (hw, cin, cout, K.conv2d_plan(tuple(x.shape), tuple(w.shape)), e0.elapsed_time(e1) / 20))
