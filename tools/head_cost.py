#!/usr/bin/env python3
"""the 16 -> 3 head of the U-Nets (3x3, filter padded to 16 couts, 3 stored) with and without tanh / residual, next to the plain 16 -> 16 layer"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("singlehdr-tf2_amd")._ops
x = torch.rand(16, 512, 512, 16, device="cuda")
K.set_bound(x, 1.0)
w = (torch.randn(3, 3, 16, 16, device="cuda") * 0.05).requires_grad_(True)
b3, b16 = torch.zeros(3, device="cuda"), torch.zeros(16, device="cuda")
res = torch.rand(16, 512, 512, 3, device="cuda")
cases = [("16->16 lrelu", dict(bias=b16, act1=K.ACT_LRELU)), ("16->3 none", dict(bias=b3, cout_valid=3)),
         ("16->3 tanh", dict(bias=b3, cout_valid=3, act1=K.ACT_TANH)), ("16->3 none+res", dict(bias=b3, cout_valid=3, residual=res)),
         ("16->3 tanh+res", dict(bias=b3, cout_valid=3, act1=K.ACT_TANH, residual=res))]
with torch.no_grad(), K.range_scope():
    for name, kw in cases:
        bias = kw.pop("bias")
        for _ in range(3):
            K.conv2d(x, w, bias, **kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(20):
            K.conv2d(x, w, bias, **kw)
        e1.record(); torch.cuda.synchronize()
        print("%-16s %.3f ms" % (name, e0.elapsed_time(e1) / 20))
