#!/usr/bin/env python3
"""why is the in-kernel up-sampling layer slower inside the net than alone?  vary epilogue and input statistics"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
K = importlib.import_module("singlehdr-tf2_amd")._ops


def timeit(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    n, h, w, cin, cout = 16, 256, 256, 128, 64
    wt = (torch.randn(3, 3, cin, cout, device="cuda") / (3 * cin ** 0.5)).requires_grad_(True)
    b, sc, sh = torch.randn(cout, device="cuda"), torch.rand(cout, device="cuda") + 0.5, torch.randn(cout, device="cuda")
    for name, x in (("randn", torch.randn(n, h, w, cin, device="cuda")), ("relu(randn)", torch.randn(n, h, w, cin, device="cuda").relu()),
                    ("randn*1e3", torch.randn(n, h, w, cin, device="cuda") * 1e3), ("randn*1e-6", torch.randn(n, h, w, cin, device="cuda") * 1e-6),
                    ("zeros", torch.zeros(n, h, w, cin, device="cuda"))):
        t1 = timeit(lambda: K.conv2d_up2(x, wt, b, act1=K.ACT_RELU))
        t2 = timeit(lambda: K.conv2d_up2(x, wt, b, act1=K.ACT_RELU, scale=sc, shift=sh, act2=K.ACT_RELU))
        xu = K.resize2x(x)
        t3 = timeit(lambda: K.conv2d(xu, wt, b, act1=K.ACT_RELU))
        print("%-12s up2 fused %.3f ms, with folded BN %.3f ms, plain x3 on the up-sampled tensor %.3f ms" % (name, t1, t2, t3), flush=True)
