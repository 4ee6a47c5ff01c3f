#!/usr/bin/env python3
"""HBM rate of the two store-bound front-end kernels at the bench shape (16 x 512^2)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
K = importlib.import_module("singlehdr-tf2_amd")._ops
x = torch.rand(16, 512, 512, 3, device="cuda")
def t(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
nb = 16 * 512 * 512 * (12 + 384)
for name, fn in (("soft_hist B=32", lambda: K.soft_hist(x, 32)), ("lin_frontend 96ch", lambda: K.lin_frontend(x, 96))):
    ms = t(fn)
    print("%-20s %.4f ms  %.2f TB/s  %.3f of 8 TB/s" % (name, ms, nb / ms / 1e9, nb / ms / 1e9 / 8))
