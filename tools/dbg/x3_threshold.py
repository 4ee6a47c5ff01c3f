#!/usr/bin/env python3
"""split-operand kernel vs fused Winograd at low block counts (the plan's fill-the-chip threshold)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
K = importlib.import_module("singlehdr-tf2_amd")._ops
os.environ["SHDR_X3_MIN_BLOCKS"] = "1"


def timeit(fn, reps=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


with torch.no_grad():
    for n, hw, cin, cout in ((1, 64, 512, 512), (2, 64, 512, 512), (4, 32, 512, 512), (8, 32, 512, 512), (16, 16, 512, 512), (1, 128, 256, 256), (1, 256, 128, 128),
                             (1, 512, 64, 64), (8, 16, 512, 512), (2, 32, 512, 512), (1, 32, 512, 512)):
        x = torch.randn(n, hw, hw, cin, device="cuda")
        w = (torch.randn(3, 3, cin, cout, device="cuda") * 0.02).requires_grad_(True)
        blocks = n * ((hw + 15) // 16) ** 2 * (cout // 64)
        K.EXACT_FP32 = False
        t3 = timeit(lambda: K.conv2d(x, w))
        K.EXACT_FP32 = True
        tw = timeit(lambda: K.conv2d(x, w))
        K.EXACT_FP32 = False
        print("n %2d %3d^2 %d->%d  blocks %5d   x3 %.4f ms  winograd %.4f ms  ratio %.2f" % (n, hw, cin, cout, blocks, t3, tw, tw / t3), flush=True)
