#!/usr/bin/env python3
"""Time direct vs Winograd conv on the Hallucination-Net 3x3 layer shapes (batch 16, 512^2 input)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("singlehdr-tf2_amd")._ops
def t(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for hw, cin, cout in ((512, 64, 64), (256, 64, 128), (256, 128, 128), (128, 128, 256), (128, 256, 256), (64, 256, 512), (64, 512, 512),
                      (32, 512, 512), (16, 512, 512), (128, 512, 256), (256, 256, 128), (512, 128, 64)):
    x = torch.randn(N, hw, hw, cin, device="cuda"); w = torch.randn(3, 3, cin, cout, device="cuda") * 0.02; b = torch.randn(cout, device="cuda")
    u = K.winograd_filter(w)
    d = t(lambda: K.conv2d(x, w, b, act1=1)); wi = t(lambda: K.conv2d_winograd(x, u, b, act1=1))
    gf = 2.0 * N * hw * hw * cin * cout * 9 / 1e9
    print("%4d^2 %4d->%-4d direct %7.3f ms (%6.1f TF)   winograd %7.3f ms (%6.1f TF alg)   ratio %.2f" % (hw, cin, cout, d, gf / d, wi, gf / wi, d / wi))
    del x, w, u
    torch.cuda.empty_cache()
