#!/usr/bin/env python3
"""Time direct vs three-kernel Winograd vs fused Winograd on the 3x3 layer shapes of the path (batch 16, 512^2 input)."""
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
K.WINOGRAD = False
for hw, cin, cout in ((512, 64, 64), (256, 64, 128), (256, 128, 128), (128, 128, 256), (128, 256, 256), (64, 256, 512), (64, 512, 512),
                      (32, 512, 512), (16, 512, 512), (128, 512, 256), (256, 256, 128), (512, 128, 64), (64, 128, 128), (64, 256, 128)):
    x = torch.randn(N, hw, hw, cin, device="cuda"); w = torch.randn(3, 3, cin, cout, device="cuda") * 0.02; b = torch.randn(cout, device="cuda")
    u = K.winograd_filter(w); up = K.winograd_filter_packed(w)
    ref = K.conv2d(x, w, b, act1=1)
    fu = K.conv2d_winograd_fused(x, up, b, act1=1)
    err = float((fu - ref).abs().max() / ref.abs().max())
    d = t(lambda: K.conv2d(x, w, b, act1=1))
    wi = t(lambda: K.conv2d_winograd(x, u, b, act1=1)) if cin % 32 == 0 else float("nan")
    f = t(lambda: K.conv2d_winograd_fused(x, up, b, act1=1))
    gf = 2.0 * N * hw * hw * cin * cout * 9 / 1e9
    print("%4d^2 %4d->%-4d direct %7.3f ms (%6.1f TF)  wino3 %7.3f ms (%6.1f)  fused %7.3f ms (%6.1f TF alg)  err %.2e" % (hw, cin, cout, d, gf / d, wi, gf / wi, f, gf / f, err), flush=True)
    del x, w, u, up, ref, fu
    torch.cuda.empty_cache()
