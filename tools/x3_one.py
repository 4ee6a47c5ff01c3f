#!/usr/bin/env python3
"""One wide layer on the split-operand kernel, ten launches (PMC / rocprof target): python3 tools/x3_one.py HW CIN COUT [N [K [C2]]]
(K = 3 or 1; C2 = channels of a second source)"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("singlehdr-tf2_amd")._ops
hw, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 16
KS = int(sys.argv[5]) if len(sys.argv) > 5 else 3
C2 = int(sys.argv[6]) if len(sys.argv) > 6 else 0
with torch.no_grad():
    x = torch.randn(N, hw, hw, cin, device="cuda")
    x2 = torch.randn(N, hw, hw, C2, device="cuda") if C2 else None
    w = (torch.randn(KS, KS, cin + C2, cout, device="cuda") * 0.02).requires_grad_(True)
    b = torch.randn(cout, device="cuda")
    assert K.conv2d_plan((N, hw, hw, cin), tuple(w.shape), c2=C2) == "x3"
    if hasattr(K, "absmax_slot"):
        K.absmax_slot(x)
        if x2 is not None:
            K.absmax_slot(x2)
    for _ in range(3): K.conv2d(x, w, b, x2=x2, act1=K.ACT_RELU)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): K.conv2d(x, w, b, x2=x2, act1=K.ACT_RELU)
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
gf = 2.0 * N * hw * hw * (cin + C2) * cout * KS * KS / 1e9
print("x3 %d^2 %d->%d: %.3f ms  %.1f TF alg  (%.2f of the fp16 MFMA peak executed)" % (hw, cin, cout, ms, gf / ms, 3 * gf / ms / 2500.0))
