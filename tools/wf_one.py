#!/usr/bin/env python3
"""Time the fused Winograd kernel on one shape: python3 tools/wf_one.py HW CIN COUT [N]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = importlib.import_module("singlehdr-tf2_amd")._ops
hw, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 16
x = torch.randn(N, hw, hw, cin, device="cuda"); w = torch.randn(3, 3, cin, cout, device="cuda") * 0.02
u = K.winograd_filter_packed(w)
b = torch.randn(cout, device="cuda"); sc = torch.rand(cout, device="cuda") + 0.5; sh = torch.randn(cout, device="cuda")
kw = dict(bias=b, act1=K.ACT_RELU, scale=sc, shift=sh, act2=K.ACT_RELU) if os.environ.get("WF_EPI") else {}
for _ in range(3): K.conv2d_winograd_fused(x, u, **kw)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(10): K.conv2d_winograd_fused(x, u, **kw)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
gf = 2.0 * N * hw * hw * cin * cout * 9 / 1e9
print("%s fused %d^2 %d->%d: %.3f ms  %.1f TF alg  (%.2f of MFMA peak executed)" % (os.environ.get("SHDR_LIB", "default").split("/")[-1], hw, cin, cout, ms, gf / ms, gf / ms / 2.25 / 157.3))
