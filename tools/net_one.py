#!/usr/bin/env python3
"""one network's inference forward, batch x size^2 (same-box A/B of two trees):  python tools/net_one.py deq|lin|hal|ref [batch] [size]"""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("singlehdr-tf2_amd")
name = sys.argv[1]
b = int(sys.argv[2]) if len(sys.argv) > 2 else 16
sz = int(sys.argv[3]) if len(sys.argv) > 3 else 512
torch.manual_seed(1)
net = {"deq": pkg.dequantization_net, "lin": pkg.linearization_net, "hal": pkg.hallucination_net, "ref": pkg.refinement_net}[name].model()
x = torch.rand(b, sz, sz, 12 if name == "ref" else 3, device="cuda")
if hasattr(pkg._ops, "set_bound"):          # as in the pipeline: an LDR image is in [0, 1] by construction (no measuring pass)
    pkg._ops.set_bound(x, 1.0)
with torch.no_grad():
    for _ in range(3):
        net(x, training=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10):
        net(x, training=False)
    e1.record(); torch.cuda.synchronize()
print("%s %d x %d^2: %.3f ms" % (name, b, sz, e0.elapsed_time(e1) / 10))
for k, v in getattr(pkg._ops, "RANGE_MISSES", {}).items():
    print("   range measured %d x:" % v, k)
