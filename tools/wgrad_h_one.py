#!/usr/bin/env python3
"""A few launches of the fp16 weight gradient on ONE layer (for PMC passes):  python3 tools/wgrad_h_one.py HW CIN COUT [K] [N]"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
shdr = importlib.import_module("singlehdr-tf2_amd")
K = shdr._ops
hw, cin, cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
k = int(sys.argv[4]) if len(sys.argv) > 4 else 3
n = int(sys.argv[5]) if len(sys.argv) > 5 else 4
x = torch.randn(n, hw, hw, cin, device="cuda").half()
dz = torch.randn(n, hw, hw, cout, device="cuda").half()
for _ in range(5):
    dw = K.conv2d_wgrad_h(x, None, dz, (k, k, cin, cout), 1)
torch.cuda.synchronize()
print("ok", float(dw.abs().mean()))
